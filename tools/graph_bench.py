#!/usr/bin/env python3
"""tools/graph_bench.py [--workload cora_shape] -- one Sync reverse step enqueued eagerly (ctypes calls from Python, two streams)
against the same step captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed: what graph replay would save at a
size.  The replay repeats the captured Philox key (the seed is a kernel argument), so this is a timing tool, not a sampler."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cora_shape")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    from graphmaker_amd import ops
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[args.workload]
    model = bench.build_model(wl, dev)
    n, T = wl["N"], wl["T"]
    Em = model.transition.E_marginal_host
    Y32 = ops.prior_Y(n, model.Y_marginal, seed=0)
    bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=0)
    x, oh = ops.prior_X(n, model.X_marginal, seed=0)

    def step():
        return model.denoise_step(T, x, oh, Y32, bits, seed=7)

    for _ in range(3):
        want = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / args.steps * 1e3
    graph = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cap):
        step()
        with torch.cuda.graph(graph, stream=cap):
            got = step()
    torch.cuda.current_stream(dev).wait_stream(cap)
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(want, got))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"workload": wl["name"], "eager_ms_per_step": eager, "graph_replay_ms_per_step": replay,
                      "replay_equals_eager": same}))


if __name__ == "__main__":
    main()
