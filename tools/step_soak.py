#!/usr/bin/env python3
"""Run the whole Sync reverse step (two streams: attribute branch beside the edge encoder, then the pair kernel) many times
on the same input and require every result to equal the first bit for bit -- a stream-ordering bug shows up as a rare
difference.  Also the 1-rank sharded schedule (each tower on its own stream) when --plan is given.
    python tools/step_soak.py [--reps 300] [--workload amazon_computer|cora_shape] [--plan]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--workload", default="amazon_computer")
    ap.add_argument("--plan", action="store_true")
    args = ap.parse_args()
    wl = bench.WORKLOAD if args.workload == "amazon_computer" else bench.SMALL
    dev = torch.device("cuda:0")
    from graphmaker_amd import ops
    model = bench.build_model(wl, dev)
    n, T = wl["N"], wl["T"]
    Em = model.transition.E_marginal_host
    Y32 = ops.prior_Y(n, model.Y_marginal, seed=1)
    bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=1)
    x, oh = ops.prior_X(n, model.X_marginal, seed=1)
    plan = group = None
    if args.plan:
        import torch.distributed as dist
        from graphmaker_amd.parallel import ShardPlan
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
        plan, group = ShardPlan.make(n, 1, 0), dist.group.WORLD
    first = None
    bad = 0
    for r in range(args.reps):
        xs, ohs, bs = model.denoise_step(T, x, oh, Y32, bits, seed=7, plan=plan, group=group)
        if first is None:
            first = (xs.clone(), ohs.clone(), bs.clone())
        elif not (torch.equal(xs, first[0]) and torch.equal(ohs, first[1]) and torch.equal(bs, first[2])):
            bad += 1
        if r % 100 == 99:
            print(f"{r + 1} steps, {bad} differing", flush=True)
    if model._pair_ws is not None:
        ops.check_pair_workspace(model._pair_ws)
    print(f"{args.workload}{' (1-rank plan over RCCL)' if args.plan else ''}: {args.reps} steps, {bad} differ from the first; "
          f"edges {int(ops.csr_from_bitmap(first[2], n)[0][n])}, attributes on {int(first[0].sum())}")
    if args.plan:
        import torch.distributed as dist
        dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
