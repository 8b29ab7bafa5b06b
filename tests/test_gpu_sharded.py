"""Multi-rank runs of the sharded sample() with the real HIP kernels on ONE GPU (GPU box).

RCCL refuses two ranks on one device, so the ranks of these tests share cuda:0 and exchange their shards over the
`gloo` backend (device tensors staged through the host by torch); the product code path -- ShardPlan, the alternating
tower stages with started-then-waited all-gathers, the one-launch folded pair chunks, the chunk-wise gather of the
sampled rows -- is the one the RCCL ranks of bench.py run.  Every rank must reproduce the single-GPU sample bit for bit.
A one-rank RCCL group covers the NCCL stream hand-over of the same path.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(name, dev):
    import graphmaker_amd as gm
    cfg = gu.CASES[name]
    Xm, Ym, Em = gu.marginals(cfg)
    sdX, sdE = gu.make_checkpoint(cfg)
    kw = dict(X_marginal=torch.from_numpy(Xm).to(dev), Y_marginal=torch.from_numpy(Ym).to(dev),
              E_marginal=torch.from_numpy(Em).to(dev), num_nodes=cfg["N"])
    if cfg["variant"] == "Sync":
        model = gm.ModelSync(T=cfg["T"], gnn_X_config=cfg["gnn_X"], gnn_E_config=cfg["gnn_E"], **kw)
    else:
        model = gm.ModelAsync(T_X=cfg["T_X"], T_E=cfg["T_E"], mlp_X_config=cfg["mlp_X"], gnn_E_config=cfg["gnn_E"], **kw)
    model.graph_encoder.pred_X.load_state_dict({k: torch.from_numpy(v) for k, v in sdX.items()})
    model.graph_encoder.pred_E.load_state_dict({k: torch.from_numpy(v) for k, v in sdE.items()})
    return cfg, model.to(dev).eval()


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def _worker(rank, world, port, backend, name, q):
    import faulthandler
    faulthandler.dump_traceback_later(45, exit=True)             # a stuck collective names itself instead of hanging
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    kw = dict(device_id=dev) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        from graphmaker_amd.parallel import ShardPlan
        name, _, variant = name.partition(":")
        cfg, model = _model(name, dev)
        plan = ShardPlan.make(cfg["N"], world, rank, replicate_graph_part=variant == "replicated")
        want = model.sample(seed=cfg["seed"])
        got = model.sample(seed=cfg["seed"], plan=plan)
        again = model.sample(seed=cfg["seed"], plan=plan)          # persistent gather buffers are reused correctly
        torch.cuda.synchronize()
        q.put((rank, _same(want, got), _same(want, again), int(want[2].sum().item())))
    finally:
        dist.destroy_process_group()


def _worker_full(rank, world, port, backend, name, q):
    """One whole reverse step at a BASELINE.json size (reference hidden sizes, random-init weights, prior-drawn state):
    the sharded step of every rank must equal the single-GPU step bit for bit."""
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from graphmaker_amd import ops
        from graphmaker_amd.parallel import ShardPlan
        wl = dict(bench.WORKLOAD, name="amazon_photo", N=7650, F=745, C=8, density=4.07e-3)
        model = bench.build_model(wl, dev)
        n, T = wl["N"], wl["T"]
        Em = model.transition.E_marginal_host
        Y32 = ops.prior_Y(n, model.Y_marginal, seed=3)
        bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=3)
        x, oh = ops.prior_X(n, model.X_marginal, seed=3)
        want = model.denoise_step(T, x, oh, Y32, bits, seed=5)
        got = model.denoise_step(T, x, oh, Y32, bits, seed=5, plan=ShardPlan.make(n, world, rank))
        torch.cuda.synchronize()
        q.put((rank, _same(want, got), True, int(want[2].ne(0).sum().item())))
    finally:
        dist.destroy_process_group()


def _worker_overflow(rank, world, port, backend, name, q):
    """An exchange segment that is too small for ONE rank's edges: every rank must reject the sample (a rank that carried on
    with a truncated graph would diverge from the others and hang the next collective)."""
    import faulthandler
    faulthandler.dump_traceback_later(45, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from graphmaker_amd import ops
        from graphmaker_amd.parallel import ShardPlan
        cfg, model = _model(name, dev)
        n, T, seed = cfg["N"], cfg["T"], cfg["seed"]
        Em = model.transition.E_marginal_host
        Y32 = ops.prior_Y(n, model.Y_marginal, seed=seed)
        bits = ops.prior_E(n, float(Em[0]), float(Em[1]), dev, seed=seed)
        x, oh = ops.prior_X(n, model.X_marginal, seed=seed)
        want = model.denoise_step(T, x, oh, Y32, bits, seed)
        dense = ops.dense_from_bitmap(want[2], n).triu(1)
        plans = [ShardPlan.make(n, world, r) for r in range(world)]
        per_rank = [int(sum(dense[b:e].sum().item() for b, e in pl.pair_chunks())) for pl in plans]
        assert min(per_rank) < max(per_rank), per_rank
        model.edge_segment_capacity = (min(per_rank) + max(per_rank)) // 2        # only the fullest rank overflows
        x_s, oh_s, bits_s = model.denoise_step(T, x, oh, Y32, bits, seed, plan=plans[rank])
        torch.cuda.synchronize()
        try:
            model._finish(oh_s, Y32, bits_s, dense=False)
            raised = False
        except RuntimeError as e:
            raised = "exchange segment" in str(e)
        # the flag was cleared with the report: a roomy segment afterwards gives the single-GPU step again
        model.edge_segment_capacity = None
        model.__dict__["_bufs"] = {}
        got = model.denoise_step(T, x, oh, Y32, bits, seed, plan=plans[rank])
        model._finish(got[1], Y32, got[2], dense=False)
        q.put((rank, raised, _same(want, got), max(per_rank)))
    finally:
        dist.destroy_process_group()


def _run(world, backend, name, worker=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker or _worker, args=(r, world, port, backend, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=160 if worker else 70) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:                                          # never leave a rank behind on the GPU
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok, ok2, edges in res:
        assert ok and ok2, f"rank {rank} of {world} ({backend}, {name}): sharded sample differs from the single-GPU sample" \
            if worker is not _worker_overflow else f"rank {rank}: overflow raised={ok}, clean rerun equal={ok2}"
        assert edges > 0


@pytest.mark.parametrize("world,name", [(2, "sync_mid"), (3, "sync_small"), (2, "async_mid")])
def test_sharded_sample_ranks_share_one_gpu(world, name):
    _run(world, "gloo", name)


@pytest.mark.parametrize("world,name", [(2, "sync_mid:replicated"), (3, "sync_small:replicated"), (2, "async_mid:replicated")])
def test_sharded_sample_with_replicated_graph_part(world, name):
    """ShardPlan(replicate_graph_part=True), the north-star layout (SURVEY.md 8e option 1): every rank runs input
    projections and message passing on all N rows, only mlp_out is row-sharded; collectives = h + X + edge segments.
    Bit-identical to the single-GPU sample, like the stage-sharded default."""
    _run(world, "gloo", name)


@pytest.mark.parametrize("name", ["sync_mid", "async_mid", "sync_mid:replicated"])
def test_sharded_sample_one_rank_rccl(name):
    _run(1, "nccl", name)


def test_edge_segment_overflow_on_one_rank_is_raised_by_every_rank():
    _run(2, "gloo", "sync_mid", worker=_worker_overflow)


def test_sharded_step_amazon_photo_size_two_ranks():
    """BASELINE.json configs[2]: amazon_photo (7650 nodes, 745 attributes) on 1 and 2 GPUs -- here both ranks on one."""
    _run(2, "gloo", "amazon_photo", worker=_worker_full)
