"""Multi-rank runs of the sharded sample() with the real HIP kernels on ONE GPU (GPU box).

RCCL refuses two ranks on one device, so the ranks of these tests share cuda:0 and exchange their shards over the
`gloo` backend (device tensors staged through the host by torch); the product code path -- ShardPlan, the alternating
tower stages with started-then-waited all-gathers, the one-launch folded pair chunks, the chunk-wise gather of the
sampled rows -- is the one the RCCL ranks of bench.py run.  Every rank must reproduce the single-GPU sample bit for bit.
A one-rank RCCL group covers the NCCL stream hand-over of the same path.
"""
import os
import socket

import numpy as np
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
        cfg, model = _model(name, dev)
        plan = ShardPlan.make(cfg["N"], world, rank)
        want = model.sample(seed=cfg["seed"])
        got = model.sample(seed=cfg["seed"], plan=plan)
        again = model.sample(seed=cfg["seed"], plan=plan)          # persistent gather buffers are reused correctly
        torch.cuda.synchronize()
        q.put((rank, _same(want, got), _same(want, again), int(want[2].sum().item())))
    finally:
        dist.destroy_process_group()


def _run(world, backend, name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=70) for _ in range(world)]
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
        assert ok and ok2, f"rank {rank} of {world} ({backend}, {name}): sharded sample differs from the single-GPU sample"
        assert edges > 0


@pytest.mark.parametrize("world,name", [(2, "sync_mid"), (3, "sync_small"), (2, "async_mid")])
def test_sharded_sample_ranks_share_one_gpu(world, name):
    _run(world, "gloo", name)


@pytest.mark.parametrize("name", ["sync_mid", "async_mid"])
def test_sharded_sample_one_rank_rccl(name):
    _run(1, "nccl", name)
