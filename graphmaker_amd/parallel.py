"""Single-node multi-GPU decomposition of one denoising step (one process per GPU, torch.distributed over RCCL).

The reference is single-device; this is the sharding the north star asks for:
  * every rank holds the full state (bitmap of E_t, attributes, labels) and runs the graph part of the towers;
  * the row-independent output MLPs of the towers are row-sharded (contiguous node blocks) and the results are
    replicated with ONE all-gather each (node embeddings h: N x 128 fp32; new attributes: N x F uint8);
  * the upper-triangular pair matrix is sharded by FOLDED row chunks: with 2G chunks of `a` rows, rank r owns chunks
    r and 2G-1-r, so every rank gets the same number of rows AND (to within one chunk's rounding) the same number
    of pairs (row i has n-1-i pairs, so rows i and n-1-i together always have n-1);
  * the sampled upper-triangle rows are gathered back with one equal-sized all-gather and every rank symmetrises.
Draws are keyed on global (i, j) / (f, v) counters, so the result is bit-identical for any number of ranks.

Only torch.distributed calls and tensor slicing live here (device-agnostic, so the logic runs under gloo on CPU in
tests/test_parallel_cpu.py); the compute is passed in by the caller.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


def _ceil_div(a, b):
    return -(-a // b)


@dataclass
class ShardPlan:
    n: int
    world: int
    rank: int
    chunk_rows: int          # a: rows per folded chunk (multiple of 64 = the pair kernel's tile edge)
    node_rows: int           # rows per rank of the contiguous node partition (towers, attributes)

    @staticmethod
    def make(n, world, rank, align=64):
        a = _ceil_div(_ceil_div(n, 2 * world), align) * align
        return ShardPlan(n=n, world=world, rank=rank, chunk_rows=a, node_rows=_ceil_div(n, world))

    def chunk(self, c):
        a, n = self.chunk_rows, self.n
        return min(c * a, n), min((c + 1) * a, n)

    def pair_chunks(self, rank=None):
        """The two row ranges of the pair matrix rank `rank` owns."""
        r = self.rank if rank is None else rank
        return [self.chunk(r), self.chunk(2 * self.world - 1 - r)]

    def node_range(self, rank=None):
        r = self.rank if rank is None else rank
        return min(r * self.node_rows, self.n), min((r + 1) * self.node_rows, self.n)

    def pairs_of(self, rank):
        n = self.n
        tot = 0
        for b, e in self.pair_chunks(rank):
            tot += sum(n - 1 - i for i in (b, e - 1)) * (e - b) // 2 if e > b else 0
        return tot


def all_gather_node_rows(plan, full, group=None):
    """`full` is an (n_pad, cols) tensor with n_pad = world*node_rows whose rows node_range(rank) are valid on this
    rank; after the call every rank holds all rows.  One equal-sized all-gather, in place."""
    r0 = plan.rank * plan.node_rows
    local = full[r0:r0 + plan.node_rows]
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full


_scratch = {}


def _persistent(name, shape, dtype, device):
    """Buffers a collective touches are kept for the life of the process (the collective's stream holds references;
    re-allocating them every step makes the caching allocator grow and stall)."""
    key = (name, tuple(shape), dtype, device)
    if key not in _scratch:
        _scratch[key] = torch.zeros(shape, dtype=dtype, device=device)
    return _scratch[key]


def gather_upper_rows(plan, upper, group=None):
    """`upper` is the (n, wpr) int32 upper-triangle bitmap whose rows pair_chunks(rank) were written on this rank.
    After the call every rank holds every row.  Pack (2a rows) -> one all-gather -> unpack."""
    a, w = plan.chunk_rows, plan.world
    wpr = upper.shape[1]
    send = _persistent("send", (2 * a, wpr), upper.dtype, upper.device)   # rows beyond a short chunk stay zero
    for k, (b, e) in enumerate(plan.pair_chunks()):
        if e > b:
            send[k * a:k * a + (e - b)] = upper[b:e]
    recv = _persistent("recv", (w * 2 * a, wpr), upper.dtype, upper.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(w, 2 * a, wpr)
    for r in range(w):
        if r == plan.rank:
            continue
        for k, (b, e) in enumerate(plan.pair_chunks(r)):
            if e > b:
                upper[b:e] = recv[r, k * a:k * a + (e - b)]
    return upper
