"""Single-node multi-GPU decomposition of one denoising step (one process per GPU, torch.distributed over RCCL).

The reference is single-device; this is the sharding the north star asks for:
  * every rank holds the full state (bitmap of E_t, attributes, labels) and runs the graph part of the towers;
  * the row-independent output MLPs of the towers are row-sharded (contiguous node blocks) and the results are
    replicated with ONE all-gather each (node embeddings h: N x 128 fp32; new attributes: N x F uint8);
  * the upper-triangular pair matrix is sharded by FOLDED row chunks: with 2G chunks of `a` rows, rank r owns chunks
    r and 2G-1-r, so every rank gets the same number of rows AND (to within one chunk's rounding) the same number
    of pairs (row i has n-1-i pairs, so rows i and n-1-i together always have n-1);
  * the sampled edges are gathered back as compact per-rank edge lists (i<j, sorted): ONE equal-sized all-gather of
    segments [ei | ej | count] whose capacity follows the edge marginal, and every rank rebuilds the symmetric bitmap
    (2 MB of edges instead of 24.8 MB of bitmap rows at amazon_computer, 1.6 MB instead of 1.25 GB at SBM-100k).
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
    # How the GNN towers are split (the pair matrix, the attribute draw and the edge exchange are the same either way):
    #   False: every stage row-sharded; the [h_X | h_Y] rows (N x 576 fp32) are all-gathered before each message-passing
    #          layer -- least compute per rank (1/G of everything), 2 extra all-gathers per tower and step;
    #   True:  the north-star / SURVEY 8e layout: every rank runs the input projections and the message-passing layers on
    #          ALL N rows (redundantly, no exchange), only the row-independent output MLP (mlp_out, ~60 % of a tower's
    #          FLOP) is row-sharded; collectives per step = h + X + edge segments only.
    # Results are bit-identical (same kernels on the same rows); which is faster depends on the all-gather latency of the
    # machine (DESIGN.md 7 has both variants' per-rank compute and bytes).
    replicate_graph_part: bool = False

    @staticmethod
    def make(n, world, rank, align=64, replicate_graph_part=False):
        a = _ceil_div(_ceil_div(n, 2 * world), align) * align
        return ShardPlan(n=n, world=world, rank=rank, chunk_rows=a, node_rows=_ceil_div(n, world),
                         replicate_graph_part=replicate_graph_part)

    def adjacency_rows(self):
        """The rows of A_t a rank aggregates: its node block, or all of them when the graph part is replicated."""
        return None if self.replicate_graph_part else self.node_range()

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

    @property
    def padded_pair_rows(self):
        """2G chunks of chunk_rows rows: the row count of the upper-triangle bitmap a rank's pair kernel writes into
        (>= n; the rows past n are never written and stay zero)."""
        return 2 * self.world * self.chunk_rows

    def pairs_of(self, rank):
        n = self.n
        tot = 0
        for b, e in self.pair_chunks(rank):
            tot += sum(n - 1 - i for i in (b, e - 1)) * (e - b) // 2 if e > b else 0
        return tot


class _Done:
    def wait(self):
        pass


def _is_rccl(group):
    return dist.get_backend(group) == "nccl"


def _all_gather(out, inp, group, async_op=False):
    """One equal-sized all-gather.  RCCL (the product transport): stream-ordered, may be left in flight (async_op).
    Any other backend (gloo: the CPU tests, and GPU tests whose ranks share one device) is completed here, and device
    tensors are staged through host copies made HERE: gloo's own handling of device tensors hung on this stack (ROCm 7,
    several ranks on one GPU), with one collective in flight as well as with two."""
    if _is_rccl(group):
        work = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
        return work if async_op else _Done()
    if out.is_cuda:
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    return _Done()


def all_gather_node_rows(plan, full, group=None):
    """`full` is an (n_pad, cols) tensor with n_pad = world*node_rows whose rows node_range(rank) are valid on this
    rank; after the call every rank holds all rows.  One equal-sized all-gather, in place."""
    r0 = plan.rank * plan.node_rows
    _all_gather(full, full[r0:r0 + plan.node_rows], group)
    return full


def start_all_gather_node_rows(plan, full, group=None):
    """all_gather_node_rows, started but not waited for: returns the Work handle.  The collective runs on the backend's
    own stream after everything already enqueued on the current stream; `handle.wait()` makes the current stream wait
    for it (no host block under RCCL), so kernels enqueued in between overlap with the transfer."""
    r0 = plan.rank * plan.node_rows
    return _all_gather(full, full[r0:r0 + plan.node_rows], group, async_op=True)


SEG_HEADER = 16          # int32 slots after the two index arrays of a segment: [count, 15 x padding]


def edge_segment_capacity(plan, density, slack=4.0, floor=65536):
    """Edges per rank an exchange segment holds: `slack` x the rank's share of the edges a graph of the given density has,
    at least `floor`, never more than the pairs a rank owns (then it cannot overflow).  Same value on every rank."""
    n, w = plan.n, plan.world
    expected = float(density) * n * n / 2.0
    want = int(max(slack * expected, 8.0 * n, floor) / w * 1.25) + 1
    most = max(plan.pairs_of(r) for r in range(w))
    return max(min(want, most), 1)


def gather_edge_segments(plan, segment, recv, group=None):
    """One equal-sized all-gather of every rank's edge segment (2*cap + SEG_HEADER int32) into recv (world, same)."""
    _all_gather(recv.view(-1), segment.view(-1), group)
    return recv
