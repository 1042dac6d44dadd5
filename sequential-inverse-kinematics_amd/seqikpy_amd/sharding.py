"""Multi-GPU plumbing: one process per GPU, sequences partitioned across ranks, no exchange
during the solve, one all-gather of the joint-angle blocks at the end (``torch.distributed``;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The unit of sharding is the *sequence* (an independent recording): chains (sequence x leg) never
talk to each other, so the solve itself needs no collective."""
from typing import Callable, List, Tuple

import numpy as np


def partition(n_units: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) slice of ``n_units`` for ``rank``."""
    base, rem = divmod(n_units, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank_share(n_seq: int, world_size: int, rank: int, scaling: str) -> Tuple[int, int, int]:
    """Sequences a rank solves in the benchmark's two scaling modes: ``(start, stop, total)``.

    ``"weak"``: every rank has ``n_seq`` sequences of its own (the job grows with the number of GPUs);
    ``"strong"``: the problem is fixed at ``n_seq`` sequences (BASELINE config 3: "1M frames x 6 legs ... 1->8 MI355X
    frame-sharded") and rank r solves the contiguous slice ``partition(n_seq, world_size, r)``."""
    if scaling == "weak":
        return rank * n_seq, (rank + 1) * n_seq, n_seq * world_size
    if scaling == "strong":
        a, b = partition(n_seq, world_size, rank)
        return a, b, n_seq
    raise ValueError("scaling must be 'weak' or 'strong'")


def all_gather_rows(local, counts: List[int], group=None):
    """All-gathers tensors that differ only in dim 0 (``counts[r]`` rows on rank r) into one tensor
    on every rank: a single padded ``all_gather_into_tensor``-style collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    m = max(counts)
    if local.shape[0] < m:
        pad = torch.zeros((m - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous(), group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def solve_sharded(pose_all: np.ndarray, solve_fn: Callable[[np.ndarray], np.ndarray], group=None) -> np.ndarray:
    """Each rank solves its slice of the sequences (dim 0 of ``pose_all``) with ``solve_fn`` and the
    angle blocks are gathered on every rank.  ``solve_fn(pose_slice) -> angles_slice`` (numpy)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [partition(pose_all.shape[0], world, r)[1] - partition(pose_all.shape[0], world, r)[0]
              for r in range(world)]
    a, b = partition(pose_all.shape[0], world, rank)
    local = torch.from_numpy(np.ascontiguousarray(solve_fn(pose_all[a:b])))
    return all_gather_rows(local, counts, group).numpy()


class GatherPipeline:
    """Per-step "final joint-angle gather" to rank 0, overlapped with the next step's kernels.

    ``n_buffers`` angle buffers alternate: ``submit(b, tensor)`` starts an asynchronous gather of buffer ``b``
    (ordered after the work already queued on the current stream), ``wait_buffer(b)`` must be called
    before buffer ``b`` is overwritten again, ``drain()`` at the end.  On the "nccl" (RCCL) backend the
    waits are stream-side; on "gloo" they block the host.  Rank 0 keeps the most recent gather of
    each buffer in ``self.recv[b]`` (list of ``world`` tensors; its own entry IS the submitted tensor).

    The gather is one group of point-to-point transfers (``batch_isend_irecv``: every other rank sends its
    block, rank 0 posts one receive per peer, all xGMI links at once) rather than ``dist.gather``: that
    collective also copies the root's own block into the gather list, a 336 MB device copy queued on the
    root's compute stream in front of its next solver launch -- measured on one rank: 14.5 -> 21.7 ms per
    step, i.e. the root would have held every step of the job back."""

    def __init__(self, dist, world: int, rank: int, like, dst: int = 0, n_buffers: int = 2):
        import torch
        self.dist, self.world, self.rank, self.dst = dist, world, rank, dst
        self.work = [None] * n_buffers
        # gloo cannot gather device tensors: stage through the host (dry runs / tests only)
        self.stage_on_host = like.is_cuda and dist.get_backend() != "nccl"
        proto = like.cpu() if self.stage_on_host else like
        self.recv = [[None if r == dst else torch.empty_like(proto) for r in range(world)] if rank == dst else None
                     for _ in range(n_buffers)]

    def wait_buffer(self, b: int):
        if self.work[b] is not None:
            for w in self.work[b]:
                w.wait()
            self.work[b] = None

    def submit(self, b: int, tensor):
        self.wait_buffer(b)
        if self.stage_on_host:
            tensor = tensor.cpu()  # synchronises with the producing stream
        dist = self.dist
        if self.rank == self.dst:
            self.recv[b][self.dst] = tensor
            ops = [dist.P2POp(dist.irecv, self.recv[b][r], r, tag=b) for r in range(self.world) if r != self.dst]
        else:
            ops = [dist.P2POp(dist.isend, tensor, self.dst, tag=b)]
        self.work[b] = dist.batch_isend_irecv(ops) if ops else None

    def drain(self):
        for b in range(len(self.work)):
            self.wait_buffer(b)
