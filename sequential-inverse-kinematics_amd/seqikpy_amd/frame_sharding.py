"""ONE long recording sharded BY FRAME over the GPUs of a node (BASELINE config 3 read literally:
"1M frames x 6 legs, frame-sharded").

The reference walks a recording serially: frame t is warm-started from frame t-1
(``seqikpy/leg_inverse_kinematics.py:272``), so contiguous frame slabs on different ranks are coupled through
one 7-angle state per leg at every slab boundary.  The scheme is the library's frame chunks
(``SeqikOptions.frame_chunk``, include/seqik.h) one level up:

1. every rank solves its slab ``[a_r, b_r)`` in ONE library call with frame chunks (speculation, verification and
   repair of the chunks inside the slab happen on its GPU); ranks > 0 start ``halo`` frames early from the seeds
   (speculation: on well-posed data the solver forgets its start point within a few frames);
2. the ranks exchange their end states (one tiny all-gather: ``world x S x L x 7`` doubles) and each rank compares
   the state its run-in reached with the true end state of its left neighbour;
3. a rank whose boundary disagrees by more than ``tol`` re-solves its slab from the true state (``init_angles``,
   bit-identical to the serial continuation); its own end state may change, so step 2 repeats until no rank
   changed -- at most ``world - 1`` rounds, zero or one on real data;
4. one padded all-gather returns the joint angles (and the FK) of all frames to every rank -- over RCCL/xGMI the
   "final joint-angle gather" of the north star.

No data-path collective runs while the kernels do: the exchange of step 2 is 56 bytes per leg and rank.
The result equals the serial solve to about ``tol`` (exactly, where a boundary had to be repaired).
"""
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .sharding import all_gather_rows, partition


def frame_slab(n_frames: int, world: int, rank: int, chunk: int):
    """[a, b) of ``rank``: whole chunks, balanced over the ranks."""
    n_chunks = -(-n_frames // chunk)
    k0, k1 = partition(n_chunks, world, rank)
    return min(k0 * chunk, n_frames), min(k1 * chunk, n_frames)


def solve_frame_sharded(pose: np.ndarray, legs: List, chunk: int = 32, halo: int = 16, tol: float = 1e-6,
                        want_fk: bool = True, affine=None, device: int = -1, group=None,
                        stats: Optional[Dict] = None):
    """``pose`` (S, L, N, 5, 3) -- the same array on every rank, of which a rank only touches its slab and the
    ``halo`` frames in front of it -- -> dict(angles (S, L, N, 7), fk (S, L, N, 9, 3) or None) on every rank.
    Needs an initialised ``torch.distributed`` process group ("nccl" = RCCL on the GPUs, "gloo" in tests)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    S, L, N = pose.shape[:3]
    a, b = frame_slab(N, world, rank, chunk)
    lead = min(halo, a) if rank > 0 else 0
    dev = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    # chunks inside the slab: the library's frame chunks; tol = 0 asks for exactness there too
    chunk_kw = dict(frame_chunk=int(chunk), frame_halo=max(int(halo), 1), chunk_tol=float(tol) if tol > 0 else -1.0)

    def local_solve(init):
        """The slab from the seeds with a run-in of ``lead`` frames (init None), or from the true state of the left
        neighbour.  Returns the slab's results and the state the run-in reached just before frame a."""
        st = {}
        if b <= a:
            return dict(angles=np.zeros((S, L, 0, 7)), fk=np.zeros((S, L, 0, 9, 3)) if want_fk else None), st
        first = a - lead if init is None else a
        res = _lib.solve_seq(np.ascontiguousarray(pose[:, :, first:b]), legs, want_fk=want_fk, affine=affine, device=device,
                             init_angles=init, **chunk_kw)
        off = a - first
        st["start_state0"] = res["angles"][:, :, off - 1].copy() if off > 0 else None
        st["chunk_stats"] = res.get("chunk_stats")
        return dict(angles=res["angles"][:, :, off:], fk=res["fk"][:, :, off:] if want_fk else None), st

    out, st = local_solve(None)
    start_state = st.get("start_state0")          # what the run-in reached just before frame a (None on rank 0)
    rounds, resolved = 0, 0
    left = None
    while world > 1:
        # end state of every rank (an empty slab hands its left neighbour's state through)
        mine = out["angles"][:, :, -1] if b > a else (left if left is not None else np.zeros((S, L, 7)))
        ends = [torch.empty((S, L, 7), dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(ends, torch.from_numpy(np.ascontiguousarray(mine)).to(dev), group=group)
        changed = 0
        if rank > 0 and b > a:
            left = ends[rank - 1].cpu().numpy()
            bad = start_state is None or np.abs(start_state - left).max() > tol
            if bad:
                out, _ = local_solve(left)
                start_state = left                # from now on this slab starts from the true state
                changed, resolved = 1, resolved + 1
        elif rank > 0:
            left = ends[rank - 1].cpu().numpy()
        flag = torch.tensor([changed], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        if int(flag.item()) == 0:
            break
        rounds += 1
        if rounds > world:  # cannot happen: a repaired boundary is exact, repairs only move rightwards
            raise RuntimeError("frame sharding did not converge")

    counts = [frame_slab(N, world, r, chunk)[1] - frame_slab(N, world, r, chunk)[0] for r in range(world)]

    def gather(x, tail):  # (S, L, n_r, ...) -> (S, L, N, ...)
        t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 2, 0))).to(dev)
        full = all_gather_rows(t, counts, group) if world > 1 else t
        return np.ascontiguousarray(np.moveaxis(full.cpu().numpy(), 0, 2))

    angles = gather(out["angles"], (7,))
    fk = gather(out["fk"], (9, 3)) if want_fk else None
    if stats is not None:
        stats.update(slab=(a, b), lead=lead, boundary_rounds=rounds, slab_resolved=resolved,
                     local_chunk_stats=st.get("chunk_stats"))
    return dict(angles=angles, fk=fk)
