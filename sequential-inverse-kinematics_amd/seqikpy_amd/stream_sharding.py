"""BASELINE config 5 on N GPUs: ONE recording streamed from pinned host slabs, its frames sharded over the ranks.

Single process (``scripts/stream_config5.py --one-recording``, ``streaming.SeqikStream(carry=True, frame_chunk=-1)``): the
recording goes through the GPU in time slabs; frame 0 of a slab continues the slab before it exactly (device-resident
hand-over of the 7 angles per leg), inside a slab the library's frame chunks run (speculate, verify, repair).

Here rank r owns the contiguous slabs ``[k0_r, k1_r)`` and streams them from its own pinned buffers over its own PCIe
link.  The only coupling between ranks is the state in front of a rank's first frame -- known when the rank to its left
has finished.  So a rank > 0

1. keeps its FIRST slab resident on its GPU (``frame_sharding.DeviceSlab`` with ``frame_lead`` = the run-in frames in
   front of it): all its chunks, the first one too, start from a run-in;
2. streams its remaining slabs behind it (``SeqikStream.set_carry`` hands the first slab's end state over on the device);
3. when everybody is done the ranks all-gather their end states (56 bytes per leg and rank) and every rank > 0 settles its
   first slab with ``chunk_resume = 2``: its first chunk is re-solved from the true state of its left neighbour -- an
   exact continuation, as a carried slab's is -- and the following chunks are verified again.  On well-posed data that
   changes the first chunk only; should it ever change the slab's END state, the rank streams its remaining slabs again
   from the corrected state, and the exchange repeats (at most world - 1 times);
4. downloads the first slab.

No data-path collective: per rank H2D 120 B + D2H 56 B (+ 216 B FK) per leg-frame over its own link.  The result equals
the single-process stream bit for bit unless a repair cascade crosses a rank boundary (module docstring of
``frame_sharding``) or the library's per-chain guard (more than one chunk in eight of a chain failing its first
verification: the chain is walked serially) fires in a rank's FIRST slab in the single-process stream -- that slab is the
only one solved here without the guard (it has a run-in in front of it); the carried slabs behind it are opened with the
automatic geometry and the guard, exactly as the single-process stream opens them.  The alignment constants of the fused prologue are whole-recording order statistics
(``AlignPose.get_fixed_pos`` / ``get_mean_length``); ``align_stats_all_slabs`` computes them on every rank from all RAW
slabs (each rank reads them over its own link, so this pass costs what it costs on one GPU and needs no exchange).

torch is used for device memory and the process group only.
"""
import time
from typing import Callable, Dict, List, Optional

import numpy as np

from . import _lib
from .frame_sharding import DeviceSlab
from .sharding import partition
from .streaming import SeqikStream


def stream_recording_sharded(get_slab: Callable[[int], np.ndarray], get_out: Callable[[int], tuple], n_slabs: int,
                             slab_frames: int, legs: List, affine=None, want_fk: bool = True, n_slots: int = 3,
                             group=None, lead_frames: Optional[Callable[[int, int], np.ndarray]] = None,
                             stats: Optional[Dict] = None):
    """Streams slabs ``[k0_r, k1_r)`` of ONE recording on this rank.

    ``get_slab(k)`` -> RAW (or aligned, when ``affine`` is None) key points of slab k, planar ``(1, L, 5, T, 3)``, pinned;
    ``get_out(k)`` -> ``(angles (1, L, 7, T), fk (1, L, T, 9, 3) or None)`` pinned output buffers of slab k;
    ``lead_frames(k, h)`` -> the last ``h`` frames in front of slab k, planar ``(1, L, 5, h, 3)`` (default: the tail of
    ``get_slab(k - 1)``).  Frame chunks inside a slab: the library's automatic geometry for ``slab_frames`` frames, as the
    single-process stream uses.  Returns this rank's ``(k0, k1)``."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T, L = int(slab_frames), len(legs)
    spans = [partition(n_slabs, world, r) for r in range(world)]
    k0, k1 = spans[rank]
    C, h, K = _lib.frame_chunk_plan(T)
    if K == 0:
        raise ValueError("slabs this short are walked serially; use slabs of at least 48 frames")
    left_of = None
    for r in range(rank - 1, -1, -1):
        if spans[r][1] > spans[r][0]:
            left_of = r
            break
    coll_dev = torch.device("cuda", torch.cuda.current_device()) if (world > 1 and dist.get_backend(group) == "nccl") \
        else torch.device("cpu")
    layout = _lib.planar_layout(T)
    first = None
    t0 = time.perf_counter()
    st = None
    rounds, restreams = 0, 0
    if k1 > k0:
        # automatic geometry (a function of T alone, == (C, h) above), as the single-process stream opens it: the carried
        # slabs then run WITH the library's per-chain guard, like there (advisor, round 3); only this rank's first slab,
        # which has a run-in in front of its chunk 0 (DeviceSlab below), is outside the guard
        st = SeqikStream(legs, 1, T, affine=affine, layout=layout, want_fk=want_fk, n_slots=n_slots, carry=True,
                         frame_chunk=-1)

    def stream_rest(start_state=None):
        """slabs k0 + 1 .. k1 - 1 (k0 .. on a rank without a left neighbour) behind the given state"""
        if start_state is not None:
            torch.cuda.current_stream().synchronize()
            st.set_carry((start_state, 1), on_device=True)
        else:
            st.reset_carry()
        for k in range(k0 + (1 if first is not None else 0), k1):
            a, f = get_out(k)
            st.submit(get_slab(k), a, f)
        st.wait()

    if k1 > k0:
        if left_of is None:
            stream_rest()
        else:
            lead = lead_frames(k0, h) if lead_frames else get_slab(k0 - 1)[:, :, :, T - h:]
            first = DeviceSlab((np.ascontiguousarray(lead), get_slab(k0)), legs, C, h, 1e-6, h, want_fk, affine, planar=True)
            first.speculate()
            end0 = first.end_state()
            stream_rest(end0)

    def my_end():
        if k1 <= k0:
            return torch.zeros((1, L, 7), dtype=torch.float64)
        if first is not None and k1 - k0 == 1:
            return first.end_state()
        return torch.from_numpy(np.ascontiguousarray(get_out(k1 - 1)[0][:, :, :, T - 1]))

    last_left = None
    while world > 1:
        ends = [torch.empty((1, L, 7), dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(ends, my_end().to(coll_dev).contiguous(), group=group)
        changed = 0
        # (a left state that is the one this rank resumed from last round needs no second verification: it would cost a
        # launch and a blocking read of the statistics per exchange round)
        if first is not None and (last_left is None or not torch.equal(last_left, ends[left_of])):
            last_left = ends[left_of].clone()
            before = first.end_state().clone()
            first.resume(ends[left_of], exact=True)
            if not torch.equal(before, first.end_state()):   # the repair ran through the whole first slab: stream the rest again
                changed = 1
                restreams += 1
                stream_rest(first.end_state())
        flag = torch.tensor([changed], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        if int(flag.item()) == 0:
            break
        rounds += 1
        if rounds > world:
            raise RuntimeError("sharded stream did not converge")
    if first is not None:   # the settled first slab goes to the host -- not before its launches are known to be clean
        first.check_faults()
        a, f = get_out(k0)
        torch.from_numpy(a).copy_(first.d_ang[:, :, :, first.lead:])
        if want_fk:
            torch.from_numpy(f).copy_(first.fk())
    if st is not None:
        st.close()
    dt = time.perf_counter() - t0
    if stats is not None:
        units = (k1 - k0) * L * T
        stats.update(slabs=(k0, k1), seconds=dt, leg_frames=units, frames_per_chunk=C, run_in_frames=h,
                     boundary_rounds=rounds, restreams=restreams, h2d_GBps=units * 120 / dt / 1e9 if dt > 0 else 0.0,
                     d2h_GBps=units * (56 + (216 if want_fk else 0)) / dt / 1e9 if dt > 0 else 0.0)
    return k0, k1


def align_stats_all_slabs(get_slab: Callable[[int], np.ndarray], n_slabs: int, slab_frames: int, n_legs: int, ranks: List[int]):
    """Pass 1 of config 5: the order statistics AlignPose needs, from ALL RAW slabs on this rank's GPU (every rank computes
    the same constants; no exchange).  -> (n_legs, 7, len(ranks)) values."""
    with _lib.AlignStats(n_legs, n_slabs * slab_frames) as ast:
        for k in range(n_slabs):
            ast.add(get_slab(k), n_seq=1, n_frames=slab_frames, layout=_lib.planar_layout(slab_frames))
        return ast.finish(ranks)
