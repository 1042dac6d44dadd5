"""ONE long recording sharded BY FRAME over the GPUs of a node (BASELINE config 3 read literally:
"1M frames x 6 legs, frame-sharded").

The reference walks a recording serially: frame t is warm-started from frame t-1
(``seqikpy/leg_inverse_kinematics.py:259-282``, warm start ``:272``), so contiguous frame slabs on different ranks are
coupled through one 7-angle state per leg at every slab boundary.  The scheme is the library's frame chunks
(``SeqikOptions.frame_chunk``, include/seqik.h) with the slab boundaries on the SAME global chunk grid, so the ranks
together do what one GPU does for the whole recording.

Protocol (LOCKSTEP, the default since round 6 -- the ranks run the pieces of ONE chunked call together):

1. every rank hands its slab ``[a_r, b_r)`` (whole chunks) plus the ``lead = min(halo, a_r)`` frames in front of it to
   ``seqik_solve_seq_device`` with ``frame_lead = lead`` and ``chunk_resume = 3``: the speculative pass only -- all chunks of
   the slab, the first one too, start from a run-in; the start states stay in a caller-owned buffer (``chunk_states``);
2. the ranks all-gather the LAST FRAME of their slabs (7 angles: 56 bytes per leg and rank); every rank works out which of
   its chunks are inconsistent now (chunk 0 against its left neighbour's last frame) and the ranks all-gather one byte per
   leg: "my last chunk is inconsistent" / "some chunk of mine is";
3. nothing inconsistent anywhere (the usual case): done.  Otherwise every rank runs ONE {scan, repair} round
   (``chunk_resume = 4``): chunk 0 is verified against the left neighbour's current last frame and held back when that
   neighbour's last chunk is itself inconsistent in this round -- exactly the "ready" rule of a round inside one call.
   Back to 2, at most ``rounds`` (3) times;
4. what is still inconsistent after that (a cascade longer than ``rounds``) is swept slab by slab, left to right
   (``chunk_resume = 5``), each slab once the slab to its left is final -- the serial sweep of one call, cut at the slab edges;
5. one padded all-gather returns the joint angles of all frames to every rank -- over RCCL / xGMI the "final joint-angle
   gather" of the north star.  The forward kinematics can be gathered the same way or stay sharded.

No collective runs while the kernels do.  The ranks together execute the same sequence of solves, round by round, that one
GPU executes for the whole recording with the same (chunk, halo, tol): **the result is that call's, bit for bit, whatever the
number of ranks** -- also where repairs cascade across a rank boundary (tests: boundaries swept across the anipose LF
kinematic-singularity episode at world 2 / 3 / 8 and four chunk geometries against the one-rank call, model on the CPU and HIP
on the GPU: tests/test_distributed_gloo.py::test_frame_sharding_is_independent_of_the_number_of_ranks).

``lockstep=False`` is the round-2 protocol: every slab runs a whole call for itself first (its first chunk unverified), the
ranks exchange END states and every rank > 0 settles its boundary in a ``chunk_resume = 1`` call, repeated while end states
change.  One exchange cheaper when a boundary chunk does need a repair -- but then a chunk behind a boundary is verified after
the exchange instead of in round 1 and can be re-solved from a predecessor state that differs in the last bits: next to the
LF episode three placements in four still give the one-rank call's bits, the others stay within 1e-7 rad of it outside the
episode and differ by up to pi on the episode's own frames (the leg has two configurations there).
The automatic mode's per-chain guard (chains with many inconsistent chunks walked serially) belongs to one-GPU calls;
here the chunk geometry is fixed up front (``_lib.frame_chunk_plan`` of the whole recording) and explicit.

torch is used for device memory and the process group only.
"""
from typing import Callable, Dict, List, Optional

import numpy as np

from . import _lib
from .sharding import partition


def frame_slab(n_frames: int, world: int, rank: int, chunk: int):
    """[a, b) of ``rank``: whole chunks, balanced over the ranks."""
    n_chunks = -(-n_frames // chunk)
    k0, k1 = partition(n_chunks, world, rank)
    return min(k0 * chunk, n_frames), min(k1 * chunk, n_frames)


def chunk_geometry(n_frames: int, chunk: Optional[int] = None, halo: Optional[int] = None):
    """(frames per chunk, run-in frames) of a recording: the given ones, or the library's automatic choice for a
    recording of this length (``seqik_frame_chunk_plan``); a recording too short for chunks is one chunk."""
    if chunk is None or chunk <= 0:
        c, h, _ = _lib.frame_chunk_plan(n_frames, -1, halo or 0)
        if c == 0:
            return max(int(n_frames), 1), max(int(halo or 1), 1)
        return c, h
    return int(chunk), max(int(halo if halo else 8), 1)


class DeviceSlab:
    """A rank's slab on its GPU: key points, angles, FK and chunk start states stay in HBM between the calls."""

    def __init__(self, pose_slab: np.ndarray, legs: List, chunk: int, halo: int, tol: float, lead: int, want_fk: bool,
                 affine=None, device: int = -1, planar: bool = False):
        """``pose_slab`` (S, L, n, 5, 3), or with ``planar`` (S, L, 5, n, 3) (the device layout: no transpose)."""
        import torch
        self.torch = torch
        if isinstance(pose_slab, tuple):
            self.S, self.L = pose_slab[1].shape[:2]
            self.n = pose_slab[0].shape[3] + pose_slab[1].shape[3]
        elif planar:
            self.S, self.L, _, self.n = pose_slab.shape[:4]
        else:
            self.S, self.L, self.n = pose_slab.shape[:3]
        self.legs, self.lead, self.want_fk, self.affine = legs, int(lead), want_fk, affine
        self.kw = dict(frame_chunk=int(chunk), frame_halo=int(halo), chunk_tol=float(tol) if tol > 0 else -1.0)
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None or device < 0 else device)
        self.K = max(_lib.frame_chunk_plan(self.n, int(chunk), int(halo), self.lead)[2], 1)
        # planar device layout (include/seqik.h, SeqikLayout): every key-point row and every joint is its own time series
        self.layout = _lib.planar_layout(self.n)
        with torch.cuda.device(self.dev):
            if isinstance(pose_slab, tuple):    # planar (lead frames, slab frames): uploaded side by side, no host concatenation
                lead_part, body = pose_slab        # (pinned host memory is DMA-copied: torch asks the driver about the pointer)
                self.d_pose = torch.empty((self.S, self.L, 5, self.n, 3), dtype=torch.float64, device=self.dev)
                self.d_pose[:, :, :, :lead_part.shape[3]].copy_(torch.from_numpy(lead_part), non_blocking=True)
                self.d_pose[:, :, :, lead_part.shape[3]:].copy_(torch.from_numpy(body), non_blocking=True)
            else:
                host = pose_slab if planar else pose_slab.transpose(0, 1, 3, 2, 4)
                self.d_pose = torch.from_numpy(np.ascontiguousarray(host)).to(self.dev)                           # [S][L][5][n][3]
            self.d_ang = torch.zeros((self.S, self.L, 7, self.n), dtype=torch.float64, device=self.dev)           # [S][L][7][n]
            self.d_fk = torch.zeros((self.S, self.L, self.n, 9, 3), dtype=torch.float64, device=self.dev) if want_fk else None
            self.d_states = torch.zeros((self.S, self.L, self.K, 7), dtype=torch.float64, device=self.dev)
            self.d_stats = torch.zeros(_lib.N_CHUNK_STATS, dtype=torch.int32, device=self.dev)
            self.d_flags = torch.zeros((self.S, self.L, self.K), dtype=torch.uint8, device=self.dev)
        self.chunked = _lib.frame_chunk_plan(self.n, int(chunk), int(halo), self.lead)[2] > 0
        self.repaired = 0
        self.C, self.tol = int(chunk), (float(tol) if tol > 0 else 0.0)
        # last frame in front of chunk k (k >= 1) of the slab's buffers: lead + k C - 1
        self._before = torch.arange(1, self.K, device=self.dev) * self.C + self.lead - 1

    def _call(self, d_init=0, resume=0, flags=False):
        torch = self.torch
        with torch.cuda.device(self.dev):
            kw = dict(self.kw, frame_lead=self.lead, d_chunk_states=self.d_states.data_ptr(), chunk_resume=resume,
                      d_chunk_stats=self.d_stats.data_ptr(), d_chunk_flags=self.d_flags.data_ptr() if flags else 0) if self.chunked else {}
            _lib.solve_seq_device(self.d_pose.data_ptr(), self.S, self.L, self.n, self.legs, self.d_ang.data_ptr(),
                                  self.d_fk.data_ptr() if self.want_fk else 0, d_init=d_init, affine=self.affine,
                                  layout=self.layout, stream=torch.cuda.current_stream(self.dev).cuda_stream, **kw)

    def speculate(self):
        self.repaired = 0   # statistics describe ONE solve (speculate + its resume calls), not a benchmark loop
        self._call()

    def resume(self, left_state, exact: bool = False):
        """``left_state`` (S, L, 7): the true state of the frame in front of the slab (device tensor).  ``exact``: the
        first chunk must continue that state bit for bit (``chunk_resume = 2``), as a carried slab of a stream does."""
        if not self.chunked:   # (only a slab that IS the whole, short recording is walked serially; it has no left neighbour)
            raise RuntimeError("a serially walked slab cannot be resumed")
        self._left = left_state.to(self.dev).contiguous()
        self._call(d_init=self._left.data_ptr(), resume=2 if exact else 1)
        st = self.d_stats.cpu().numpy()      # blocking read: the speculative pass and this resume call have finished
        self._check_stream_faults()
        self.repaired += int(st[3:7].sum())

    # ---- lockstep pieces of ONE chunked call (SeqikOptions.chunk_resume = 3 / 4 / 5, include/seqik.h) ------------------------------
    def speculate_only(self):
        """The speculative pass alone (no repair round, no sweep): every chunk of the slab from its run-in."""
        self.repaired = 0
        self._call(resume=3 if self.chunked else 0)

    def inc_flags(self, left_state=None):
        """(S, L, K) bool on the GPU: which chunks are inconsistent NOW -- the state chunk k was started from against the current
        last frame in front of it; chunk 0 against `left_state` (S, L, 7), the current last frame of the slab to the left (never
        on the first slab).  The library's own test (chunk_inconsistent in seqik_hip.hip), NaN counting as a mismatch."""
        torch = self.torch
        inc = torch.zeros((self.S, self.L, self.K), dtype=torch.bool, device=self.dev)
        if not self.chunked:
            return inc
        if self.K > 1:
            before = self.d_ang[:, :, :, self._before].permute(0, 1, 3, 2)                      # (S, L, K - 1, 7)
            inc[:, :, 1:] = ~((self.d_states[:, :, 1:] - before).abs() <= self.tol).all(-1)
        if left_state is not None:
            inc[:, :, 0] = ~((self.d_states[:, :, 0] - left_state.to(self.dev)).abs() <= self.tol).all(-1)
        return inc

    def one_round(self, left_state=None, left_blocked=None):
        """ONE {scan, repair} round.  `left_blocked` (S, L) bool: the last chunk of the slab to the left is itself inconsistent in
        this round, so an inconsistent chunk 0 waits (its predecessor is about to change), as inside one call."""
        if not self.chunked:
            return
        if left_state is not None:
            self._left = left_state.to(self.dev).contiguous()
            if left_blocked is not None:
                self.d_flags[:, :, 0] |= left_blocked.to(self.dev).to(self.torch.uint8) * _lib.CHUNK_FLAG_LEFT_BLOCKED
        self._call(d_init=self._left.data_ptr() if left_state is not None else 0, resume=4, flags=True)
        self._after_repairs()

    def sweep_only(self, left_state=None):
        """The final scan + serial sweep; `left_state` must be FINAL (the slab to the left has been swept)."""
        if not self.chunked:
            return
        if left_state is not None:
            self._left = left_state.to(self.dev).contiguous()
        self._call(d_init=self._left.data_ptr() if left_state is not None else 0, resume=5, flags=True)
        self._after_repairs()

    def _after_repairs(self):
        st = self.d_stats.cpu().numpy()      # blocking read: the call has finished (repairs are the rare path)
        self._check_stream_faults()
        self.repaired += int(st[3:7].sum())

    def _check_stream_faults(self):
        with self.torch.cuda.device(self.dev):
            _lib.check_faults(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def check_faults(self):
        """The launches of this slab go through the asynchronous device entry point, which cannot report a watchdog fault
        of its own launch (include/seqik.h, "Device faults"): synchronise the slab's stream and raise if one of them did.
        Called wherever a slab's results leave the GPU (``solve_frame_sharded``, ``stream_recording_sharded``)."""
        self.torch.cuda.current_stream(self.dev).synchronize()
        self._check_stream_faults()

    def end_state(self):
        return self.d_ang[:, :, :, self.n - 1].contiguous()

    def angles(self):
        """(S, L, frames of the slab, 7) view of the planar buffer"""
        return self.d_ang[:, :, :, self.lead:].permute(0, 1, 3, 2)

    def fk(self):
        return self.d_fk[:, :, self.lead:] if self.want_fk else None

    def chunk_stats(self):
        return _lib.chunk_stats_dict(self.d_stats.cpu().numpy())


class FrameShardedRecording:
    """One recording (or S recordings advancing together) sharded by frame over the ranks of ``group``.

    ``pose`` (S, L, N, 5, 3) -- the same array on every rank, of which a rank only touches its slab and the run-in frames
    in front of it.  ``solve()`` runs steps 1-4 of the module docstring on the data resident on the GPUs and may be
    called repeatedly (benchmark steps).  ``slab_factory``: test hook (an oracle-built slab on the CPU)."""

    def __init__(self, pose: np.ndarray, legs: List, chunk: Optional[int] = None, halo: Optional[int] = None,
                 tol: float = 1e-6, want_fk: bool = True, affine=None, device: int = -1, group=None,
                 slab_factory: Optional[Callable] = None, lockstep: bool = True, rounds: int = 3):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.S, self.L, self.N = pose.shape[:3]
        self.C, self.h = chunk_geometry(self.N, chunk, halo)
        self.tol, self.want_fk = tol, want_fk
        self.slabs = [frame_slab(self.N, self.world, r, self.C) for r in range(self.world)]
        self.a, self.b = self.slabs[self.rank]
        self.lead = min(self.h, self.a) if self.rank > 0 else 0
        # the nearest rank to the left that owns frames (slabs can be empty when there are fewer chunks than ranks)
        self.left_of = None
        for r in range(self.rank - 1, -1, -1):
            if self.slabs[r][1] > self.slabs[r][0]:
                self.left_of = r
                break
        self.slab = None
        if self.b > self.a:
            make = slab_factory or DeviceSlab
            self.slab = make(np.ascontiguousarray(pose[:, :, self.a - self.lead:self.b]), legs, self.C, self.h, tol, self.lead,
                             want_fk, affine, device)
        on_gpu = self.world > 1 and dist.get_backend(group) == "nccl"
        self.coll_dev = (self.slab.dev if (self.slab is not None and hasattr(self.slab, "dev")) else
                         torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
        self.stats: Dict = {}
        #: True (default since round 6): the ranks run the rounds of ONE chunked call together (`_solve_lockstep`): the result is the
        #: one-GPU call's bit for bit whatever the number of ranks.  False: the round-2 protocol (every slab settles itself, then
        #: its boundary in resume calls): one exchange fewer when nothing needs a repair, but next to a singular episode the bits
        #: can depend on where the rank boundaries fall (module docstring)
        self.lockstep, self.rounds = bool(lockstep), int(rounds)
        #: set to a list to have solve() append a (start, end) pair of HIP events around each speculative pass
        self.spec_events: Optional[list] = None

    def _gather_ends(self):
        torch, dist = self.torch, self.dist
        mine = (self.slab.end_state() if self.slab is not None else
                torch.zeros((self.S, self.L, 7), dtype=torch.float64)).to(self.coll_dev).contiguous()
        ends = [torch.empty((self.S, self.L, 7), dtype=torch.float64, device=self.coll_dev) for _ in range(self.world)]
        dist.all_gather(ends, mine, group=self.group)
        return ends

    def _speculate(self, only: bool):
        torch = self.torch
        if self.slab is None:
            return
        run = self.slab.speculate_only if only else self.slab.speculate
        if self.spec_events is not None and hasattr(self.slab, "dev"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            self.spec_events.append((e0, e1))
        else:
            run()

    def _exchange(self):
        """-> (this rank's left neighbour's current last frame or None, (S, L) bool "its last chunk is inconsistent now", this
        rank's own (S, L, K) inconsistency flags, "some chunk of some rank is inconsistent").  Two small all-gathers: the last
        frames (56 B per leg and rank), then what every rank makes of them (one byte per leg and rank)."""
        torch, dist = self.torch, self.dist
        ends = self._gather_ends()
        left = ends[self.left_of] if (self.slab is not None and self.left_of is not None) else None
        inc = self.slab.inc_flags(left) if self.slab is not None else None
        mine = torch.zeros((self.S, self.L), dtype=torch.uint8)
        if inc is not None:
            mine = inc[:, :, -1].to(torch.uint8) | (inc.any(-1).to(torch.uint8) << 1)    # bit 0: my last chunk, bit 1: any chunk of mine
        mine = mine.to(self.coll_dev).contiguous()
        flags = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(flags, mine, group=self.group)
        any_inc = any(bool((f & 2).any().item()) for f in flags)
        blocked = (flags[self.left_of] & 1).bool() if left is not None else None
        return left, blocked, any_inc

    def _solve_lockstep(self):
        """The rounds of ONE chunked call, run by all ranks together (SeqikOptions.chunk_resume = 3 / 4 / 5): speculative pass ->
        up to `rounds` x {exchange, one {scan, repair} round on every slab} while anything is inconsistent anywhere -> exchange
        -> if something still is: sweep slab by slab, left to right.  == tests/chunk_model.py::lockstep_sharded_oracle."""
        dist = self.dist
        self._speculate(only=True)
        done, calls, swept = 0, 0, False
        for _ in range(self.rounds):
            left, blocked, any_inc = self._exchange()
            if not any_inc:
                break
            if self.slab is not None:
                self.slab.one_round(left, blocked)
                calls += 1
            done += 1
        else:
            left, blocked, any_inc = self._exchange()
        if any_inc:   # rare: a cascade longer than `rounds`; a slab may be swept once the slab to its left is final
            swept = True
            for r in range(self.world):
                if r == self.rank and self.slab is not None:
                    self.slab.sweep_only(left)
                    calls += 1
                if r + 1 < self.world:
                    ends = self._gather_ends()      # (every rank takes part; only the right neighbour of r uses r's new end)
                    if self.slab is not None and self.left_of == r:
                        left = ends[r]
        self.stats = dict(slab=(self.a, self.b), lead=self.lead, chunk=self.C, halo=self.h, boundary_rounds=done, resume_calls=calls,
                          swept=swept, protocol="lockstep",
                          chunks_repaired_after_exchange=getattr(self.slab, "repaired", 0) if self.slab is not None else 0)

    def solve(self, gather_fk: Optional[bool] = None):
        """-> dict(angles (S, L, N, 7), fk (S, L, N, 9, 3) or None) as tensors on every rank (on the GPU under RCCL)."""
        torch, dist = self.torch, self.dist
        if self.lockstep and self.world > 1:
            self._solve_lockstep()
            if gather_fk is None:
                gather_fk = self.want_fk
            return dict(angles=self._gather(lambda s: s.angles(), (7,)),
                        fk=self._gather(lambda s: s.fk(), (9, 3)) if (self.want_fk and gather_fk) else None)
        if self.slab is not None:
            if self.spec_events is not None and hasattr(self.slab, "dev"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.slab.speculate()
                e1.record()
                self.spec_events.append((e0, e1))
            else:
                self.slab.speculate()
        rounds, resumed = 0, 0
        if self.world > 1:
            left_prev = None
            while True:
                ends = self._gather_ends()
                changed = 0
                if self.slab is not None and self.left_of is not None:
                    left = ends[self.left_of]
                    if left_prev is None or not torch.equal(left, left_prev):
                        before = self.slab.end_state().clone()
                        self.slab.resume(left)
                        resumed += 1
                        left_prev = left.clone()
                        changed = int(not torch.equal(before, self.slab.end_state()))
                flag = torch.tensor([changed], dtype=torch.int64, device=self.coll_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group)
                if int(flag.item()) == 0:
                    break
                rounds += 1
                if rounds > self.world:  # cannot happen: a settled boundary is exact, changes only travel rightwards
                    raise RuntimeError("frame sharding did not converge")
        self.stats = dict(slab=(self.a, self.b), lead=self.lead, chunk=self.C, halo=self.h, boundary_rounds=rounds,
                          resume_calls=resumed,
                          chunks_repaired_after_exchange=getattr(self.slab, "repaired", 0) if self.slab is not None else 0)
        if gather_fk is None:
            gather_fk = self.want_fk
        return dict(angles=self._gather(lambda s: s.angles(), (7,)),
                    fk=self._gather(lambda s: s.fk(), (9, 3)) if (self.want_fk and gather_fk) else None)

    def check_faults(self):
        """Synchronises this rank's slab and raises ``SeqikLibraryError`` if one of its launches reported a kernel fault
        (``solve()`` is asynchronous on the GPU; call this before trusting what it returned)."""
        if self.slab is not None and hasattr(self.slab, "check_faults"):
            self.slab.check_faults()

    def _gather(self, get, tail):
        """(S, L, n_r, ...) per rank -> (S, L, N, ...) on every rank: one padded all-gather along the frame axis."""
        torch, dist = self.torch, self.dist
        if self.slab is not None:
            x = get(self.slab)
        else:
            x = torch.zeros((self.S, self.L, 0) + tail, dtype=torch.float64)
        if self.world == 1:
            return x
        counts = [b - a for a, b in self.slabs]
        m = max(counts)
        buf = torch.zeros((m, self.S, self.L) + tail, dtype=torch.float64, device=self.coll_dev)
        buf[:x.shape[2]] = x.to(self.coll_dev).movedim(2, 0)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf, group=self.group)
        full = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
        return full.movedim(0, 2).contiguous()


def solve_frame_sharded(pose: np.ndarray, legs: List, chunk: Optional[int] = None, halo: Optional[int] = None,
                        tol: float = 1e-6, want_fk: bool = True, affine=None, device: int = -1, group=None,
                        stats: Optional[Dict] = None, slab_factory: Optional[Callable] = None, lockstep: bool = True):
    """``pose`` (S, L, N, 5, 3) -> dict(angles (S, L, N, 7), fk (S, L, N, 9, 3) or None) as numpy arrays on every rank.
    ``chunk`` / ``halo`` None: the library's automatic geometry for a recording of N frames.  Needs an initialised
    ``torch.distributed`` process group when there is more than one rank ("nccl" = RCCL on the GPUs, "gloo" in tests)."""
    rec = FrameShardedRecording(pose, legs, chunk, halo, tol, want_fk, affine, device, group, slab_factory, lockstep=lockstep)
    out = rec.solve()
    rec.check_faults()
    if stats is not None:
        stats.update(rec.stats)
        if rec.slab is not None and hasattr(rec.slab, "chunk_stats"):
            stats["local_chunk_stats"] = rec.slab.chunk_stats()
    return dict(angles=out["angles"].cpu().numpy(), fk=out["fk"].cpu().numpy() if out["fk"] is not None else None)
