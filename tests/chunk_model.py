"""TEST INFRASTRUCTURE -- model of the frame-chunk mode (``SeqikOptions.frame_chunk``, include/seqik.h, ABI 3) built on the
C oracle: the same sequence of solves the library enqueues (speculative pass, first verification with the per-chain
guard of the automatic mode, ``rounds`` x {scan, repair}, scan + serial sweep, serial walk of the chains the guard gave
up on), every solve being an oracle run over the frames of a chunk.  The HIP path must reproduce it bit for bit,
statistics and per-chunk report included.  Also models a SLAB of a longer recording (``frame_lead``) and the
``chunk_resume`` call that settles its first chunk once the true state in front of it is known (frame sharding over ranks).
"""
import numpy as np

FAILED_FIRST, REPAIRED, SWEPT, SERIAL = 1, 2, 4, 8


def plan(n_frames, frame_chunk=-1, frame_halo=0, frame_lead=0):
    """Restatement of pick_frame_chunks (csrc/seqik_hip.hip): (C, h, K) or (0, 0, 0) when the call is walked serially."""
    if frame_chunk == 0 or frame_lead >= n_frames:
        return 0, 0, 0
    n = n_frames - frame_lead
    h = frame_halo if frame_halo > 0 else 8
    c = frame_chunk
    if c < 0:
        if n < 48:
            return 0, 0, 0
        c = ((6 * n // 196608 + 7) // 8) * 8
        c = min(max(c, 8), 64)
        if c == 8 and 6 * ((n + 7) // 8) <= 1024:
            c = 4
            if frame_halo <= 0:
                h = 4
    if c >= n and frame_lead == 0:
        return 0, 0, 0
    return int(c), int(h), int(-(-n // c))


class ChunkedChain:
    """One chain of a chunked call.  ``speculate()`` then ``settle()`` = one library call; a later ``settle(init=...,
    resume=True)`` = a ``chunk_resume`` call."""

    def __init__(self, oracle, pose, seg, bounds, seeds, chunk, halo, tol=1e-6, rounds=3, init=None, guard=False, lead=0):
        self.o, self.pose, self.par = oracle, pose, (seg, bounds, seeds)
        self.N, self.C, self.h, self.lead = pose.shape[0], int(chunk), int(halo), int(lead)
        self.K = -(-(self.N - self.lead) // self.C)
        self.tol, self.rounds, self.init, self.guard = tol, rounds, init, guard and lead == 0
        self.angles, self.fk = np.zeros((self.N, 7)), np.zeros((self.N, 9, 3))
        self.ss = np.zeros((self.K, 7))
        self.flags = np.zeros(self.K, np.uint8)
        self.stats = np.zeros(16, np.int32)
        self.serial = False

    def span(self, k):
        return self.lead + k * self.C, min(self.lead + (k + 1) * self.C, self.N)

    def _solve(self, k, repair, flag=0):
        a, b = self.span(k)
        seg, bounds, seeds = self.par
        if repair:
            start = self.init if k == 0 else self.angles[a - 1]
            self.ss[k] = start
            r = self.o.seq_leg(self.pose[a:b], seg, bounds, seeds, init=np.array(start, dtype=np.float64).copy())
            off = 0
            self.flags[k] |= flag
        else:
            run_in = k > 0 or self.lead > 0
            t0 = a - self.h if (k > 0 and a > self.h) else 0
            r = self.o.seq_leg(self.pose[t0:b], seg, bounds, seeds, init=None if run_in else self.init)
            off = a - t0
            if run_in:
                self.ss[k] = r["angles"][off - 1]
        self.angles[a:b] = r["angles"][off:]
        self.fk[a:b] = r["fk"][off:]

    def speculate(self):
        for k in range(self.K):
            self._solve(k, False)

    def _inconsistent(self, k):
        truth = self.init if k == 0 else self.angles[self.lead + k * self.C - 1]
        return not np.all(np.abs(self.ss[k] - truth) <= self.tol)

    def settle(self, init=None, resume=False):
        if init is not None:
            self.init = init
        self.stats[:] = 0
        self.stats[:3] = (self.K, self.C, self.h)
        k_first = 0 if ((self.lead > 0 or resume) and self.init is not None) else 1
        fails = [k for k in range(k_first, self.K) if self._inconsistent(k)]
        if not resume:
            self.flags[:] = 0
        self.flags[fails] |= FAILED_FIRST
        self.stats[7] = len(fails)
        if self.guard and not resume and len(fails) * 8 > self.K:
            self.serial = True
            self.flags |= SERIAL
            self.stats[8], self.stats[9] = 1, self.K
            seg, bounds, seeds = self.par
            r = self.o.seq_leg(self.pose, seg, bounds, seeds, init=self.init)
            self.angles[:], self.fk[:] = r["angles"], r["fk"]
            return self
        for r in range(self.rounds + 1):
            inc = [k for k in range(k_first, self.K) if self._inconsistent(k)]
            if not inc:
                break
            if r < self.rounds:
                ready = [k for k in inc if not (k > k_first and (k - 1) in inc)]
                self.stats[3 + min(r, 2)] += len(ready)
                for k in ready:
                    self._solve(k, True, REPAIRED)
            else:
                for k in range(k_first, self.K):
                    if self._inconsistent(k):
                        self._solve(k, True, SWEPT)
                        self.stats[6] += 1
        return self


    # ---- lockstep pieces of one call (chunk_resume = 3 / 4 / 5): a slab of a recording whose other slabs live on other GPUs ----
    def k_first(self):
        return 0 if self.init is not None else 1

    def inc_flags(self, init=None):
        """bool[K]: which chunks are inconsistent now (chunk 0 against `init`, the current last frame of the slab to the left;
        never on the first slab)."""
        if init is not None:
            self.init = init
        return np.array([k >= self.k_first() and self._inconsistent(k) for k in range(self.K)])

    def speculate_only(self):
        """chunk_resume = 3: the speculative pass (and the first verification's statistics, which the result does not need)."""
        self.speculate()
        self.flags[:] = 0
        return self

    def one_round(self, init=None, left_blocked=False):
        """chunk_resume = 4: ONE {scan, repair} round; chunk 0 is held back when the last chunk of the slab to the left is itself
        inconsistent in this round (its end state is about to change)."""
        inc = self.inc_flags(init)
        kf = self.k_first()
        ready = [k for k in range(kf, self.K) if inc[k] and not (k > kf and inc[k - 1])]
        if left_blocked and kf == 0 and 0 in ready:
            ready.remove(0)
        for k in ready:
            self._solve(k, True, REPAIRED)
        return len(ready)

    def sweep_only(self, init=None):
        """chunk_resume = 5: left to right, every chunk that is still inconsistent is re-solved from the final state in front of it."""
        if init is not None:
            self.init = init
        n = 0
        for k in range(self.k_first(), self.K):
            if self._inconsistent(k):
                self._solve(k, True, SWEPT)
                n += 1
        return n


def lockstep_sharded_oracle(oracle, pose, seg, bounds, seeds, chunk, halo, world, tol=1e-6, rounds=3):
    """One chain of ONE recording sharded by frame over `world` ranks, LOCKSTEP protocol of seqikpy_amd.frame_sharding (round 6):
    every slab speculates; then, as long as some chunk anywhere is inconsistent and fewer than `rounds` rounds have run, the
    slabs exchange (last frame, "my last chunk is inconsistent") and run ONE {scan, repair} round each; what is still
    inconsistent after that is swept slab by slab, left to right.  By construction the sequence of solves one GPU runs for the
    whole recording -> the one-rank result bit for bit.  -> dict(angles, fk, rounds, swept)."""
    N = pose.shape[0]
    n_chunks = -(-N // chunk)

    def part(r):
        base, rem = divmod(n_chunks, world)
        a = r * base + min(r, rem)
        return a, a + base + (1 if r < rem else 0)
    slabs = [(min(part(r)[0] * chunk, N), min(part(r)[1] * chunk, N)) for r in range(world)]
    owners = [r for r, (a, b) in enumerate(slabs) if b > a]
    models = {}
    for i, r in enumerate(owners):
        a, b = slabs[r]
        lead = min(halo, a) if i > 0 else 0
        models[r] = (ChunkedChain(oracle, pose[a - lead:b], seg, bounds, seeds, chunk, halo, tol=tol, lead=lead).speculate_only(), lead)
    left_of = {r: (owners[i - 1] if i > 0 else None) for i, r in enumerate(owners)}
    done_rounds, swept = 0, 0

    def exchange():
        ends = {r: models[r][0].angles[-1].copy() for r in owners}
        incs = {r: models[r][0].inc_flags(ends[left_of[r]].copy() if left_of[r] is not None else None) for r in owners}
        return ends, incs
    for _ in range(rounds):
        ends, incs = exchange()
        if not any(v.any() for v in incs.values()):
            break
        for r in owners:
            lf = left_of[r]
            models[r][0].one_round(ends[lf].copy() if lf is not None else None, left_blocked=bool(incs[lf][-1]) if lf is not None else False)
        done_rounds += 1
    ends, incs = exchange()
    if any(v.any() for v in incs.values()):
        for r in owners:                      # left to right: a slab is swept once the slab to its left is final
            lf = left_of[r]
            swept += models[r][0].sweep_only(models[lf][0].angles[-1].copy() if lf is not None else None)
    return dict(angles=np.concatenate([models[r][0].angles[models[r][1]:] for r in owners]),
                fk=np.concatenate([models[r][0].fk[models[r][1]:] for r in owners]), rounds=done_rounds, swept=swept, slabs=slabs)


def chunked_oracle(oracle, pose, seg, bounds, seeds, chunk, halo, tol=1e-6, rounds=3, init=None, guard=False):
    """One chain: pose (N, 5, 3) -> dict(angles (N, 7), fk (N, 9, 3), stats int32[16], flags uint8[K]) (as chunk_stats /
    chunk_flags)."""
    m = ChunkedChain(oracle, pose, seg, bounds, seeds, chunk, halo, tol, rounds, init, guard)
    m.speculate()
    m.settle()
    return dict(angles=m.angles, fk=m.fk, stats=m.stats, flags=m.flags)


def sharded_chunked_oracle(oracle, pose, seg, bounds, seeds, chunk, halo, world, tol=1e-6):
    """One chain of ONE recording sharded by frame over `world` ranks, in one process: the loop of
    seqikpy_amd.frame_sharding.FrameShardedRecording.solve (contiguous slabs of whole chunks on the global chunk grid, every slab
    speculates with `lead` = min(halo, first frame) run-in frames, end states exchanged, ranks > 0 settle their first chunk in
    resume calls until no end state changes) with a ChunkedChain per slab.  -> dict(angles (N, 7), fk, rounds, resumes, slabs,
    repaired_after_exchange).  tests/test_distributed_gloo.py checks it against the real loop over a gloo process group."""
    N = pose.shape[0]
    n_chunks = -(-N // chunk)

    def part(r):
        base, rem = divmod(n_chunks, world)
        a = r * base + min(r, rem)
        return a, a + base + (1 if r < rem else 0)
    slabs = [(min(part(r)[0] * chunk, N), min(part(r)[1] * chunk, N)) for r in range(world)]
    models = {}
    for r, (a, b) in enumerate(slabs):
        if b <= a:
            continue
        lead = min(halo, a) if r > 0 else 0
        if lead == 0 and chunk >= b - a and r == 0 and world == 1:
            raise ValueError("a recording of one chunk is walked serially")
        m = ChunkedChain(oracle, pose[a - lead:b], seg, bounds, seeds, chunk, halo, tol=tol, lead=lead)
        m.speculate()
        m.settle()
        models[r] = (m, lead)
    owners = sorted(models)
    left_of = {r: max([q for q in owners if q < r], default=None) for r in owners}
    left_prev, rounds, resumes, repaired = {}, 0, 0, 0
    while world > 1:
        ends = {r: models[r][0].angles[-1].copy() for r in owners}
        changed = 0
        for r in owners:
            if left_of[r] is None:
                continue
            left = ends[left_of[r]]
            if r not in left_prev or not np.array_equal(left, left_prev[r]):
                m = models[r][0]
                before = m.angles[-1].copy()
                m.settle(init=left.copy(), resume=True)
                repaired += int(m.stats[3:7].sum())
                resumes += 1
                left_prev[r] = left.copy()
                changed += int(not np.array_equal(before, m.angles[-1]))
        if changed == 0:
            break
        rounds += 1
    return dict(angles=np.concatenate([models[r][0].angles[models[r][1]:] for r in owners]),
                fk=np.concatenate([models[r][0].fk[models[r][1]:] for r in owners]), rounds=rounds, resumes=resumes, slabs=slabs,
                repaired_after_exchange=repaired)
