"""Frame-parallel solve of LONG recordings, HOST-ORCHESTRATED (round 1).  Superseded by the library's frame chunks
(``SeqikOptions.frame_chunk``, include/seqik.h: the same speculate / verify / repair scheme entirely on the device, no host
round trips, 7x faster on the shipped recordings) -- ``LegInvKinSeq.run_ik_and_fk`` and ``frame_sharding`` use those.
Kept as a reference implementation of the scheme in ~100 lines of numpy ("sharding by frame within
one sequence").

The reference walks a recording serially because frame t is warm-started from frame t-1
(``seqikpy/leg_inverse_kinematics.py:272``).  One lane per (recording, leg) therefore leaves a GPU
idle when there are only a few long recordings.  This module cuts every recording into chunks and
solves all chunks concurrently, then *verifies* each chunk against its true predecessor:

1. speculative pass: chunk k >= 1 is started from the seeds ``halo`` frames before its first frame
   (the solver forgets its start point within a few frames on well-posed data); chunk 0 starts from
   the seeds at frame 0, as the reference does;
2. verification: the state the halo run reaches just before the chunk (7 joint angles) is compared
   with the true state = last frame of the preceding chunk; a chunk is accepted if they agree to
   ``tol`` rad;
3. repair: rejected chunks are re-solved from the true state (``init_angles`` of the C ABI --
   bit-identical to what the serial run does), their successors are re-verified, until none is left.

Accepted chunks start within ``tol`` of the serial trajectory and the solver contracts such
differences, so the result equals the serial one to about ``tol`` (default 1e-6 rad = 1 % of the
1e-4 rad parity bar); where the reference itself is chaotic (kinematic singularities)
a 1e-6 difference can still grow -- use the default serial mode when bit-reproducibility matters.
"""
from typing import Dict, List, Optional

import numpy as np

from . import _lib


def solve_frame_parallel(pose: np.ndarray, legs: List, chunk: int = 32, halo: int = 16, tol: float = 1e-6,
                         want_fk: bool = True, affine=None, device: int = -1, stats: Optional[Dict] = None,
                         lead: int = 0, init_angles: Optional[np.ndarray] = None):
    """``pose`` (S, L, lead + N, 5, 3) -> dict(angles (S, L, N, 7), fk (S, L, N, 9, 3) or None).

    Equivalent to ``_lib.solve_seq(pose, legs)`` (stages 1-4) up to ``tol``; see the module docstring.
    ``stats`` (optional dict) receives ``chunks``, ``repaired``, ``rounds`` and ``start_state0``.

    ``lead`` / ``init_angles`` are for pieces of a longer recording (``frame_sharding.py``): the first ``lead``
    frames are only a run-in for the first chunk (solved, not returned; the state they reach,
    ``stats["start_state0"]`` (S, L, 7), is what the caller verifies against the true predecessor), and
    ``init_angles`` (S, L, 7) warm-starts frame 0 instead of the seeds (continuation from a known state)."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    S, L, n_local = pose.shape[:3]
    if chunk < 1 or halo < 0 or lead < 0 or lead > n_local:
        raise ValueError("chunk must be >= 1, halo >= 0 and 0 <= lead <= number of frames")
    N = n_local - lead
    K = -(-N // chunk) if N else 0
    if K <= 1:
        out = _lib.solve_seq(pose, legs, want_fk=want_fk, affine=affine, device=device, init_angles=init_angles)
        if stats is not None:
            stats.update(chunks=K, repaired=0, rounds=0,
                         start_state0=out["angles"][:, :, lead - 1].copy() if lead > 0 else None)
        return dict(angles=out["angles"][:, :, lead:], fk=out["fk"][:, :, lead:] if want_fk else None)

    angles = np.empty((S, L, N, 7))
    fk = np.empty((S, L, N, 9, 3)) if want_fk else None

    def store(k, res_angles, res_fk, offset):
        a, b = k * chunk, min((k + 1) * chunk, N)
        angles[:, :, a:b] = res_angles[:, :, offset:offset + (b - a)]
        if want_fk:
            fk[:, :, a:b] = res_fk[:, :, offset:offset + (b - a)]

    def local(idx):  # output-frame indices (clipped copies of the last frame pad the final chunk) -> pose frames
        return lead + np.clip(idx, 0, N - 1)

    # ---- 1. speculative pass -----------------------------------------------------------------------
    first = _lib.solve_seq(pose[:, :, np.concatenate([np.arange(lead), local(np.arange(chunk))])], legs,
                           want_fk=want_fk, affine=affine, device=device, init_angles=init_angles)
    store(0, first["angles"], first["fk"], lead)
    start_state0 = first["angles"][:, :, lead - 1].copy() if lead > 0 else None
    h = min(halo, chunk)  # a halo longer than a chunk would reach past the predecessor
    win = np.stack([local(np.arange(k * chunk - h, (k + 1) * chunk)) for k in range(1, K)])  # (K-1, W)
    batch = pose[:, :, win]                      # (S, L, K-1, W, 5, 3)
    batch = np.ascontiguousarray(batch.transpose(0, 2, 1, 3, 4, 5)).reshape(S * (K - 1), L, win.shape[1], 5, 3)
    spec = _lib.solve_seq(batch, legs, want_fk=want_fk, affine=affine, device=device)
    spec_ang = spec["angles"].reshape(S, K - 1, L, win.shape[1], 7)
    spec_fk = spec["fk"].reshape(S, K - 1, L, win.shape[1], 9, 3) if want_fk else None
    for k in range(1, K):
        store(k, spec_ang[:, k - 1], spec_fk[:, k - 1] if want_fk else None, h)
    # state reached by the halo run just before each chunk (window frame h - 1 = output frame k * chunk - 1)
    start_state = spec_ang[:, :, :, h - 1] if h > 0 else None    # (S, K-1, L, 7)

    # ---- 2./3. verify, repair, cascade ----------------------------------------------------------------
    def true_state(k):  # joint angles of the frame preceding chunk k: (S, L, 7)
        return angles[:, :, k * chunk - 1]

    dirty = np.zeros((S, K), dtype=bool)
    for k in range(1, K):
        if h == 0:
            dirty[:, k] = True
        else:
            dirty[:, k] = np.abs(start_state[:, k - 1] - true_state(k)).max(axis=(1, 2)) > tol
    repaired = 0
    rounds = 0
    while dirty.any():
        rounds += 1
        # a dirty chunk can be repaired once its predecessor is final
        ready = dirty & ~np.concatenate([np.zeros((S, 1), bool), dirty[:, :-1]], axis=1)
        s_idx, k_idx = np.nonzero(ready)
        frames = np.stack([local(np.arange(k * chunk, (k + 1) * chunk)) for k in k_idx])               # (R, chunk)
        rp = np.stack([pose[s][:, f] for s, f in zip(s_idx, frames)])                                   # (R, L, chunk, 5, 3)
        init = np.stack([angles[s, :, k * chunk - 1] for s, k in zip(s_idx, k_idx)])                    # (R, L, 7)
        res = _lib.solve_seq(rp, legs, want_fk=want_fk, affine=affine, device=device, init_angles=init)
        for i, (s, k) in enumerate(zip(s_idx, k_idx)):
            a, b = k * chunk, min((k + 1) * chunk, N)
            angles[s, :, a:b] = res["angles"][i, :, : b - a]
            if want_fk:
                fk[s, :, a:b] = res["fk"][i, :, : b - a]
            dirty[s, k] = False
            repaired += 1
            # the successor was verified against the old end state of this chunk: check again
            if k + 1 < K and not dirty[s, k + 1]:
                if h == 0 or np.abs(start_state[s, k] - angles[s, :, (k + 1) * chunk - 1]).max() > tol:
                    dirty[s, k + 1] = True
    if stats is not None:
        stats.update(chunks=int(S * K), repaired=int(repaired), rounds=int(rounds), start_state0=start_state0)
    return dict(angles=angles, fk=fk)
