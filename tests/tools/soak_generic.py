#!/usr/bin/env python3
"""Soak test of the GENERIC chain (LegInvKinGeneric path): many made-up legs (random segment lengths, joint limits, seeds on
bounds) with unreachable / degenerate / repeated targets, HIP vs the C oracle, bit for bit (angles, FK, scipy status, nfev),
over both instantiations of the kernel -- thin waves with lane groups of 8 (few chains) and one lane per chain (>= 1024
chains).  Prints one JSON line.

    python tests/tools/soak_generic.py --legs 64 --seqs 160 --frames 12          (needs a GPU)
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sequential-inverse-kinematics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from conftest import random_leg_case  # noqa: E402
from oracle import c_oracle  # noqa: E402  (checker)
from seqikpy_amd import _lib  # noqa: E402


GENERIC_LINK_DOF = [2, 0, 1, 3, 4, 5, 6]   # link i + 1 of the generic chain carries this DOF (kinematic_chain.py:464-530)


def generic_case(rng, n_frames):
    """random_leg_case with the stage-4 seed vector re-drawn for the GENERIC chain, which applies it positionally to
    Base, roll, yaw, pitch, CTr_pitch, CTr_roll, FTi, TiTa, Claw (leg_inverse_kinematics.py:588): inside the limits of
    those links, sometimes exactly on one, sometimes exactly 0."""
    pose, seg, bounds, seeds = random_leg_case(rng, n_frames)
    seeds = seeds.copy()
    seeds[18] = 0.0
    for i, dof in enumerate(GENERIC_LINK_DOF):
        lb, ub = bounds[dof]
        u = rng.choice([0.0, 1.0, 2.0, rng.random()], p=[0.1, 0.1, 0.1, 0.7])
        v = lb if u == 0.0 else (ub if u == 1.0 else (min(max(0.0, lb), ub) if u == 2.0 else lb + u * (ub - lb)))
        seeds[19 + i] = min(max(v, lb), ub)
    seeds[26] = rng.uniform(-1.0, 1.0)
    return pose, seg, bounds, seeds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", type=int, default=64, help="made-up legs (8 per launch)")
    ap.add_argument("--seqs", type=int, default=160, help="recordings per leg in the one-lane-per-chain launches (8 x seqs >= 1024)")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--queue", action="store_true",
                    help="every one-lane-per-chain launch also on the chain queue (SeqikOptions.reserved[1] = 2): same bits wanted")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    c_oracle.lib()
    bad, n_lf, t_cpu, t_gpu = [], 0, 0.0, 0.0
    status_hist = np.zeros(5, dtype=np.int64)
    nfev_max = 0
    launches = {"lane groups (thin waves)": 0, "one lane per chain": 0, "chain queue": 0}
    bad_queue = []
    for g0 in range(0, a.legs, 8):
        legs = [generic_case(rng, a.frames) for _ in range(8)]
        # recordings: the leg's own nasty poses first, then the poses of other made-up legs (any key points are a valid input)
        S = a.seqs if (g0 // 8) % 2 == 0 else int(rng.integers(1, 5))
        pose = np.empty((S, 8, a.frames, 5, 3))
        for li, c in enumerate(legs):
            pose[0, li] = c[0]
            for s in range(1, S):
                pose[s, li] = random_leg_case(rng, a.frames)[0]
        params = [_lib.leg_params_from_arrays(c[1], c[2], c[3]) for c in legs]

        def ref(idx):
            s, li = divmod(idx, 8)
            c = legs[li]
            return c_oracle.generic_leg(pose[s, li], c[1], c[2], c[3][18:27])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(8) as ex:
            refs = list(ex.map(ref, range(S * 8)))
        t_cpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        out = _lib.solve_generic(pose, params, want_fk=True, want_diag=True)
        t_gpu += time.perf_counter() - t0
        launches["one lane per chain" if S * 8 >= 1024 else "lane groups (thin waves)"] += 1
        if a.queue and S * 8 >= 1024:
            outq = _lib.solve_generic(pose, params, want_fk=True, want_diag=True, chain_queue=2)
            launches["chain queue"] += 1
            if not all(np.array_equal(outq[k], out[k]) for k in ("angles", "fk", "status", "nfev")):
                bad_queue.append(g0)
        for idx, r in enumerate(refs):
            s, li = divmod(idx, 8)
            ok = (np.array_equal(out["angles"][s, li], r["angles"]) and np.array_equal(out["fk"][s, li], r["fk"]) and
                  np.array_equal(out["status"][s, li], r["status"]) and np.array_equal(out["nfev"][s, li], r["nfev"]))
            if not ok:
                bad.append([g0, s, li])
            status_hist += np.bincount(r["status"], minlength=5)[:5]
            nfev_max = max(nfev_max, int(r["nfev"].max()))
        n_lf += S * 8 * a.frames
    print(json.dumps({"legs": a.legs, "frames_per_recording": a.frames, "leg_frames": n_lf, "mismatching_recordings": len(bad),
                      "first_mismatches": bad[:10], "launches": launches,
                      "chain_queue_launches_differing_from_the_static_launch": len(bad_queue),
                      "status_histogram": {int(i): int(v) for i, v in enumerate(status_hist)}, "max_nfev": nfev_max,
                      "cpu_oracle_seconds": t_cpu, "gpu_seconds_incl_transfers": t_gpu}))


if __name__ == "__main__":
    main()
