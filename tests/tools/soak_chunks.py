#!/usr/bin/env python3
"""Soak test of the frame-chunk machinery (SeqikOptions.frame_chunk): made-up legs with unreachable / degenerate /
repeated targets -- data on which run-ins often fail verification, so repair rounds, cascades and the serial sweep all
run -- HIP path vs the oracle-built model of the launch sequence (tests/chunk_model.py), bit for bit, statistics
included.  Both kernels (lane per chunk, stage pipeline).  Prints one JSON line.

    python tests/tools/soak_chunks.py --cases 4000 --frames 96          (needs a GPU)
"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sequential-inverse-kinematics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from chunk_model import chunked_oracle  # noqa: E402
from conftest import random_leg_case  # noqa: E402
from oracle import c_oracle  # noqa: E402  (checker)
from seqikpy_amd import _lib  # noqa: E402

# (chunk, halo, tol, rounds); chunk -1 = the automatic mode: geometry from the number of frames, per-chain guard (a chain
# of which more than one chunk in eight fails the first verification is walked serially)
SHAPES = [(8, 3, 1e-6, 2), (5, 8, 1e-6, 3), (16, 4, 1e-3, 1), (7, 2, 1e-6, 8), (-1, 0, 1e-6, 3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=4000)
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cases = [random_leg_case(rng, args.frames) for _ in range(args.cases)]
    c_oracle.lib()

    from chunk_model import plan

    def geometry(shape):
        c, h, tol, r = shape
        if c < 0:
            c, h, _ = plan(args.frames, -1, 0)
            return c, h, tol, r, True
        return c, h, tol, r, False

    def ref(i):
        c, h, tol, r, guard = geometry(SHAPES[(i // 8) % len(SHAPES)])
        return chunked_oracle(c_oracle, *cases[i], c, h, tol=tol, rounds=r, guard=guard)

    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(ref, range(args.cases)))
    bad, totals = [], np.zeros(16, np.int64)
    for g0 in range(0, args.cases, 8):
        grp = cases[g0:g0 + 8]
        shape = SHAPES[(g0 // 8) % len(SHAPES)]
        c, h, tol, r, guard = geometry(shape)
        pose = np.stack([x[0] for x in grp])[None]
        params = [_lib.leg_params_from_arrays(x[1], x[2], x[3]) for x in grp]
        want_stats = np.sum([refs[g0 + i]["stats"] for i in range(len(grp))], 0)
        want_stats[1:3] = (c, h)
        for pl in (1, 2):
            kw = dict(frame_chunk=-1) if guard else dict(frame_chunk=c, frame_halo=h)
            out = _lib.solve_seq(pose, params, want_fk=True, chunk_tol=tol, chunk_rounds=r, pipeline=pl, want_chunk_flags=True, **kw)
            got_stats = np.array([out["chunk_stats"][k] for k in _lib.CHUNK_STATS_FIELDS])
            ok = np.array_equal(got_stats, want_stats[:len(got_stats)])
            for i in range(len(grp)):
                ok = ok and np.array_equal(out["angles"][0, i], refs[g0 + i]["angles"]) and np.array_equal(out["fk"][0, i], refs[g0 + i]["fk"])
                ok = ok and np.array_equal(out["chunk_flags"][0, i], refs[g0 + i]["flags"])
            if not ok:
                bad.append((g0, pl))
        totals += want_stats
    print(json.dumps({"cases": args.cases, "frames_per_case": args.frames, "shapes_chunk_halo_tol_rounds": SHAPES,
                      "launches": 2 * ((args.cases + 7) // 8), "mismatching_launches": len(bad), "first_mismatches": bad[:10],
                      "chunks": int(totals[0]), "inconsistent_at_first_check": int(totals[7]),
                      "repaired_in_rounds": [int(v) for v in totals[3:6]], "repaired_by_sweep": int(totals[6]),
                      "chains_walked_serially_by_the_guard": int(totals[8])}))


if __name__ == "__main__":
    main()
