#!/usr/bin/env python3
"""Wall-clock of LegInvKinGeneric-shaped calls (one 9-link chain per leg, 7 unknowns per frame, serial in time) with
and without the split of a pass over groups of 8 lanes (seqik_generic.hpp "Lane groups").  One JSON line per case.

    python scripts/latency_generic.py            (needs a GPU)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np  # noqa: E402

from seqikpy_amd import _lib  # noqa: E402


def params(z, legs):
    return [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]


def main():
    za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
    zd = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
    legs6 = [str(l) for l in zd["legs"]]
    cases = [("anipose RF, 6000 frames (the reference's generic example)", np.stack([za["RF_pose"]])[None], params(za, ["RF"])),
             ("anipose RF + LF, 6000 frames", np.stack([za["RF_pose"], za["LF_pose"]])[None], params(za, ["RF", "LF"])),
             ("df3d RF, 1000 frames", np.stack([zd["RF_pose"]])[None], params(zd, ["RF"]))]
    for name, pose, p in cases:
        outs = {}
        for groups in (False, True):
            _lib.solve_generic(pose, p, lane_groups=groups)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                outs[groups] = _lib.solve_generic(pose, p, lane_groups=groups)
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"case": name, "lane_groups": groups, "ms": round(best * 1e3, 2),
                              "frames_per_s": round(pose.shape[1] * pose.shape[2] / best)}), flush=True)
        same = np.array_equal(outs[True]["angles"], outs[False]["angles"]) and np.array_equal(outs[True]["fk"], outs[False]["fk"])
        print(json.dumps({"case": name, "bit_identical_with_and_without_lane_groups": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
