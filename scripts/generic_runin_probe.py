#!/usr/bin/env python3
"""Can frame chunks help LegInvKinGeneric?  A chunk would start `h` frames early from the seeds and has to REACH the state
of the serial walk (7 angles) to be accepted.  The generic chain has 7 unknowns and 3 equations: the warm start decides
which point of the 4-dimensional solution set a frame ends up in, so nothing pulls a run-in towards the serial walk's
angles.  This probe measures it on the shipped recording: for run-ins of h = 8, 32, 128 frames starting every 250 frames,
the distance between the run-in's last frame and the serial walk's frame (angles and claw).  One JSON line.  Needs a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np  # noqa: E402

from seqikpy_amd import _lib  # noqa: E402


def main():
    z = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
    p = [_lib.leg_params_from_arrays(z["RF_seg"], z["RF_bounds"], z["RF_seeds"])]
    pose = z["RF_pose"][None, None]                                  # (1, 1, 6000, 5, 3)
    serial = _lib.solve_generic(pose, p)
    out = {}
    for h in (8, 32, 128):
        starts = list(range(250, 6000, 250))
        runs = np.stack([pose[0, 0, t - h:t + 1] for t in starts])[:, None]   # (n, 1, h + 1, 5, 3): ends ON frame t
        got = _lib.solve_generic(runs, p)
        d_ang = np.abs(got["angles"][:, 0, -1] - serial["angles"][0, 0, starts]).max(-1)
        d_claw = np.abs(got["fk"][:, 0, -1, 8] - serial["fk"][0, 0, starts, 8]).max(-1)
        out[f"run_in_{h}"] = {"chunks": len(starts), "angles_within_1e-6": int((d_ang <= 1e-6).sum()),
                              "angles_within_1e-3": int((d_ang <= 1e-3).sum()), "median_angle_distance_rad": float(np.median(d_ang)),
                              "max_claw_distance": float(d_claw.max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
