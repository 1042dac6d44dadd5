"""Six legs of a locomotion recording -- counterpart of the reference's examples/example_leg_inv_kinematics_parallel.py,
which starts one worker process per leg.  Here the six legs are six chains of ONE launch; --pool keeps the reference's
shape (a process pool, one LegInvKinSeq per leg, results merged) for callers built around it: the host objects pickle,
every worker submits to the same GPU.

    python examples/leg_inv_kinematics_parallel.py [-p <dir with pose3d_aligned.pkl>] [--pool] [--frame-chunks]

Without -p the 1000-frame df3d cut in tests/golden/df3d_1000.npz is used.
"""
import argparse
import os
import sys
import time
from multiprocessing import get_context
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402

from seqikpy_amd.data import BOUNDS_LOCOMOTION, INITIAL_ANGLES_LOCOMOTION, TEMPLATE_NMF_LOCOMOTION  # noqa: E402
from seqikpy_amd.kinematic_chain import KinematicChainSeq  # noqa: E402
from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq  # noqa: E402
from seqikpy_amd.utils import calculate_body_size, load_file  # noqa: E402

LEGS = ["RF", "RM", "RH", "LF", "LM", "LH"]


def solve(aligned_pos, legs, frame_parallel="auto"):
    kin_chain = KinematicChainSeq(bounds_dof=BOUNDS_LOCOMOTION, body_size=calculate_body_size(TEMPLATE_NMF_LOCOMOTION, legs),
                                  legs_list=legs)
    seq_ik = LegInvKinSeq(aligned_pos=aligned_pos, kinematic_chain_class=kin_chain, initial_angles=INITIAL_ANGLES_LOCOMOTION,
                          log_level="ERROR")
    return seq_ik.run_ik_and_fk(hide_progress_bar=True, frame_parallel=frame_parallel)


def worker_wrapper(aligned_pos, single_leg):
    """One leg per task, as in the reference."""
    return solve(aligned_pos, [single_leg])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-p", "--path", default=None)
    ap.add_argument("--pool", action="store_true", help="the reference's process pool, one leg per task")
    ap.add_argument("--processes", type=int, default=6, help="workers of the pool (the reference uses 6)")
    ap.add_argument("--frame-chunks", action="store_true", help="(default since round 6; kept for older command lines)")
    ap.add_argument("--serial", action="store_true", help="frame_parallel=False: every chain walked frame by frame, as the reference does")
    args = ap.parse_args(argv)
    if args.path:
        pose_data = load_file(Path(args.path) / "pose3d_aligned.pkl")
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
        pose_data = {f"{leg}_leg": z[f"{leg}_pose"] for leg in LEGS}
    legs = [leg for leg in LEGS if f"{leg}_leg" in pose_data]
    start = time.time()
    if args.pool:
        # spawn: a forked child must not inherit an initialised GPU runtime
        with get_context("spawn").Pool(processes=max(1, min(len(legs), args.processes))) as pool:
            results = pool.starmap(worker_wrapper, [(pose_data, leg) for leg in legs])
        all_legs_joint_angles, all_legs_for_kins = {}, {}
        for ik, fk in results:
            all_legs_joint_angles.update(ik)
            all_legs_for_kins.update(fk)
    else:
        all_legs_joint_angles, all_legs_for_kins = solve(pose_data, legs, False if args.serial else "auto")
    n = len(next(iter(all_legs_joint_angles.values())))
    how = "process pool" if args.pool else "one launch"
    print(f"Sequential IK of {len(legs)} legs x {n} frames took {time.time() - start:.3f} s [{how}]")
    for leg in legs:
        series = np.stack([all_legs_joint_angles[k] for k in all_legs_joint_angles if k.startswith(f"Angle_{leg}_")], 1)
        print(f"  {leg}: {series.shape[1]} angles, frame-to-frame |step| median {np.median(np.abs(np.diff(series, axis=0))):.4f} rad")
    return all_legs_joint_angles, all_legs_for_kins


if __name__ == "__main__":
    main()
