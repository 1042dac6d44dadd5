"""Sequential and generic leg IK of one recording on MI355X -- counterpart of the reference's
examples/example_leg_inv_kinematics.py ("will take about 30 minutes") and, with several legs, of
examples/example_leg_inv_kinematics_parallel.py (one process per leg there; one launch here).

    python examples/leg_inv_kinematics.py -p <dir with pose3d_aligned.pkl> [--legs RF LF] [--generic] [--export]

Without -p the shipped anipose recording cut into tests/golden/anipose_shipped.npz is used (RF, LF x 6000 frames).
"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402

from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES  # noqa: E402
from seqikpy_amd.kinematic_chain import KinematicChainGeneric, KinematicChainSeq  # noqa: E402
from seqikpy_amd.leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq  # noqa: E402
from seqikpy_amd.utils import load_file  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-p", "--path", default=None, help="directory holding pose3d_aligned.pkl")
    ap.add_argument("--legs", nargs="+", default=["RF", "LF"])
    ap.add_argument("--generic", action="store_true", help="also run the single-chain (generic) IK")
    ap.add_argument("--frame-chunks", action="store_true", help="(default since round 6; kept for older command lines)")
    ap.add_argument("--serial", action="store_true",
                    help="walk every recording frame by frame as the reference does (frame_parallel=False: bit-exact restatement of "
                         "the reference) instead of the default, concurrently solved and verified frame chunks (frame_parallel="
                         "'auto': 10-70x faster for one recording, equal to the serial walk to ~1e-5 rad)")
    ap.add_argument("--export", action="store_true", help="write leg_joint_angles.pkl / forward_kinematics.pkl")
    args = ap.parse_args()
    if args.path:
        data_path = Path(args.path)
        aligned_pos = load_file(data_path / "pose3d_aligned.pkl")
    else:
        data_path = Path(ROOT) / "gpurun_out"
        z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
        aligned_pos = {f"{leg}_leg": z[f"{leg}_pose"] for leg in ("RF", "LF")}
    export = data_path if args.export else None

    start = time.time()
    seq_ik = LegInvKinSeq(aligned_pos=aligned_pos,
                          kinematic_chain_class=KinematicChainSeq(bounds_dof=BOUNDS, legs_list=args.legs, body_size=None),
                          initial_angles=INITIAL_ANGLES)
    angles_seq, fk_seq = seq_ik.run_ik_and_fk(export_path=export, hide_progress_bar=True,
                                              frame_parallel=False if args.serial else "auto")
    n = len(next(iter(angles_seq.values())))
    print(f"Sequential IK of {len(fk_seq)} legs x {n} frames took {time.time() - start:.3f} s")

    if args.generic:
        start = time.time()
        gen_ik = LegInvKinGeneric(aligned_pos=aligned_pos,
                                  kinematic_chain_class=KinematicChainGeneric(bounds_dof=BOUNDS, legs_list=args.legs,
                                                                              body_size=None),
                                  initial_angles=INITIAL_ANGLES)
        angles_gen, fk_gen = gen_ik.run_ik_and_fk(export_path=None, hide_progress_bar=True)
        print(f"Generic IK took {time.time() - start:.3f} s")
        for name in fk_seq:
            d = np.linalg.norm(fk_seq[name][:, -1] - fk_gen[name][:, -1], axis=1)
            print(f"  {name}: claw position, sequential vs generic: median {np.median(d):.4f}, max {d.max():.4f} (mm)")


if __name__ == "__main__":
    main()
