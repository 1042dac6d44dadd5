"""GPU tier: the end-to-end pipeline (examples/entire_pipeline.py) on a cut of the shipped recording:
converted segment dictionary -> alignment -> head / antenna angles -> sequential leg IK -> pickles with the
reference's file names and keys."""
import importlib.util
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import DOFS, ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import head_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frame_parallel", [False, True])  # False: --serial; True: the default (verified frame chunks)
def test_entire_pipeline(tmp_path, oracle, hiplib, monkeypatch, frame_parallel):
    z = load_golden("anipose_raw_cut")
    raw = {str(k): z[f"raw_{k}"] for k in z["segments"]}
    with open(tmp_path / "converted_dict.pkl", "wb") as f:
        pickle.dump(raw, f)
    spec = importlib.util.spec_from_file_location("entire_pipeline", os.path.join(ROOT, "examples", "entire_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ["entire_pipeline.py", "-p", str(tmp_path)] + ([] if frame_parallel else ["--serial"])
    monkeypatch.setattr(sys, "argv", argv)
    mod.main()
    for name in ("pose3d_aligned.pkl", "head_joint_angles.pkl", "leg_joint_angles.pkl", "forward_kinematics.pkl",
                 "body_joint_angles.pkl"):
        assert os.path.exists(tmp_path / name), name
    body = pickle.load(open(tmp_path / "body_joint_angles.pkl", "rb"))
    assert len(body) == 21 and all(v.shape == (1500,) for v in body.values())
    # what the reference's consumers need from these files (visualization.py:191-213, 443-492; utils.py:235-245)
    import loader_contract as lc
    joint_angles, aligned_pose = lc.load_grid_plot_data(tmp_path)
    lc.check_joint_angles(joint_angles, ["RF", "LF"], 1500, with_head=True)
    lc.check_points3d(aligned_pose, 1500, leg_points=5)
    lc.check_points3d(lc.load_file(tmp_path / "forward_kinematics.pkl"), 1500, leg_points=9)
    os.remove(tmp_path / "body_joint_angles.pkl")  # the loader's second branch: head + leg files merged
    merged, _ = lc.load_grid_plot_data(tmp_path)
    assert set(merged) == set(body) and all(np.array_equal(merged[k], body[k]) for k in body)
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
    from seqikpy_amd.utils import calculate_body_size
    body_size = calculate_body_size(NMF_TEMPLATE, ["RF", "LF"])
    tol = 0.0 if not frame_parallel else 2e-5  # chunks start within 1e-6 rad of the serial state (tests/test_frame_chunks.py)
    for leg in ("RF", "LF"):
        seg, b, seeds = oracle.leg_params(leg, BOUNDS, body_size, INITIAL_ANGLES)
        ref = oracle.seq_leg(z[f"aligned_{leg}_leg"], seg, b, seeds)
        got = np.stack([body[f"Angle_{leg}_{d}"] for d in DOFS], 1)
        if leg == "LF":  # the singular episode (frames ~284-301) amplifies 1e-6 differences
            keep = np.r_[0:280, 310:1500]
            assert np.abs(got - ref["angles"])[keep].max() <= tol
        else:
            assert np.abs(got - ref["angles"]).max() <= tol
    from seqikpy_amd.head_inverse_kinematics import HeadInverseKinematics
    hk = HeadInverseKinematics({k: z[f"aligned_{k}"] for k in ("R_head", "L_head", "Neck")}, NMF_TEMPLATE, log_level="ERROR")
    want = head_oracle.head_angles(z["aligned_R_head"], z["aligned_L_head"], z["aligned_Neck"][:, 0],
                                   hk.rest_head_pitch, hk.rest_antenna_pitch)
    assert np.abs(body["Angle_antenna_pitch_R"] - want[6]).max() < 1e-6
    assert np.abs(body["Angle_head_roll"] - want[0]).max() < 1e-6


def test_legs_and_head_in_one_submission(hiplib):
    """Config 4: pipeline.run_body_ik (leg kernel + head / antenna kernel on two streams, one upload / sync /
    download) returns the bits of HeadInverseKinematics + LegInvKinSeq run one after the other."""
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
    from seqikpy_amd.head_inverse_kinematics import HeadInverseKinematics
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.pipeline import run_body_ik
    z = load_golden("anipose_raw_cut")
    aligned = {k: z[f"aligned_{k}"] for k in ("R_head", "L_head", "Neck", "RF_leg", "LF_leg")}
    kc = KinematicChainSeq(BOUNDS, ["RF", "LF"])
    body, fk = run_body_ik(aligned, kc, NMF_TEMPLATE, INITIAL_ANGLES)
    head = HeadInverseKinematics(aligned, NMF_TEMPLATE, log_level="ERROR").compute_head_angles()
    legs, fk_ref = LegInvKinSeq(aligned, kc, INITIAL_ANGLES, log_level="ERROR").run_ik_and_fk()
    assert list(body.keys()) == list(head.keys()) + list(legs.keys()) and len(body) == 21
    for k, v in {**head, **legs}.items():
        assert np.array_equal(body[k], v), k
    assert list(fk.keys()) == list(fk_ref.keys()) and all(np.array_equal(fk[k], fk_ref[k]) for k in fk)
    legs_only, _ = run_body_ik({k: aligned[k] for k in ("RF_leg", "LF_leg")}, kc, NMF_TEMPLATE)
    assert len(legs_only) == 14 and np.array_equal(legs_only["Angle_LF_TiTa_pitch"], legs["Angle_LF_TiTa_pitch"])
