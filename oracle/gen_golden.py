#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz (build container only).

Runs the REFERENCE'S OWN source (``/root/reference/seqikpy``: ``LegInvKinSeq``,
``LegInvKinGeneric``, ``KinematicChainSeq``, ``KinematicChainGeneric``,
``AlignPose``) unmodified, over the build-owned ``ikpy`` shim + the container's real
scipy (``oracle/ref_import.py``), and cuts the reference's shipped data files
(inputs and shipped pipeline outputs) into small fixtures.  A fixture is data only:
inputs, expected outputs, and the per-leg parameter arrays (segment lengths, bounds,
seeds) that the run used.  Nothing here travels to the GPU box except the .npz files.

    python oracle/gen_golden.py [--only NAME ...]

Fixtures (all float64 unless noted; DOF order = c_oracle.DOFS):
  anipose_shipped.npz   pose RF/LF (6000,5,3) from pose3d_aligned.pkl; shipped
                        leg_joint_angles.pkl as (6000,7) per leg; shipped
                        forward_kinematics.pkl cut to frames 0:100 and 250:330
  anipose_scipy_cut.npz reference source run here on frames 0:330 (RF, LF):
                        angles, fk, scipy status / nfev per (frame, stage)
  df3d_100.npz          shipped df3d pose3d_aligned.pkl (6 legs x 100) + reference run here
  df3d_1000.npz         full 1000-frame df3d recording aligned here with the
                        reference's AlignPose + reference run here
  generic_rf_100.npz    LegInvKinGeneric on anipose RF frames 0:100, run here
  anipose_raw_cut.npz   converted_dict.pkl[:1500] (un-aligned legs, antennae, thorax) + reference AlignPose output
  anipose_head.npz      aligned head key points, shipped head_joint_angles.pkl, HeadInverseKinematics run here
  df3d_align_pins.npz   reference-HELD pins of the alignment row on all six legs: the un-aligned df3d key points of
                        frames 300:400 (seqikpy_locomotion.ipynb cell 2), the six find_scale_leg values the
                        notebook's stored cell-6 output prints, and the shipped pose3d_aligned.pkl of that cut
  df3d_notebook_cell16.png  (an image, not an .npz) the STORED OUTPUT of seqikpy_locomotion.ipynb cell 16: real ikpy 3.3.4's
                        joint angles of all six legs x 100 frames (the df3d_100 inputs), as the plot the notebook keeps --
                        the only real-IKPy output for RM / RH / LM / LH and BOUNDS_LOCOMOTION in reach (tests/test_notebook_pin.py)
"""
import argparse
import importlib.util
import multiprocessing as mp
import os
import pickle
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_import import REFERENCE_ROOT, import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
SEGMENTS = ["Coxa", "Femur", "Tibia", "Tarsus"]
ANIPOSE = os.path.join(REFERENCE_ROOT, "data", "anipose_220525_aJO_Fly001_001", "pose-3d")
DF3D = os.path.join(REFERENCE_ROOT, "data", "df3d_pose_result__210902_PR_Fly1")
CUT = np.r_[0:100, 250:330]


def load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def locomotion_constants():
    """BOUNDS/INITIAL_ANGLES/TEMPLATE of the locomotion example (data only; the module's
    work is under ``if __name__ == '__main__'``: examples/example_leg_inv_kinematics_parallel.py:21-140)."""
    import_reference()
    path = os.path.join(REFERENCE_ROOT, "examples", "example_leg_inv_kinematics_parallel.py")
    spec = importlib.util.spec_from_file_location("_ref_example_parallel", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.TEMPLATE_NMF_LOCOMOTION, mod.INITIAL_ANGLES_LOCOMOTION, mod.BOUNDS_LOCOMOTION


def leg_param_arrays(leg, bounds_dof, body_size, initial_angles):
    seg = np.array([body_size[f"{leg}_{s}"] for s in SEGMENTS], dtype=np.float64)
    bounds = np.array([bounds_dof[f"{leg}_{d}"] for d in DOFS], dtype=np.float64)
    seeds = np.concatenate([np.asarray(initial_angles[leg][f"stage_{k}"], dtype=np.float64) for k in (1, 2, 3, 4)])
    return seg, bounds, seeds


def _run_one_leg(args):
    """Worker: the reference's LegInvKinSeq / LegInvKinGeneric on one leg (cf. worker_wrapper,
    examples/example_leg_inv_kinematics_parallel.py:143-160)."""
    kind, leg, pose, bounds_dof, body_size, initial_angles = args
    import_reference()
    from ikpy.chain import Chain
    from seqikpy.kinematic_chain import KinematicChainGeneric, KinematicChainSeq
    from seqikpy.leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq
    Chain.solve_log = []
    aligned = {f"{leg}_leg": pose}
    if kind == "seq":
        ik = LegInvKinSeq(aligned, KinematicChainSeq(bounds_dof, [leg], body_size), initial_angles, log_level="ERROR")
    else:
        ik = LegInvKinGeneric(aligned, KinematicChainGeneric(bounds_dof, [leg], body_size), initial_angles,
                              log_level="ERROR")
    ang, fk = ik.run_ik_and_fk(hide_progress_bar=True)
    n = pose.shape[0]
    angles = np.stack([ang[f"Angle_{leg}_{d}"] for d in DOFS], axis=1)
    log = np.asarray(Chain.solve_log, dtype=np.int32)
    if kind == "seq":
        log = log.reshape(4, n, 2).transpose(1, 0, 2)  # reference loop order is stage-major
    else:
        log = log.reshape(n, 1, 2)
    return leg, angles, fk[f"{leg}_leg"], log[..., 0].copy(), log[..., 1].copy()


def run_reference(kind, poses, bounds_dof, body_size, initial_angles, procs=6):
    jobs = [(kind, leg, pose, bounds_dof, body_size, initial_angles) for leg, pose in poses.items()]
    t0 = time.time()
    with mp.Pool(min(procs, len(jobs))) as pool:
        res = pool.map(_run_one_leg, jobs)
    dt = time.time() - t0
    out = {}
    for leg, angles, fk, status, nfev in res:
        seg, bounds, seeds = leg_param_arrays(leg, bounds_dof, body_size, initial_angles)
        out.update({f"{leg}_pose": poses[leg], f"{leg}_angles": angles, f"{leg}_fk": fk,
                    f"{leg}_status": status, f"{leg}_nfev": nfev,
                    f"{leg}_seg": seg, f"{leg}_bounds": bounds, f"{leg}_seeds": seeds})
    out["legs"] = np.array(list(poses.keys()))
    out["ref_seconds"] = np.float64(dt)
    return out


def gen_anipose_shipped():
    import_reference()
    from seqikpy.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
    from seqikpy.utils import calculate_body_size
    pose = load_pickle(os.path.join(ANIPOSE, "pose3d_aligned.pkl"))
    gold = load_pickle(os.path.join(ANIPOSE, "leg_joint_angles.pkl"))
    gfk = load_pickle(os.path.join(ANIPOSE, "forward_kinematics.pkl"))
    body = calculate_body_size(NMF_TEMPLATE, ["RF", "LF"])
    out = {"legs": np.array(["RF", "LF"]), "fk_frames": CUT}
    for leg in ("RF", "LF"):
        seg, bounds, seeds = leg_param_arrays(leg, BOUNDS, body, INITIAL_ANGLES)
        out.update({f"{leg}_pose": pose[f"{leg}_leg"],
                    f"{leg}_angles": np.stack([gold[f"Angle_{leg}_{d}"] for d in DOFS], axis=1),
                    f"{leg}_fk_cut": gfk[f"{leg}_leg"][CUT],
                    f"{leg}_seg": seg, f"{leg}_bounds": bounds, f"{leg}_seeds": seeds})
    return out


def gen_anipose_scipy_cut():
    import_reference()
    from seqikpy.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
    from seqikpy.utils import calculate_body_size
    pose = load_pickle(os.path.join(ANIPOSE, "pose3d_aligned.pkl"))
    body = calculate_body_size(NMF_TEMPLATE, ["RF", "LF"])
    poses = {leg: pose[f"{leg}_leg"][:330].copy() for leg in ("RF", "LF")}
    return run_reference("seq", poses, BOUNDS, body, INITIAL_ANGLES)


def gen_df3d_100():
    import_reference()
    from seqikpy.utils import calculate_body_size
    template, init, bounds = locomotion_constants()
    legs = ["RF", "RM", "RH", "LF", "LM", "LH"]
    pose = load_pickle(os.path.join(DF3D, "pose3d_aligned.pkl"))
    poses = {leg: pose[f"{leg}_leg"] for leg in legs}
    return run_reference("seq", poses, bounds, calculate_body_size(template, legs), init)


def gen_df3d_1000():
    """Config 2: the full locomotion recording, aligned as in seqikpy_locomotion.ipynb cells 2-6
    but without the 300:400 cut."""
    import_reference()
    from seqikpy.alignment import AlignPose, convert_from_df3dpp_to_dict
    from seqikpy.utils import calculate_body_size
    template, init, bounds = locomotion_constants()
    legs = ["RF", "RM", "RH", "LF", "LM", "LH"]
    raw = load_pickle(os.path.join(DF3D, "pose_result__210902_PR_Fly1_aligned.pkl"))
    converted = convert_from_df3dpp_to_dict(raw)
    align = AlignPose(pose_data_dict=converted, legs_list=legs, include_claw=False,
                      body_template=template, body_size=None, log_level="ERROR")
    aligned = align.align_pose(export_path=None)
    poses = {leg: np.ascontiguousarray(aligned[f"{leg}_leg"], dtype=np.float64) for leg in legs}
    out = run_reference("seq", poses, bounds, calculate_body_size(template, legs), init)
    for leg in legs:  # also keep the un-aligned key points: input of the alignment row
        out[f"{leg}_raw"] = np.ascontiguousarray(converted[f"{leg}_leg"], dtype=np.float64)
    return out


def gen_generic_rf_100():
    import_reference()
    from seqikpy.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
    from seqikpy.utils import calculate_body_size
    pose = load_pickle(os.path.join(ANIPOSE, "pose3d_aligned.pkl"))
    body = calculate_body_size(NMF_TEMPLATE, ["RF", "LF"])
    poses = {leg: pose[f"{leg}_leg"][:100].copy() for leg in ("RF", "LF")}
    return run_reference("generic", poses, BOUNDS, body, INITIAL_ANGLES)


def gen_anipose_head():
    """Head / antenna row: aligned R_head, L_head, Neck of the shipped anipose recording,
    the shipped head_joint_angles.pkl and the reference's HeadInverseKinematics re-run here."""
    import_reference()
    from seqikpy.data import NMF_TEMPLATE
    from seqikpy.head_inverse_kinematics import HeadInverseKinematics
    pose = load_pickle(os.path.join(ANIPOSE, "pose3d_aligned.pkl"))
    gold = load_pickle(os.path.join(ANIPOSE, "head_joint_angles.pkl"))
    names = ["Angle_head_roll", "Angle_head_pitch", "Angle_head_yaw", "Angle_antenna_yaw_L",
             "Angle_antenna_pitch_L", "Angle_antenna_yaw_R", "Angle_antenna_pitch_R"]
    hk = HeadInverseKinematics(aligned_pos=pose, body_template=NMF_TEMPLATE, log_level="ERROR")
    live = hk.compute_head_angles()
    return {"R_head": pose["R_head"], "L_head": pose["L_head"], "Neck": pose["Neck"],
            "names": np.array(names),
            "shipped": np.stack([gold[n] for n in names], axis=1),
            "ref_run": np.stack([live[n] for n in names], axis=1),
            "rest_head_pitch": np.asarray(hk.rest_head_pitch, dtype=np.float64).reshape(-1)[:1],
            "rest_antenna_pitch": np.asarray(hk.rest_antenna_pitch, dtype=np.float64).reshape(-1)[:1]}


def gen_anipose_raw_cut():
    """Alignment row incl. antennae: first 1500 frames of the shipped converted_dict.pkl (un-aligned
    segments) and the reference's AlignPose.align_pose output on exactly that cut."""
    import_reference()
    from seqikpy.alignment import AlignPose
    from seqikpy.data import NMF_TEMPLATE
    raw = load_pickle(os.path.join(ANIPOSE, "converted_dict.pkl"))
    cut = {k: np.ascontiguousarray(v[:1500]) for k, v in raw.items()}
    al = AlignPose(pose_data_dict=cut, legs_list=["RF", "LF"], include_claw=False, body_template=NMF_TEMPLATE,
                   log_level="ERROR")
    aligned = al.align_pose()
    out = {f"raw_{k}": v for k, v in cut.items()}
    out.update({f"aligned_{k}": np.asarray(v) for k, v in aligned.items()})
    out["segments"] = np.array(list(cut.keys()))
    return out


def gen_df3d_align_pins():
    """Pins held by the reference itself (no run of anything here): examples/seqikpy_locomotion.ipynb cell 2 cuts frames
    300:400 of the df3d recording, cell 6 aligns them and its STORED OUTPUT prints the six `find_scale_leg` results
    (seqikpy/alignment.py:417-423); data/df3d_pose_result__210902_PR_Fly1/pose3d_aligned.pkl is what that cell
    exported.  Stored: the cut's un-aligned key points, the six printed floats, the shipped aligned arrays."""
    import json
    import re
    import_reference()
    from seqikpy.alignment import convert_from_df3dpp_to_dict
    legs = ["RF", "RM", "RH", "LF", "LM", "LH"]
    raw = load_pickle(os.path.join(DF3D, "pose_result__210902_PR_Fly1_aligned.pkl"))
    converted = convert_from_df3dpp_to_dict(raw)
    shipped = load_pickle(os.path.join(DF3D, "pose3d_aligned.pkl"))
    with open(os.path.join(REFERENCE_ROOT, "examples", "seqikpy_locomotion.ipynb")) as fh:
        nb = json.load(fh)
    printed = {}
    for cell in nb["cells"]:
        for o in cell.get("outputs", []):
            for m in re.finditer(r"Scale factor for (\w\w) leg: ([0-9.eE+-]+)", "".join(o.get("text", []))):
                printed[m.group(1)] = float(m.group(2))
    assert sorted(printed) == sorted(legs), printed
    out = {"legs": np.array(legs), "frames": np.array([300, 400]),
           "printed_scale_factors": np.array([printed[l] for l in legs], dtype=np.float64)}
    for leg in legs:
        out[f"{leg}_raw"] = np.ascontiguousarray(converted[f"{leg}_leg"][300:400], dtype=np.float64)
        out[f"{leg}_shipped_aligned"] = np.ascontiguousarray(shipped[f"{leg}_leg"], dtype=np.float64)
    return out


def gen_df3d_notebook_cell16():
    """Image data held by the reference (nothing is run or imported): the PNG that examples/seqikpy_locomotion.ipynb keeps as
    the output of its cell 16 -- `leg_joint_angles` of cell 14 (LegInvKinSeq.run_ik_and_fk over real ikpy on frames 300:400 of
    the df3d recording, all six legs) plotted in degrees, 3 x 2 axes, figsize (9, 7), dpi 200, lw 2.  Returned as bytes."""
    import base64
    import json
    with open(os.path.join(REFERENCE_ROOT, "examples", "seqikpy_locomotion.ipynb")) as fh:
        nb = json.load(fh)
    cell = nb["cells"][16]
    assert cell["cell_type"] == "code" and "leg_joint_angles[f\"Angle_{leg_name}_{angle_name}\"]" in "".join(cell["source"])
    pngs = [o["data"]["image/png"] for o in cell["outputs"] if "image/png" in o.get("data", {})]
    assert len(pngs) == 1
    return base64.b64decode(pngs[0])


GENERATORS = {
    "anipose_shipped": gen_anipose_shipped,
    "anipose_scipy_cut": gen_anipose_scipy_cut,
    "df3d_100": gen_df3d_100,
    "df3d_1000": gen_df3d_1000,
    "generic_rf_100": gen_generic_rf_100,
    "anipose_head": gen_anipose_head,
    "anipose_raw_cut": gen_anipose_raw_cut,
    "df3d_align_pins": gen_df3d_align_pins,
    "df3d_notebook_cell16": gen_df3d_notebook_cell16,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        data = fn()
        if isinstance(data, bytes):          # an image the reference holds, kept as it is
            path = os.path.join(OUT, name + ".png")
            with open(path, "wb") as fh:
                fh.write(data)
        else:
            path = os.path.join(OUT, name + ".npz")
            np.savez_compressed(path, **data)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
