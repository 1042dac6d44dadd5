#!/usr/bin/env python3
"""Parity report of the HIP path (through the C ABI, on the GPU) against every committed fixture:

  * shipped pipeline outputs of the reference (leg_joint_angles.pkl / forward_kinematics.pkl / head_joint_angles.pkl)
  * the reference's unmodified source run in the build container (ikpy stand-in + real scipy)
  * the C oracle (bit for bit)

Prints one JSON document: per recording / leg / joint  max |d theta|, the number of frames above the 1e-4 rad bar
and where they are (the documented degenerate LF episode, frames 280-301 of the anipose recording).

    python tests/tools/parity_report.py > profiles/rNN_parity_report.json          (needs a GPU)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402

from oracle import c_oracle  # noqa: E402  (checker only)
from seqikpy_amd import _lib  # noqa: E402

DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
TOL = 1e-4


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def params(z, legs):
    return [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]


def compare(got, want):
    err = np.abs(got - want)
    bad = np.where(err.max(1) >= TOL)[0]
    return {"max_abs_per_joint": {d: float(err[:, i].max()) for i, d in enumerate(DOFS)},
            "max_abs": float(err.max()), "median_abs": float(np.median(err)),
            "frames_ge_1e-4": int(len(bad)),
            "frames_ge_1e-4_span": [int(bad.min()), int(bad.max())] if len(bad) else None,
            "max_abs_outside_those_frames": float(np.delete(err, bad, axis=0).max()) if len(bad) < len(err) else None}


def legs_report(name, against):
    z = golden(name)
    legs = [str(l) for l in z["legs"]]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    out = _lib.solve_seq(pose, params(z, legs), want_fk=True, want_diag=True)
    # the other launch paths of the same call (round 2): stage pipeline and fused lane-per-chain kernel must give the
    # same bits as the per-stage kernels above; frame chunks (automatic parameters) are compared like the serial walk
    piped = _lib.solve_seq(pose, params(z, legs), want_fk=True, pipeline=2)
    fused = _lib.solve_seq(pose, params(z, legs), want_fk=True, pipeline=1)
    chunked = _lib.solve_seq(pose, params(z, legs), want_fk=True, frame_chunk=-1)
    rep = {"frames": int(pose.shape[2]), "against": against,
           "stage_pipeline_and_fused_kernel_equal_the_per_stage_kernels_bit_for_bit":
               bool(np.array_equal(piped["angles"], out["angles"]) and np.array_equal(piped["fk"], out["fk"]) and
                    np.array_equal(fused["angles"], out["angles"]) and np.array_equal(fused["fk"], out["fk"])),
           "frame_chunks": {"chunk_stats": {k: v for k, v in chunked["chunk_stats"].items() if v},
                            "max_abs_vs_serial_walk": float(np.abs(chunked["angles"] - out["angles"]).max()),
                            "leg_frames_ge_1e-4_vs_serial_walk": int((np.abs(chunked["angles"] - out["angles"]).max(-1) >= TOL).sum())},
           "legs": {}}
    for i, leg in enumerate(legs):
        ref = c_oracle.seq_leg(z[f"{leg}_pose"], z[f"{leg}_seg"], z[f"{leg}_bounds"], z[f"{leg}_seeds"])
        r = compare(out["angles"][0, i], z[f"{leg}_angles"])
        r["frame_chunks"] = {k: v for k, v in compare(chunked["angles"][0, i], z[f"{leg}_angles"]).items()
                             if k in ("max_abs", "frames_ge_1e-4", "frames_ge_1e-4_span", "max_abs_outside_those_frames")}
        r["equals_c_oracle_bit_for_bit"] = bool(np.array_equal(out["angles"][0, i], ref["angles"]) and
                                                np.array_equal(out["fk"][0, i], ref["fk"]) and
                                                np.array_equal(out["nfev"][0, i], ref["nfev"]) and
                                                np.array_equal(out["status"][0, i], ref["status"]))
        if f"{leg}_nfev" in z.files:
            r["nfev_equal_to_scipy_fraction_per_stage"] = [float((out["nfev"][0, i][:, k] == z[f"{leg}_nfev"][:, k]).mean())
                                                           for k in range(4)]
        fk_key = f"{leg}_fk" if f"{leg}_fk" in z.files else None
        if fk_key:
            r["fk_max_abs"] = float(np.abs(out["fk"][0, i] - z[fk_key]).max())
        elif f"{leg}_fk_cut" in z.files:
            cut = z["fk_frames"]
            r["fk_max_abs_on_cut_frames"] = float(np.abs(out["fk"][0, i][cut] - z[f"{leg}_fk_cut"]).max())
        rep["legs"][leg] = r
    return rep


def head_report():
    from seqikpy_amd.head_inverse_kinematics import HeadInverseKinematics
    z = golden("anipose_head")
    aligned = {"R_head": z["R_head"], "L_head": z["L_head"], "Neck": z["Neck"]}
    from seqikpy_amd.data import NMF_TEMPLATE
    hk = HeadInverseKinematics(aligned_pos=aligned, body_template=NMF_TEMPLATE, log_level="ERROR")
    ang = hk.compute_head_angles()
    names = [str(n) for n in z["names"]]
    got = np.stack([ang[n] for n in names], 1)
    return {"frames": int(got.shape[0]),
            "max_abs_vs_shipped_per_angle": {n: float(np.abs(got[:, i] - z["shipped"][:, i]).max()) for i, n in enumerate(names)},
            "max_abs_vs_reference_run": float(np.abs(got - z["ref_run"]).max())}


def main():
    if _lib.load().seqik_device_count() < 1:
        raise SystemExit("parity_report.py needs a GPU")
    rep = {"tolerance_rad": TOL,
           "note": "LF frames 284-301 of the anipose recording are a kinematic-singularity episode in which the reference "
                   "itself is not reproducible (DESIGN.md 2, profiles/r02_perturbation_report.json); every other frame must "
                   "be below the tolerance.",
           "anipose_6000_vs_shipped_outputs": legs_report("anipose_shipped", "reference's shipped leg_joint_angles.pkl / forward_kinematics.pkl"),
           "anipose_330_vs_reference_source_run": legs_report("anipose_scipy_cut", "reference source over real scipy, build container"),
           "df3d_100_vs_reference_source_run": legs_report("df3d_100", "reference source over real scipy, build container"),
           "df3d_1000_vs_reference_source_run": legs_report("df3d_1000", "reference source over real scipy, build container"),
           "head_antenna_6000": head_report()}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
