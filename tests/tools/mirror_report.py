#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- left / right mirror check of the four legs no reference-held output pins (round-3 review, item 8).

The only outputs of REAL IKPy this build can compare with are the shipped anipose files: RF and LF, `BOUNDS`.  The middle and
hind legs, `BOUNDS_LOCOMOTION` and the locomotion template are pinned through runs of the reference's unmodified source over
the build's own IKPy stand-in (oracle/shim/ikpy) -- a mistake shared by the stand-in and the restatement on those legs
would be invisible.  The cheapest cross-check that does not need IKPy: the fly is mirror-symmetric.  Reflecting a
recording at the sagittal plane (y -> -y) and swapping R <-> L legs must give the mirrored angles
    (yaw, pitch, roll, CTr_pitch, CTr_roll, FTi_pitch, TiTa_pitch) -> (-yaw, pitch, -roll, CTr_pitch, -CTr_roll, FTi_pitch, TiTa_pitch)
(rotations about x and z change sign, rotations about y do not) because BOUNDS_LOCOMOTION, INITIAL_ANGLES_LOCOMOTION and
the template ARE mirror images of each other (checked first).  A sign or axis mistake in how an implementation applies
the mirrored limits of a left leg shows up as an asymmetry of the order of the angle itself.  Run for
    C   the C restatement (oracle/seqik_oracle.c)
    S   real scipy over the build's own chain tables (oracle/scipy_oracle.py)
    R   the reference's own LegInvKinSeq / KinematicChainSeq source over the IKPy stand-in (only where /root/reference exists)
on the df3d locomotion recording, all six legs.  Floating point is sign-symmetric except where scipy breaks ties by sign
(`h = eps * sign(x)` with sign(0) = +1), so C is expected to mirror to the last bit almost everywhere and S / R to the
noise of LAPACK's null-space vectors (~1e-5 rad).

    python tests/tools/mirror_report.py [--frames 300] > profiles/r04_mirror_check.json        (CPU only)
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402

FLIP = np.array([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0, 1.0])     # DOF order of the ABI
PARTNER = {"RF": "LF", "LF": "RF", "RM": "LM", "LM": "RM", "RH": "LH", "LH": "RH"}
DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]


def mirror_pose(pose):
    out = pose.copy()
    out[..., 1] = -out[..., 1]
    return out


def tables_are_mirror_images():
    from seqikpy_amd import data
    ok = True
    for r, l in (("RF", "LF"), ("RM", "LM"), ("RH", "LH")):
        for d, sg in zip(DOFS, FLIP):
            br, bl = np.array(data.BOUNDS_LOCOMOTION[f"{r}_{d}"]), np.array(data.BOUNDS_LOCOMOTION[f"{l}_{d}"])
            ok &= bool(np.array_equal(bl, br if sg > 0 else -br[::-1]))
        for seg in ("Coxa", "Femur", "Tibia", "Tarsus", "Claw"):
            a, b = np.array(data.TEMPLATE_NMF_LOCOMOTION[f"{r}_{seg}"]), np.array(data.TEMPLATE_NMF_LOCOMOTION[f"{l}_{seg}"])
            ok &= bool(np.allclose(a * [1, -1, 1], b))
        # seeds: stage chains list (base, yaw, pitch, [roll, CTr_pitch, [CTr_roll, FTi, [TiTa, [claw]]]]) -- see data.py
        for k, links in ((1, ["b", "ThC_yaw", "ThC_pitch", "CTr_pitch"]), (4, ["b", "ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch", "c"])):
            sr, sl = np.array(data.INITIAL_ANGLES_LOCOMOTION[r][f"stage_{k}"]), np.array(data.INITIAL_ANGLES_LOCOMOTION[l][f"stage_{k}"])
            sg = np.array([1.0 if n in ("b", "c") else FLIP[DOFS.index(n)] for n in links])
            ok &= bool(np.allclose(sr * sg, sl))
    return ok


def task(args):
    impl, leg, pose, n = args
    warnings.filterwarnings("ignore")
    from seqikpy_amd import data, utils
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, [leg])
    if impl == "C":
        from oracle import c_oracle
        seg, b, seeds = c_oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
        return impl, leg, c_oracle.seq_leg(pose[:n], seg, b, seeds, want_fk=False)["angles"]
    if impl == "S":
        from oracle import scipy_oracle as so
        return impl, leg, so.seq_leg(pose[:n], leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)["angles"]
    from oracle import ref_import
    ref = ref_import.import_reference()
    kc = ref.kinematic_chain.KinematicChainSeq(bounds_dof=data.BOUNDS_LOCOMOTION, legs_list=[leg], body_size=body)
    ik = ref.leg_inverse_kinematics.LegInvKinSeq(aligned_pos={f"{leg}_leg": pose[:n]}, kinematic_chain_class=kc,
                                                 initial_angles=data.INITIAL_ANGLES_LOCOMOTION, log_level="ERROR")
    ang, _ = ik.run_ik_and_fk(hide_progress_bar=True)
    return impl, leg, np.stack([ang[f"Angle_{leg}_{d}"] for d in DOFS], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--processes", type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args()
    from oracle import ref_import
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    impls = ["C", "S"] + (["R"] if ref_import.reference_available() else [])
    n = {"C": min(1000, max(a.frames, 1000)), "S": a.frames, "R": a.frames}
    tasks = []
    for impl in impls:
        for leg in legs:
            tasks.append((impl, leg, np.ascontiguousarray(z[f"{leg}_pose"]), n[impl]))                       # the recording
            tasks.append((impl + "m", leg, mirror_pose(np.ascontiguousarray(z[f"{PARTNER[leg]}_pose"])), n[impl]))  # its mirror image
    with mp.get_context("fork").Pool(processes=a.processes) as pool:
        res = {(k, leg): ang for k, leg, ang in pool.map(_dispatch, tasks, chunksize=1)}
    out = {"what": "angles(leg; mirrored recording of the partner leg) against mirror(angles(partner leg; recording)), df3d locomotion "
                   "recording, BOUNDS_LOCOMOTION / INITIAL_ANGLES_LOCOMOTION / TEMPLATE_NMF_LOCOMOTION",
           "tables_are_mirror_images": tables_are_mirror_images(), "frames": n,
           "implementations": {"C": "C restatement", "S": "real scipy over the build's own chain tables",
                               "R": "the reference's source over the IKPy stand-in"}, "legs": {}}
    for leg in legs:
        e = {}
        for impl in impls:
            direct = res[(impl, PARTNER[leg])] * FLIP       # mirror of the partner's angles on the real recording
            mirrored = res[(impl + "m", leg)]               # this leg on the mirrored recording of its partner
            d = np.abs(direct - mirrored)
            e[impl] = {"max_abs_asymmetry_rad": float(d.max()), "frames_over_1e-4": int((d.max(1) > 1e-4).sum()),
                       "bit_identical": bool(np.array_equal(direct, mirrored)),
                       "worst_dof": DOFS[int(np.unravel_index(d.argmax(), d.shape)[1])]}
        out["legs"][leg] = e
    print(json.dumps(out, indent=1))


def _dispatch(t):
    impl, leg, pose, n = t
    k, lg, ang = task((impl.rstrip("m"), leg, pose, n))
    return impl, lg, ang


if __name__ == "__main__":
    main()
