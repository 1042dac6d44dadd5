"""Left / right mirror symmetry of the four legs no reference-held output pins (round-3 review, item 8; the tool with real
scipy and the reference's source over the IKPy stand-in is tests/tools/mirror_report.py -> profiles/r04_mirror_check.json).

The shipped outputs of REAL IKPy cover RF and LF with `BOUNDS`; the middle and hind legs and `BOUNDS_LOCOMOTION` are
pinned through the build's own IKPy stand-in.  The fly and its tables are mirror-symmetric, so reflecting the recording at
the sagittal plane and swapping R <-> L must mirror the angles: (yaw, roll, CTr_roll) change sign, the pitches do not.  A
sign or axis mistake in how the mirrored limits of a left leg are applied breaks that by the size of the angle."""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import mirror_report as mr  # noqa: E402


def _locomotion(leg):
    from oracle import c_oracle
    from seqikpy_amd import data, utils
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, [leg])
    return c_oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)


def test_locomotion_tables_are_mirror_images():
    assert mr.tables_are_mirror_images()


def test_c_restatement_mirrors_bit_for_bit_on_all_six_legs(oracle):
    z = load_golden("df3d_1000")
    for leg in [str(l) for l in z["legs"]]:
        partner = mr.PARTNER[leg]
        direct = oracle.seq_leg(z[f"{partner}_pose"], *_locomotion(partner), want_fk=False)["angles"] * mr.FLIP
        mirrored = oracle.seq_leg(mr.mirror_pose(z[f"{partner}_pose"]), *_locomotion(leg), want_fk=False)["angles"]
        assert np.array_equal(direct, mirrored), leg


def test_real_scipy_mirrors_to_the_noise_floor_on_a_middle_and_a_hind_leg():
    warnings.filterwarnings("ignore")
    from oracle import scipy_oracle as so
    from seqikpy_amd import data, utils
    z = load_golden("df3d_1000")
    for leg in ("LM", "RH"):
        partner = mr.PARTNER[leg]
        run = lambda l, pose: so.seq_leg(pose[:25], l, data.BOUNDS_LOCOMOTION, utils.calculate_body_size(  # noqa: E731
            data.TEMPLATE_NMF_LOCOMOTION, [l]), data.INITIAL_ANGLES_LOCOMOTION)["angles"]
        direct = run(partner, z[f"{partner}_pose"]) * mr.FLIP
        mirrored = run(leg, mr.mirror_pose(z[f"{partner}_pose"]))
        assert np.abs(direct - mirrored).max() < 1e-4, leg


@pytest.mark.gpu
def test_hip_mirrors_bit_for_bit_on_all_six_legs(hiplib):
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(*_locomotion(l)) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    mirrored_pose = np.stack([mr.mirror_pose(z[f"{mr.PARTNER[l]}_pose"]) for l in legs])[None]
    a = hiplib.solve_seq(pose, params, want_fk=False)["angles"][0]
    b = hiplib.solve_seq(mirrored_pose, params, want_fk=False)["angles"][0]
    for i, leg in enumerate(legs):
        assert np.array_equal(b[i], a[legs.index(mr.PARTNER[leg])] * mr.FLIP), leg
