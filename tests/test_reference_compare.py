"""CPU tier, build container only: compare against the reference's own source (imported from
/root/reference over the ikpy shim).  Skipped where the reference is absent (GPU box)."""
import os
import sys

import numpy as np
import pytest

from conftest import DOFS, ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "oracle"))
from ref_import import import_reference, reference_available  # noqa: E402

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference checkout not present")


def test_constants_equal_the_reference():
    import_reference()
    import seqikpy.data as rd
    from seqikpy.utils import calculate_body_size as ref_body
    from seqikpy_amd import data, utils
    for mine, ref in ((data.BOUNDS, rd.BOUNDS), (data.NMF_TEMPLATE, rd.NMF_TEMPLATE)):
        assert set(mine) == set(ref)
        for k in mine:
            assert np.array_equal(np.asarray(mine[k]), np.asarray(ref[k])), k
    for leg in rd.INITIAL_ANGLES:
        for st in rd.INITIAL_ANGLES[leg]:
            assert np.array_equal(data.INITIAL_ANGLES[leg][st], rd.INITIAL_ANGLES[leg][st])
    a, b = utils.calculate_body_size(data.NMF_TEMPLATE, ["RF", "LF"]), ref_body(rd.NMF_TEMPLATE, ["RF", "LF"])
    assert set(a) == set(b) and all(a[k] == b[k] for k in a)


def test_locomotion_constants_equal_the_reference_example():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from gen_golden import locomotion_constants
    from seqikpy_amd import data
    template, init, bounds = locomotion_constants()
    for mine, ref in ((data.TEMPLATE_NMF_LOCOMOTION, template), (data.BOUNDS_LOCOMOTION, bounds)):
        assert set(mine) == set(ref)
        for k in mine:
            assert np.array_equal(np.asarray(mine[k]), np.asarray(ref[k])), k
    for leg in init:
        for st in init[leg]:
            assert np.array_equal(data.INITIAL_ANGLES_LOCOMOTION[leg][st], init[leg][st])


def test_oracle_vs_reference_source_live(oracle):
    """Run the reference's LegInvKinSeq here (shim + real scipy) on a short cut and compare."""
    import_reference()
    from seqikpy.data import BOUNDS, INITIAL_ANGLES
    from seqikpy.kinematic_chain import KinematicChainSeq
    from seqikpy.leg_inverse_kinematics import LegInvKinSeq
    z = load_golden("anipose_shipped")
    pose = z["RF_pose"][:25]
    ik = LegInvKinSeq({"RF_leg": pose}, KinematicChainSeq(BOUNDS, ["RF"]), INITIAL_ANGLES, log_level="ERROR")
    ang, fk = ik.run_ik_and_fk(hide_progress_bar=True)
    ref = np.stack([ang[f"Angle_RF_{d}"] for d in DOFS], 1)
    r = oracle.seq_leg(pose, z["RF_seg"], z["RF_bounds"], z["RF_seeds"])
    assert np.abs(r["angles"] - ref).max() < 1e-4
    assert np.abs(r["fk"] - fk["RF_leg"]).max() < 1e-4
    assert list(ang.keys()) == [f"Angle_RF_{d}" for d in DOFS]


def test_chain_link_names_equal_the_reference():
    import_reference()
    from seqikpy.data import BOUNDS
    from seqikpy.kinematic_chain import KinematicChainSeq as RefSeq
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    z = load_golden("anipose_shipped")
    angles = {f"Angle_LF_{d}": z["LF_angles"][:, i] for i, d in enumerate(DOFS)}
    for stage in (1, 2, 3, 4):
        mine = KinematicChainSeq(BOUNDS, ["LF"]).create_leg_chain("LF", stage=stage, angles=angles, t=3)
        ref = RefSeq(BOUNDS, ["LF"]).create_leg_chain("LF", stage=stage, angles=angles, t=3)
        assert [l.name for l in mine.links] == [l.name for l in ref.links]
        assert mine.name == ref.name
        for a, b in zip(mine.links, ref.links):
            assert a.bounds == tuple(b.bounds)
