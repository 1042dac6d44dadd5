"""the slab-wise on-disk format of streamed results and the loader contract of the reference's
consumers (tests/loader_contract.py restates seqikpy/visualization.py:191-213, 443-492 and utils.py:235-245)."""
import os
import pickle

import numpy as np
import pytest

import loader_contract as lc
from conftest import DOFS, load_golden


def _fake_results(S, L, N, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((S, L, N, 7)), rng.standard_normal((S, L, N, 9, 3))


@pytest.mark.parametrize("S", [1, 3])
def test_slab_round_trip_and_reference_layout(tmp_path, S):
    from seqikpy_amd.slab_format import SlabReader, SlabWriter
    legs = ["RF", "LH"]
    ang, fk = _fake_results(S, 2, 96)
    with SlabWriter(tmp_path / "run", legs, 32, n_seq=S) as w:
        for k in range(3):
            w.write(ang[:, :, 32 * k:32 * (k + 1)], fk[:, :, 32 * k:32 * (k + 1)])
    with pytest.raises(FileExistsError):
        SlabWriter(tmp_path / "run", legs, 32, n_seq=S)
    r = SlabReader(tmp_path / "run")
    assert len(r) == 3 and r.manifest["legs"] == legs and r.manifest["dofs"] == DOFS
    ja, fkd = r.load_slab(1)
    assert list(ja) == [f"Angle_{l}_{d}" for l in legs for d in DOFS] and list(fkd) == ["RF_leg", "LH_leg"]
    want = ang[:, 1, 32:64, 3]
    assert np.array_equal(ja["Angle_LH_CTr_pitch"], want[0] if S == 1 else want)
    ja_all, fk_all = r.load_all()
    assert np.array_equal(fk_all["RF_leg"], fk[0, 0] if S == 1 else fk[:, 0])
    assert ja_all["Angle_RF_ThC_yaw"].shape == ((96,) if S == 1 else (S, 96))
    if S == 1:  # every slab on its own, and the joined set, satisfy the consumers' contract
        lc.check_joint_angles(ja, legs, 32, with_head=False)
        lc.check_points3d(fkd, 32, leg_points=9)
        r.to_pickles(tmp_path / "export")
        saved = pickle.load(open(tmp_path / "export" / "leg_joint_angles.pkl", "rb"))
        lc.check_joint_angles(saved, legs, 96, with_head=False)
        lc.check_points3d(pickle.load(open(tmp_path / "export" / "forward_kinematics.pkl", "rb")), 96, leg_points=9)


def test_slabs_of_different_recordings_stack(tmp_path):
    from seqikpy_amd.slab_format import SlabReader, SlabWriter
    ang, _ = _fake_results(4, 1, 16)
    with SlabWriter(tmp_path / "many", ["RM"], 16, n_seq=1, in_time=False) as w:
        for s in range(4):
            w.write(ang[s:s + 1])
    ja, fk = SlabReader(tmp_path / "many").load_all()
    assert fk == {} and np.array_equal(ja["Angle_RM_FTi_pitch"], ang[:, 0, :, 5])


def test_writer_rejects_wrong_shapes(tmp_path):
    from seqikpy_amd.slab_format import SlabWriter
    w = SlabWriter(tmp_path / "bad", ["RF"], 8)
    with pytest.raises(ValueError):
        w.write(np.zeros((1, 2, 8, 7)))
    with pytest.raises(ValueError):
        w.write(np.zeros((1, 1, 8, 7)), np.zeros((1, 1, 8, 9, 2)))
    with pytest.raises(ValueError):
        w.write(np.zeros((2, 1, 8, 7)))


def test_contract_accepts_the_reference_shipped_layout():
    """The restated contract holds for arrays shaped like the reference's own shipped outputs (fixture cut)."""
    z = load_golden("anipose_shipped")
    ja = {f"Angle_{l}_{d}": np.ascontiguousarray(z[f"{l}_angles"][:, i]) for l in ("RF", "LF") for i, d in enumerate(DOFS)}
    lc.check_joint_angles(ja, ["RF", "LF"], 6000, with_head=False)
    lc.check_points3d({"RF_leg": z["RF_pose"], "LF_leg": z["LF_pose"], "Neck": np.zeros((1, 1, 3))}, 6000, leg_points=5)


@pytest.mark.gpu
def test_streamed_recording_to_slabs_equals_one_call(tmp_path, hiplib):
    from seqikpy_amd.slab_format import stream_recording_to_slabs
    z = load_golden("df3d_1000")
    legs = ["RF", "LM", "LH"]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"][:960] for l in legs])[None]
    r = stream_recording_to_slabs(pose, params, legs, tmp_path / "rec", slab_frames=120)
    assert len(r) == 8
    ja, fk = r.load_all()
    one = hiplib.solve_seq(pose, params)
    for li, l in enumerate(legs):
        assert np.array_equal(fk[f"{l}_leg"], one["fk"][0, li])
        for d, dof in enumerate(DOFS):
            assert np.array_equal(ja[f"Angle_{l}_{dof}"], one["angles"][0, li, :, d])
    lc.check_joint_angles(ja, legs, 960, with_head=False)
    lc.check_points3d(fk, 960, leg_points=9)


@pytest.mark.gpu
def test_pinned_array_outlives_its_holder(hiplib):
    """the idiom PinnedArray(...).array must not leave a dangling array."""
    import gc
    from seqikpy_amd.streaming import PinnedArray, pinned_array
    a = PinnedArray((4, 1024)).array
    gc.collect()
    a[:] = 3.0
    v = pinned_array((1 << 16,))[100:200]
    gc.collect()
    v[:] = 7.0
    assert float(a.sum()) == 3.0 * 4096 and float(v.sum()) == 700.0
