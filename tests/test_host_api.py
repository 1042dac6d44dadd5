"""CPU tier: host-side mirror of the reference interface -- chain factories, constants,
argument checks (mirrors the reference's tests/test_kin_chain.py: link names per stage,
ValueError on bad leg / stage)."""
import os

import numpy as np
import pytest

from conftest import DOFS, load_golden

from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES, NMF_TEMPLATE
from seqikpy_amd.kinematic_chain import KinematicChainGeneric, KinematicChainSeq
from seqikpy_amd.leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq
from seqikpy_amd.utils import calculate_body_size


@pytest.fixture()
def angles():
    z = load_golden("anipose_shipped")
    return {f"Angle_{leg}_{d}": z[f"{leg}_angles"][:, i] for leg in ("RF", "LF") for i, d in enumerate(DOFS)}


def test_chain_classes_have_attributes():
    for cls in (KinematicChainSeq, KinematicChainGeneric):
        kc = cls(bounds_dof=BOUNDS, legs_list=["RF", "LF"], body_size=None)
        assert hasattr(kc, "body_size") and hasattr(kc, "bounds_dof")
        assert kc.body_size["RF_Coxa"] == pytest.approx(0.40)
        assert kc.body_size["RF"] == pytest.approx(2.26)


def test_seq_chain_link_names_per_stage(angles):
    kc = KinematicChainSeq(bounds_dof=BOUNDS, legs_list=["RF", "LF"], body_size=None)
    expect = {
        1: ["Base link", "RF_ThC_yaw", "RF_ThC_pitch", "RF_CTr_pitch"],
        2: ["Base link", "RF_ThC_yaw", "RF_ThC_pitch", "RF_ThC_roll", "RF_CTr_pitch", "RF_FTi_pitch"],
        3: ["Base link", "RF_ThC_yaw", "RF_ThC_pitch", "RF_ThC_roll", "RF_CTr_pitch", "RF_CTr_roll",
            "RF_FTi_pitch", "RF_TiTa_pitch"],
        4: ["Base link", "RF_ThC_yaw", "RF_ThC_pitch", "RF_ThC_roll", "RF_CTr_pitch", "RF_CTr_roll",
            "RF_FTi_pitch", "RF_TiTa_pitch", "RF_Claw"],
    }
    for stage, names in expect.items():
        chain = kc.create_leg_chain(leg_name="RF", stage=stage, angles=angles, t=0)
        assert [l.name for l in chain.links] == names
        assert chain.name == f"chain_stage_{stage}"
    # fixed links carry the earlier angles of frame t
    chain = kc.create_leg_chain(leg_name="RF", stage=3, angles=angles, t=5)
    assert chain.links[1].joint_type == "fixed"
    assert chain.links[1].origin_orientation[0] == angles["Angle_RF_ThC_yaw"][5]
    assert chain.links[4].origin_translation[2] == -kc.body_size["RF_Coxa"]
    assert chain.links[5].joint_type == "revolute" and tuple(chain.links[5].rotation) == (0, 0, 1)


def test_generic_chain_link_names():
    kc = KinematicChainGeneric(bounds_dof=BOUNDS, legs_list=["RF", "LF"], body_size=None)
    chain = kc.create_leg_chain(leg_name="LF")
    assert [l.name for l in chain.links] == ["Base link", "LF_ThC_roll", "LF_ThC_yaw", "LF_ThC_pitch",
                                             "LF_CTr_pitch", "LF_CTr_roll", "LF_FTi_pitch", "LF_TiTa_pitch", "LF_Claw"]


def test_chain_factory_errors(angles):
    kc = KinematicChainSeq(bounds_dof=BOUNDS, legs_list=["RF", "LF"], body_size=None)
    with pytest.raises(ValueError):
        kc.create_leg_chain(leg_name="XX", stage=1)
    with pytest.raises(ValueError):
        kc.create_leg_chain(leg_name="RF", stage=5, angles=angles, t=0)
    with pytest.raises(ValueError):
        KinematicChainGeneric(BOUNDS, ["RF"]).create_leg_chain(leg_name="R1")
    with pytest.raises(NameError):
        calculate_body_size(NMF_TEMPLATE, ["RF", "ZZ"])


def test_run_ik_and_fk_validates_stages_before_any_launch():
    pose = {"RF_leg": np.zeros((3, 5, 3))}
    ik = LegInvKinSeq(pose, KinematicChainSeq(BOUNDS, ["RF"]), INITIAL_ANGLES, log_level="ERROR")
    for stages in ([1, 3], [1, 2, 3, 4, 5], [2, 1]):
        with pytest.raises(ValueError, match="Maximum stage number is 4"):
            ik.run_ik_and_fk(stages=stages)
    with pytest.raises(ValueError, match="not valid"):
        ik.calculate_ik_stage(np.zeros((3, 3)), np.zeros(3), INITIAL_ANGLES["RF"]["stage_1"], "XX", stage=1)
    with pytest.raises(ValueError, match="between 1 and 4"):
        ik.calculate_ik_stage(np.zeros((3, 3)), np.zeros(3), INITIAL_ANGLES["RF"]["stage_1"], "RF", stage=7)
    assert ik.initial_angles is INITIAL_ANGLES and ik.joint_angles_dict == {}


def test_later_stage_without_earlier_angles_raises_keyerror():
    pose = {"RF_leg": np.zeros((3, 5, 3))}
    ik = LegInvKinSeq(pose, KinematicChainSeq(BOUNDS, ["RF"]), INITIAL_ANGLES, log_level="ERROR")
    with pytest.raises(KeyError):
        ik.run_ik_and_fk(stages=[2, 3])


def test_calculate_fk_matches_oracle(oracle, angles):
    kc = KinematicChainSeq(bounds_dof=BOUNDS, legs_list=["RF", "LF"], body_size=None)
    ik = LegInvKinSeq({}, kc, INITIAL_ANGLES, log_level="ERROR")
    seg, b, _ = oracle.leg_params("RF", BOUNDS, kc.body_size, INITIAL_ANGLES)
    z = load_golden("anipose_shipped")
    t = 17
    chain = kc.create_leg_chain(leg_name="RF", stage=4, angles=angles, t=t)
    q = np.concatenate([[0.0], z["RF_angles"][t], [0.0]])
    got = ik.calculate_fk(chain, q)
    want = oracle.stage_fk(4, seg, b, z["RF_angles"][t], q)
    assert np.abs(got - want).max() < 1e-14
    assert ik.get_scale_factor(np.array([[0, 0, 0], [0, 0, 1.0], [0, 0, 3.0]]), 6.0) == pytest.approx(2.0)


def test_synthetic_fk_matches_oracle(oracle):
    from seqikpy_amd import data, synthetic
    body = calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, data.LEGS)
    pose, theta = synthetic.synthetic_pose(1, 5, ["RM"], data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                           noise=0.0, return_theta=True)
    seg, b, _ = oracle.leg_params("RM", data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
    for t in range(5):
        q = np.concatenate([[0.0], theta[0, 0, t], [0.0]])
        pos = oracle.stage_fk(4, seg, b, theta[0, 0, t], q) + data.TEMPLATE_NMF_LOCOMOTION["RM_Coxa"]
        assert np.abs(pos[[0, 4, 6, 7, 8]] - pose[0, 0, t]).max() < 1e-12


def test_many_recordings_bucketing_and_padding(monkeypatch):
    """batch.run_ik_and_fk_many: recordings are bucketed by length (optionally rounded up to a multiple, padded by
    repeating the last frame), one library call per bucket, results cut back and returned in input order."""
    from seqikpy_amd import _lib, batch, data
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    calls = []

    def fake_solve(pose, legs, want_fk=True, device=0, affine=None, frame_chunk=0, **_):
        assert frame_chunk == -1   # the default of run_ik_and_fk_many is run_ik_and_fk's: automatic frame chunks
        calls.append(pose.shape)
        S, L, N = pose.shape[:3]
        ang = np.broadcast_to(pose[..., 1, 0][..., None], (S, L, N, 7)).copy()   # echoes a key-point coordinate
        return dict(angles=ang, fk=np.zeros((S, L, N, 9, 3)), chunk_stats=dict(chunks=0), chunk_flags=None)

    monkeypatch.setattr(_lib, "solve_seq", fake_solve)
    kc = KinematicChainSeq(data.BOUNDS, ["RF", "LF"])
    rng = np.random.default_rng(0)
    recs = [{"RF_leg": rng.normal(size=(n, 5, 3)), "LF_leg": rng.normal(size=(n, 5, 3)), "Neck": np.zeros((1, 1, 3))}
            for n in (10, 7, 10, 0, 3)]
    out = batch.run_ik_and_fk_many(recs, kc)
    assert sorted(calls) == [(1, 2, 3, 5, 3), (1, 2, 7, 5, 3), (2, 2, 10, 5, 3)]
    for rec, (ang, fk) in zip(recs, out):
        n = rec["RF_leg"].shape[0]
        assert list(ang) == [f"Angle_{leg}_{d}" for leg in ("RF", "LF") for d in DOFS] and list(fk) == ["RF_leg", "LF_leg"]
        assert ang["Angle_LF_FTi_pitch"].shape == (n,) and fk["RF_leg"].shape == (n, 9, 3)
        assert np.array_equal(ang["Angle_RF_ThC_yaw"], rec["RF_leg"][:, 1, 0])
    calls.clear()
    out8 = batch.run_ik_and_fk_many(recs, kc, pad_to_multiple=8)
    assert sorted(calls) == [(2, 2, 8, 5, 3), (2, 2, 16, 5, 3)]
    assert all(np.array_equal(a[0]["Angle_LF_ThC_yaw"], b[0]["Angle_LF_ThC_yaw"]) for a, b in zip(out, out8))
    with pytest.raises(ValueError):
        batch.run_ik_and_fk_many([{"RF_leg": np.zeros((4, 5, 3)), "LF_leg": np.zeros((5, 5, 3))}], kc)


def test_host_objects_are_picklable_like_the_reference_parallel_example_needs():
    """examples/example_leg_inv_kinematics_parallel.py ships (pose_data, leg) to worker processes and builds the
    chain / IK objects there; users also pickle the objects themselves.  Nothing here may hold a library handle."""
    import pickle
    z = load_golden("df3d_100")
    kc = KinematicChainSeq(BOUNDS, ["RF", "LF"])
    ik = LegInvKinSeq({"RF_leg": z["RF_pose"], "LF_leg": z["LF_pose"]}, kc, INITIAL_ANGLES, log_level="ERROR")
    kc2 = pickle.loads(pickle.dumps(kc))
    ik2 = pickle.loads(pickle.dumps(ik))
    assert kc2.body_size == kc.body_size and kc2.bounds_dof == kc.bounds_dof
    assert np.array_equal(ik2.aligned_pos["LF_leg"], ik.aligned_pos["LF_leg"]) and ik2.joint_angles_dict == {}
    gen = pickle.loads(pickle.dumps(LegInvKinGeneric({"RF_leg": z["RF_pose"]}, KinematicChainGeneric(BOUNDS, ["RF"]),
                                                     INITIAL_ANGLES, log_level="ERROR")))
    assert gen.kinematic_chain_class.create_leg_chain("RF").links[-1].name == "RF_Claw"


def test_package_import_leaves_the_environment_alone():
    """Round-5 review, item 7: importing the package does not touch os.environ (a library must not change the process it
    is loaded into); `recommended_env()` RETURNS what a process should start with, `SEQIK_SET_ENV=1` makes the import apply
    it (explicit settings win), `runtime_env()` reports what is set."""
    import subprocess
    import sys
    from conftest import PKG_PARENT
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); os.environ.pop('SEQIK_SET_ENV', None); "
            "before = dict(os.environ); import seqikpy_amd; print(dict(os.environ) == before); "
            "print(seqikpy_amd.recommended_env()['GPU_MAX_HW_QUEUES'], seqikpy_amd.recommended_env(20)['GPU_MAX_HW_QUEUES'], "
            "seqikpy_amd.recommended_env(40)['GPU_MAX_HW_QUEUES']); "
            "os.environ['SEQIK_SET_ENV'] = '1'; import importlib; importlib.reload(seqikpy_amd); print(os.environ['GPU_MAX_HW_QUEUES']); "
            "os.environ['GPU_MAX_HW_QUEUES'] = '2'; importlib.reload(seqikpy_amd); print(os.environ['GPU_MAX_HW_QUEUES']); "
            "print(sorted(seqikpy_amd.runtime_env()))") % PKG_PARENT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "True" and out[1].split() == ["8", "22", "22"] and out[2] == "8" and out[3] == "2"
    assert "GPU_MAX_HW_QUEUES" in out[4] and "SEQIK_SET_ENV" in out[4]


def test_frame_parallel_default_and_chunk_report(monkeypatch):
    """The default of run_ik_and_fk is "auto" (verified frame chunks; round 6) unless SEQIK_FRAME_PARALLEL says serial; the
    per-leg report is built from the library's chunk statistics and per-chunk flags."""
    from seqikpy_amd import _lib
    from seqikpy_amd.leg_inverse_kinematics import chunk_report, default_frame_parallel
    monkeypatch.delenv("SEQIK_FRAME_PARALLEL", raising=False)
    assert default_frame_parallel() == "auto"
    monkeypatch.setenv("SEQIK_FRAME_PARALLEL", "auto")
    assert default_frame_parallel() == "auto"
    for v in ("0", "serial", "off", "False"):
        monkeypatch.setenv("SEQIK_FRAME_PARALLEL", v)
        assert default_frame_parallel() is False
    flags = np.zeros((2, 2, 5), np.uint8)
    flags[0, 0, 2] = _lib.CHUNK_FLAG_FAILED_FIRST | _lib.CHUNK_FLAG_REPAIRED
    flags[0, 0, 4] = _lib.CHUNK_FLAG_SWEPT
    flags[1, 1, :] = _lib.CHUNK_FLAG_SERIAL | _lib.CHUNK_FLAG_FAILED_FIRST
    out = dict(angles=np.zeros((2, 2, 37, 7)), chunk_flags=flags,
               chunk_stats=dict(chunks=20, frames_per_chunk=8, run_in_frames=4))
    rep = chunk_report(out, ["RF", "LF"], 37)
    assert rep[0]["RF"] == dict(frames_per_chunk=8, run_in_frames=4, failed_first_check=[16], frames_repaired=8 + 5,
                                walked_serially=False)     # the last chunk holds 37 - 32 = 5 frames
    assert rep[0]["LF"]["failed_first_check"] == [] and rep[1]["LF"]["walked_serially"] and not rep[1]["RF"]["walked_serially"]
    assert rep[1]["LF"]["failed_first_check"] == [0, 8, 16, 24, 32]
    assert chunk_report(dict(angles=np.zeros((3, 1, 4, 7)), chunk_flags=None, chunk_stats=dict(chunks=0)), ["RF"], 4) == [{}, {}, {}]


def test_output_side_converters_and_resampling():
    """utils.dict_to_nparray_angle / interpolate_signal / interpolate_joint_angles (reference seqikpy/utils.py:313-362) and
    the constant tables the reference's callers import (data.NMF_SIZE, SKELETON, get_pts2align)."""
    from scipy.interpolate import PchipInterpolator
    from seqikpy_amd import data, utils
    rng = np.random.default_rng(0)
    dofs = ["ThC_roll", "ThC_yaw", "ThC_pitch", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
    ang = {"RF_leg": {d: rng.normal(size=50) for d in dofs}}
    full = utils.dict_to_nparray_angle(ang, "RF", True)
    assert full.shape == (50, 7) and all(np.array_equal(full[:, i], ang["RF_leg"][d]) for i, d in enumerate(dofs))
    assert np.array_equal(utils.dict_to_nparray_angle(ang, "RF", False), full[:, :6])
    series = {"Angle_RF_ThC_yaw": np.sin(np.linspace(0, 3, 100)), "Angle_RF_ThC_pitch": rng.normal(size=100)}
    out = utils.interpolate_joint_angles(series, original_ts=1e-2, new_ts=1e-3)
    assert list(out) == list(series)
    x_old, x_new = np.arange(0, 1.0, 1e-2), np.arange(0, 1.0, 1e-3)
    for k in series:
        assert out[k].shape == x_new.shape
        assert np.array_equal(out[k], PchipInterpolator(x_old, series[k])(x_new))
        assert np.array_equal(out[k][::10][:99], series[k][:99])      # passes through the samples
    bad = np.ones(10)
    bad[3] = np.inf
    fixed = utils.interpolate_signal(bad, 1.0, 0.5)      # the reference's repair: inf and the last sample zeroed in place
    assert np.isfinite(fixed).all() and bad[3] == 0 and bad[-1] == 0
    assert len(data.NMF_SIZE) == 32 and data.NMF_SIZE["RF"] == 2.26 and data.NMF_SIZE["RM_Coxa"] == 0.182
    for leg in data.LEGS:
        assert sum(data.NMF_SIZE[f"{leg}_{s}"] for s in data.SEGMENTS) == pytest.approx(data.NMF_SIZE[leg], abs=1e-12)
    assert data.SKELETON[:2] == ["base_anten_R", "tip_anten_R"] and len(data.SKELETON) == 17
    assert "RF_leg" not in data.get_pts2align("rec_RF_x") and "LF_leg" in data.get_pts2align("rec_RF_x")
    assert set(data.get_pts2align("rec_RLF")) == {"R_head", "Thorax", "L_head"} and "RF_leg" in data.PTS2ALIGN


def test_bench_depth_candidates_cover_the_measured_region():
    """bench.py calibrates how many steps it keeps in flight over exactly the region it measures: the candidate
    list is a function of the step count -- the fixed depths, the balanced depth (fewest rounds of at most 20, equal size) and
    depths 16 / 20 with a short last round on the latency kernel; never more than 20 (more streams than hardware queues collapse)."""
    import os
    import sys
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench_support as bench
    assert bench.MAX_DEPTH == 20
    c20 = bench.depth_candidates(20)
    assert c20[:5] == [(3, 0, None), (8, 1, None), (12, 1, None), (16, 1, None), (20, 1, None)]
    # 20 steps: all at once (depth 20), or 16 + a last, partial round of 4 on the library's kernel choice
    assert (16, 1, (16, 20)) in c20 and len(c20) == 6
    assert (20, 1, None) in bench.depth_candidates(100) and (17, 1, None) in bench.depth_candidates(50)       # 5 x 20; 3 x 17
    assert (16, 1, (96, 100)) in bench.depth_candidates(100) and (20, 1, (40, 45)) in bench.depth_candidates(45)
    for k in (1, 4, 6, 16, 17, 20, 32, 33, 45, 100, 1000):
        for streams, pipe, lat in bench.depth_candidates(k):
            assert 1 <= streams <= bench.MAX_DEPTH and pipe in (0, 1)
            assert lat is None or (streams in (16, 20) and streams <= lat[0] < lat[1] == k and lat[1] - lat[0] <= 8)
    assert all(t is None for _, _, t in bench.depth_candidates(16)) and all(t is None for _, _, t in bench.depth_candidates(4))
    # the whole problem brings 1 465 wavefronts per step: depths beyond 2 x 3 072 slots / that are left out; a 1/8 share keeps all
    assert bench.depth_candidates(20, 93750) == [(3, 0, None)] and len(bench.depth_candidates(20, 11718)) == 6
    assert [c[0] for c in bench.depth_candidates(20, 46872)] == [3, 8]
    # bench.py asks for the environment of 20 steps in flight (22 hardware queues: 24 held queues cost the other kernels of a
    # process 10 %) in its __main__ block, before anything imports the HIP runtime, and records what it ran with
    import seqikpy_amd
    assert seqikpy_amd.recommended_env(steps_in_flight=bench.MAX_DEPTH)["GPU_MAX_HW_QUEUES"] == "22"
    with open(os.path.join(ROOT, "bench.py")) as fh:
        text = fh.read()
    assert "recommended_env(steps_in_flight=20)" in text and '"env": runtime_env()' in text


def test_bench_lifeline_prints_the_line_so_far_and_leaves():
    """bench_support.Lifeline: when an armed deadline passes, rank 0 writes the line the caller registered to the JSON descriptor
    and the process leaves with the exit code the deadline was armed with -- 75 while only the provisional headline exists, 0 once
    the verified headline is what gets printed (round-5 advice: a stuck run must not look healthy); other ranks write nothing; a
    disarmed lifeline does nothing."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench_support as b\n"
            "rank, code = int(sys.argv[1]), int(sys.argv[2])\n"
            "l = b.Lifeline(rank, 1)\n"
            "l.arm(30.0, lambda: 'never', 'first stage', 75); l.disarm(); time.sleep(0.6)\n"
            "l.arm(0.2, lambda: '{\"value\": 42}', 'second stage', code)\n"
            "time.sleep(8); print('NOT REACHED')\n" % ROOT)
    for want in (75, 0):
        r0 = subprocess.run([sys.executable, "-c", code, "0", str(want)], capture_output=True, text=True, timeout=300)
        assert r0.returncode == want and r0.stdout.strip() == '{"value": 42}', r0.stdout + r0.stderr
        assert "second stage did not finish in time" in r0.stderr and "NOT REACHED" not in r0.stdout
    assert b_exit_provisional() == 75
    r1 = subprocess.run([sys.executable, "-c", code, "1", "75"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 75 and r1.stdout.strip() == "" and "leaving" in r1.stderr


def b_exit_provisional():
    import sys
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench_support
    return bench_support.Lifeline.EXIT_PROVISIONAL
