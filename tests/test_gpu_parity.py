"""GPU tier (`-m gpu`): parity of the HIP path, called through the C ABI, against the oracle
(bit for bit), the committed golden vectors (1e-4 rad), and size-independent properties at
BASELINE.json's full sizes."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from conftest import DOFS, GOLDEN, PKG_PARENT, ROOT, good_frames, leg_arrays, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4  # rad -- BASELINE.json north star


def _params(lib, z, legs):
    return [lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]


def _stack(z, legs, sl=slice(None)):
    return np.stack([z[f"{l}_pose"][sl] for l in legs])[None]


@pytest.fixture(scope="module")
def lib(hiplib):
    if hiplib.load().seqik_device_count() < 1:
        pytest.fail("GPU tier needs a GPU: the HIP path must not be skipped silently")
    return hiplib


@pytest.mark.parametrize("name", ["anipose_scipy_cut", "df3d_100", "df3d_1000"])
def test_hip_equals_oracle_bit_for_bit(lib, oracle, name):
    z = load_golden(name)
    legs = [str(l) for l in z["legs"]]
    out = lib.solve_seq(_stack(z, legs), _params(lib, z, legs), want_fk=True, want_diag=True)
    for i, leg in enumerate(legs):
        ref = oracle.seq_leg(*leg_arrays(z, leg))
        assert np.array_equal(out["angles"][0, i], ref["angles"]), leg
        assert np.array_equal(out["fk"][0, i], ref["fk"]), leg
        assert np.array_equal(out["status"][0, i], ref["status"]), leg
        assert np.array_equal(out["nfev"][0, i], ref["nfev"]), leg


def test_hip_vs_shipped_golden_full_recording(lib, oracle):
    """Config 4 stand-in: the shipped 6000-frame anipose recording, RF + LF in one launch."""
    z = load_golden("anipose_shipped")
    legs = ["RF", "LF"]
    out = lib.solve_seq(_stack(z, legs), _params(lib, z, legs), want_fk=True)
    cut = z["fk_frames"]
    for i, leg in enumerate(legs):
        ok = good_frames(leg, 6000)
        err = np.abs(out["angles"][0, i] - z[f"{leg}_angles"])
        assert err[ok].max() < TOL, (leg, err[ok].max())
        assert np.all(~ok[np.where(err.max(1) >= TOL)[0]])
        assert np.abs(out["fk"][0, i][cut] - z[f"{leg}_fk_cut"])[ok[cut]].max() < TOL
        ref = oracle.seq_leg(*leg_arrays(z, leg))
        assert np.array_equal(out["angles"][0, i], ref["angles"])


def test_hip_vs_reference_source_run_df3d(lib):
    """Config 2: all 6 legs of the full locomotion recording vs the reference source run here."""
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    out = lib.solve_seq(_stack(z, legs), _params(lib, z, legs), want_fk=True)
    for i, leg in enumerate(legs):
        assert np.abs(out["angles"][0, i] - z[f"{leg}_angles"]).max() < TOL, leg
        assert np.abs(out["fk"][0, i] - z[f"{leg}_fk"]).max() < TOL, leg


def test_config1_single_leg_100_frames(lib):
    z = load_golden("anipose_shipped")
    out = lib.solve_seq(_stack(z, ["RF"], slice(0, 100)), _params(lib, z, ["RF"]))
    assert np.abs(out["angles"][0, 0] - z["RF_angles"][:100]).max() < TOL


def test_batched_sequences_equal_individual_runs(lib):
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    offs = [0, 130, 500, 777, 900]
    pose = np.concatenate([_stack(z, legs, slice(o, o + 50)) for o in offs])
    for block in (64, 256):
        batch = lib.solve_seq(pose, params, want_fk=True, block_size=block)
        for s in range(len(offs)):
            one = lib.solve_seq(pose[s:s + 1], params, want_fk=True)
            assert np.array_equal(batch["angles"][s], one["angles"][0])
            assert np.array_equal(batch["fk"][s], one["fk"][0])


@pytest.mark.parametrize("split", [1, 2, 3])
def test_stage_subsets_compose(lib, split):
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = _stack(z, legs)
    full = lib.solve_seq(pose, params, want_fk=True)
    a = lib.solve_seq(pose, params, 1, split, want_fk=True)
    assert a["fk"] is None
    b = lib.solve_seq(pose, params, split + 1, 4, angles=a["angles"], want_fk=True)
    assert np.array_equal(b["angles"], full["angles"])
    assert np.array_equal(b["fk"], full["fk"])


def test_empty_and_tiny_inputs(lib):
    z = load_golden("df3d_100")
    params = _params(lib, z, ["RF"])
    pose = _stack(z, ["RF"])
    assert lib.solve_seq(pose[:, :, :0], params)["angles"].shape == (1, 1, 0, 7)
    assert lib.solve_seq(pose[:0], params)["angles"].shape == (0, 1, 100, 7)
    one = lib.solve_seq(pose[:, :, :1], params)
    assert np.array_equal(one["angles"][0, 0, 0], lib.solve_seq(pose, params)["angles"][0, 0, 0])


def test_bad_seed_raises_value_error(lib):
    z = load_golden("df3d_100")
    seeds = z["RF_seeds"].copy()
    seeds[2] = 3.0  # stage-1 pitch seed outside +-90 deg
    with pytest.raises(ValueError, match="outside of provided bounds"):
        lib.solve_seq(_stack(z, ["RF"]), [lib.leg_params_from_arrays(z["RF_seg"], z["RF_bounds"], seeds)])


def test_python_api_matches_reference_layout(lib, tmp_path):
    """LegInvKinSeq.run_ik_and_fk: dict keys, shapes, dtypes, export files, leg filtering."""
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    z = load_golden("anipose_shipped")
    aligned = {"R_head": np.zeros((200, 2, 3)), "RF_leg": z["RF_pose"][:200], "LF_leg": z["LF_pose"][:200],
               "RM_leg": z["RF_pose"][:200], "Neck": np.zeros((1, 1, 3))}
    ik = LegInvKinSeq(aligned, KinematicChainSeq(BOUNDS, ["RF", "LF"]), INITIAL_ANGLES, log_level="ERROR")
    ang, fk = ik.run_ik_and_fk(export_path=tmp_path, hide_progress_bar=True)
    assert list(ang.keys()) == [f"Angle_{leg}_{d}" for leg in ("RF", "LF") for d in DOFS]
    assert list(fk.keys()) == ["RF_leg", "LF_leg"]  # RM is not in the chain's body_size: skipped
    for leg in ("RF", "LF"):
        assert fk[f"{leg}_leg"].shape == (200, 9, 3) and fk[f"{leg}_leg"].dtype == np.float64
        got = np.stack([ang[f"Angle_{leg}_{d}"] for d in DOFS], 1)
        assert got.shape == (200, 7)
        assert np.abs(got - z[f"{leg}_angles"][:200]).max() < TOL
        assert np.array_equal(fk[f"{leg}_leg"][:, 0], aligned[f"{leg}_leg"][:, 0])
    with open(tmp_path / "leg_joint_angles.pkl", "rb") as f:
        saved = pickle.load(f)
    assert np.array_equal(saved["Angle_LF_TiTa_pitch"], ang["Angle_LF_TiTa_pitch"])
    assert os.path.exists(tmp_path / "forward_kinematics.pkl")
    assert ang is ik.joint_angles_dict


def test_python_stagewise_api_and_per_frame_seam(lib):
    """calculate_ik_stage stage by stage == run_ik_and_fk; calculate_ik / calculate_fk on one frame."""
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    z = load_golden("anipose_shipped")
    pose = z["RF_pose"][:60]
    kc = KinematicChainSeq(BOUNDS, ["RF"])
    ik = LegInvKinSeq({"RF_leg": pose}, kc, INITIAL_ANGLES, log_level="ERROR")
    full_ang, full_fk = ik.run_ik_and_fk(frame_parallel=False)
    full_ang = {k: v.copy() for k, v in full_ang.items()}
    ik2 = LegInvKinSeq({"RF_leg": pose}, kc, INITIAL_ANGLES, log_level="ERROR")
    for stage in (1, 2, 3, 4):
        fk = ik2.calculate_ik_stage(pose[:, stage], pose[:, 0], INITIAL_ANGLES["RF"][f"stage_{stage}"], "RF",
                                    stage=stage)
    for k in full_ang:
        assert np.array_equal(ik2.joint_angles_dict[k], full_ang[k]), k
    assert np.array_equal(fk, full_fk["RF_leg"])
    # two-call form with accumulated dict
    ik3 = LegInvKinSeq({"RF_leg": pose}, kc, INITIAL_ANGLES, log_level="ERROR")
    ik3.run_ik_and_fk(stages=[1, 2])
    assert set(ik3.joint_angles_dict) == {f"Angle_RF_{d}" for d in DOFS[:4]}
    ang3, fk3 = ik3.run_ik_and_fk(stages=[3, 4])
    assert all(np.array_equal(ang3[k], full_ang[k]) for k in full_ang)
    assert np.array_equal(fk3["RF_leg"], full_fk["RF_leg"])
    # per-frame seam: frame 0, stage 3
    chain = kc.create_leg_chain("RF", stage=3, angles=full_ang, t=0)
    x = ik.calculate_ik(chain, pose[0, 3] - pose[0, 0], INITIAL_ANGLES["RF"]["stage_3"])
    assert x.shape == (8,)
    assert x[5] == full_ang["Angle_RF_CTr_roll"][0] and x[6] == full_ang["Angle_RF_FTi_pitch"][0]
    q = np.concatenate([[0.0], [full_ang[f"Angle_RF_{d}"][7] for d in DOFS], [0.0]])
    chain4 = kc.create_leg_chain("RF", stage=4, angles=full_ang, t=7)
    assert np.abs(ik.calculate_fk(chain4, q) + pose[7, 0] - full_fk["RF_leg"][7]).max() < 1e-12


def test_device_pointer_entry_point_with_torch(lib):
    import torch
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = np.concatenate([_stack(z, legs)] * 3)
    host = lib.solve_seq(pose, params, want_fk=True, want_diag=True)
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((3, 6, 100, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((3, 6, 100, 9, 3), dtype=torch.float64, device="cuda")
    d_nfev = torch.zeros((3, 6, 100, 4), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lib.solve_seq_device(d_pose.data_ptr(), 3, 6, 100, params, d_ang.data_ptr(), d_fk.data_ptr(),
                             d_nfev=d_nfev.data_ptr(), stream=side.cuda_stream)
    side.synchronize()
    assert np.array_equal(d_ang.cpu().numpy(), host["angles"])
    assert np.array_equal(d_fk.cpu().numpy(), host["fk"])
    assert np.array_equal(d_nfev.cpu().numpy(), host["nfev"])


def test_planar_device_layout_equals_dense(lib):
    """SeqikLayout: pose [chain][5][frame][3] + angles [chain][7][frame] give the same bits as the dense
    layout, including a later-stage run that reads earlier angles through the strides."""
    import torch
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = np.concatenate([_stack(z, legs, slice(0, 64)), _stack(z, legs, slice(30, 94))])
    host = lib.solve_seq(pose, params, want_fk=True)
    S, L, T = pose.shape[:3]
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_ang = torch.zeros((S, L, 7, T), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, L, T, 9, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for first, last in ((1, 2), (3, 4)):
        lib.solve_seq_device(d_pose.data_ptr(), S, L, T, params, d_ang.data_ptr(), d_fk.data_ptr(),
                             first_stage=first, last_stage=last, stream=stream, layout=lib.planar_layout(T))
    torch.cuda.synchronize()
    assert np.array_equal(d_ang.cpu().numpy().transpose(0, 1, 3, 2), host["angles"])
    assert np.array_equal(d_fk.cpu().numpy(), host["fk"])


@pytest.mark.parametrize("variant", ["iid", "smooth"])
def test_full_size_synthetic_properties(lib, oracle, variant):
    """Config 3 at scale (default 131072 sequences x 6 legs x 8 frames = 6.3 M leg-frames would take
    seconds; use 16384 x 6 x 64 = 6.3 M): size-independent properties instead of an oracle run.
      * the returned FK is the forward kinematics of the returned angles (independent numpy FK);
      * every angle lies inside its bounds;
      * the claw residual is small for these reachable, lightly perturbed targets;
      * a random sample of chains equals the oracle bit for bit."""
    from seqikpy_amd import data, synthetic, utils
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    S, N = 16384, 64
    pose = synthetic.synthetic_pose(S, N, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                    variant=variant)
    params = [lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    out = lib.solve_seq(pose, params, want_fk=True)
    ang, fk = out["angles"], out["fk"]
    assert np.isfinite(ang).all() and np.isfinite(fk).all()
    for li, leg in enumerate(legs):
        lb = np.array([data.BOUNDS_LOCOMOTION[f"{leg}_{d}"][0] for d in DOFS])
        ub = np.array([data.BOUNDS_LOCOMOTION[f"{leg}_{d}"][1] for d in DOFS])
        assert (ang[:, li] >= lb).all() and (ang[:, li] <= ub).all()
        seg = [body[f"{leg}_{s}"] for s in data.SEGMENTS]
        kp = synthetic.leg_forward_kinematics(ang[:, li], seg) + pose[:, li, :, :1]
        assert np.abs(kp - fk[:, li][:, :, [0, 4, 6, 7, 8]]).max() < 1e-12
        assert np.array_equal(fk[:, li, :, 0], pose[:, li, :, 0])
    # sequential fitting of noisy reachable targets: every stage should land near its key point
    resid = np.linalg.norm(fk[:, :, :, [4, 6, 7, 8]] - pose[:, :, :, 1:], axis=-1)
    assert np.median(resid) < 0.05
    rng = np.random.default_rng(7)
    for s, li in zip(rng.integers(0, S, 24), rng.integers(0, 6, 24)):
        leg = legs[li]
        seg, b, seeds = oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
        ref = oracle.seq_leg(pose[s, li], seg, b, seeds)
        assert np.array_equal(ang[s, li], ref["angles"]) and np.array_equal(fk[s, li], ref["fk"])


def test_random_legs_and_nasty_targets_bit_for_bit(lib, oracle):
    """Made-up legs (segment lengths, joint limits, seeds on bounds) with unreachable, degenerate and repeated
    targets, 8 different legs per launch: HIP == oracle bit for bit, angles inside their limits, all finite."""
    from conftest import random_leg_case
    rng = np.random.default_rng(977)
    for _ in range(12):
        cases = [random_leg_case(rng, 40) for _ in range(8)]
        pose = np.stack([c[0] for c in cases])[None]
        params = [lib.leg_params_from_arrays(c[1], c[2], c[3]) for c in cases]
        out = lib.solve_seq(pose, params, want_fk=True, want_diag=True)
        assert np.isfinite(out["angles"]).all() and np.isfinite(out["fk"]).all()
        for i, (p, seg, b, seeds) in enumerate(cases):
            ref = oracle.seq_leg(p, seg, b, seeds)
            assert np.array_equal(out["angles"][0, i], ref["angles"])
            assert np.array_equal(out["fk"][0, i], ref["fk"])
            assert np.array_equal(out["status"][0, i], ref["status"])
            assert np.array_equal(out["nfev"][0, i], ref["nfev"])
            assert (out["angles"][0, i] >= b[:, 0]).all() and (out["angles"][0, i] <= b[:, 1]).all()


def test_open_and_one_sided_joint_limits_bit_for_bit(lib, oracle):
    """Joints without limits (IKPy's default bounds are (-inf, inf)) or with a limit on one side only, 8 different legs
    per launch, lane per chain and stage pipeline: HIP == oracle bit for bit.  The kernel folds isfinite(bound) into
    per-leg constants (StageConst::gate_lb / gate_ub: NaN comparands); here those constants are NaN."""
    from conftest import random_leg_case
    rng = np.random.default_rng(1234)
    for rep in range(6):
        cases = []
        for _ in range(8):
            p, seg, b, seeds = random_leg_case(rng, 32)
            b = b.copy()
            for j in range(7):
                kind = rng.integers(0, 4)
                if kind == 1:
                    b[j, 0] = -np.inf
                elif kind == 2:
                    b[j, 1] = np.inf
                elif kind == 3:
                    b[j] = (-np.inf, np.inf)
            cases.append((p, seg, b, seeds))
        pose = np.stack([c[0] for c in cases])[None]
        params = [lib.leg_params_from_arrays(c[1], c[2], c[3]) for c in cases]
        out = lib.solve_seq(pose, params, want_fk=True, want_diag=(rep % 2 == 0))
        assert np.isfinite(out["angles"]).all() and np.isfinite(out["fk"]).all()
        for i, (p, seg, b, seeds) in enumerate(cases):
            ref = oracle.seq_leg(p, seg, b, seeds)
            assert np.array_equal(out["angles"][0, i], ref["angles"])
            assert np.array_equal(out["fk"][0, i], ref["fk"])
            if rep % 2 == 0:
                assert np.array_equal(out["status"][0, i], ref["status"])
                assert np.array_equal(out["nfev"][0, i], ref["nfev"])


@pytest.mark.parametrize("lanes", [1, 3, 5, 8, 9, 64])
def test_lanes_per_wave_does_not_change_results(lib, lanes):
    """The lane -> chain mapping (SeqikOptions.reserved[0]) only decides where a chain runs: 23 sequences x 6
    legs (ragged against 5 and 64 lanes per wave) give the same bits as the automatic choice."""
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    base = np.stack([z[f"{l}_pose"] for l in legs])
    pose = np.stack([base[:, o:o + 30] for o in [(41 * i) % 960 for i in range(23)]])
    auto = lib.solve_seq(pose, params, want_fk=True, want_diag=True)
    got = lib.solve_seq(pose, params, want_fk=True, want_diag=True, lanes_per_wave=lanes)
    for k in ("angles", "fk", "status", "nfev"):
        assert np.array_equal(got[k], auto[k]), k
    zg = load_golden("generic_rf_100")
    gp = [lib.leg_params_from_arrays(zg["RF_seg"], zg["RF_bounds"], zg["RF_seeds"])]
    gpose = np.stack([zg["RF_pose"][o:o + 6] for o in range(0, 70, 10)])[:, None]  # 7 sequences x 1 leg x 6 frames
    gen_auto = lib.solve_generic(gpose, gp)
    gen = lib.solve_generic(gpose, gp, lanes_per_wave=lanes)
    assert np.array_equal(gen["angles"], gen_auto["angles"]) and np.array_equal(gen["fk"], gen_auto["fk"])
    # up to 8 chains per wavefront the generic kernel splits a pass over groups of 8 lanes: same bits without
    one_lane = lib.solve_generic(gpose, gp, lanes_per_wave=lanes, lane_groups=False, want_diag=True)
    grouped = lib.solve_generic(gpose, gp, lanes_per_wave=lanes, want_diag=True)
    for k in ("angles", "fk", "status", "nfev"):
        assert np.array_equal(one_lane[k], grouped[k]), k
    wide = lib.solve_generic(gpose, gp, lanes_per_wave=lanes, block_size=256)  # four wavefronts per workgroup
    assert np.array_equal(wide["angles"], gen_auto["angles"]) and np.array_equal(wide["fk"], gen_auto["fk"])


def test_single_launch_equals_one_launch_per_stage(lib):
    """Default = every wave takes its chains through stages 1-4 in one launch; staged=1 = one launch per stage
    (what a run with diagnostics or a stage subset uses): same bits, with and without FK, any wave shape."""
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    base = np.stack([z[f"{l}_pose"] for l in legs])
    pose = np.stack([base[:, o:o + 40] for o in [(41 * i) % 950 for i in range(200)]])
    # staged=1 rules out the automatic stage pipeline (1200 chains would otherwise get it): the four stage kernels run
    staged = lib.solve_seq(pose, params, want_fk=True, staged=1)
    for kw in (dict(), dict(pipeline=1), dict(pipeline=2), dict(pipeline=1, lanes_per_wave=64), dict(lanes_per_wave=64),
               dict(lanes_per_wave=3), dict(pipeline=1, lanes_per_wave=3), dict(block_size=256), dict(pipeline=1, block_size=256),
               dict(interleave_legs=1), dict(pipeline=1, interleave_legs=1), dict(interleave_legs=1, lanes_per_wave=7)):
        one = lib.solve_seq(pose, params, want_fk=True, **kw)
        assert np.array_equal(one["angles"], staged["angles"]) and np.array_equal(one["fk"], staged["fk"]), kw
    assert np.array_equal(lib.solve_seq(pose, params, want_fk=False)["angles"], staged["angles"])
    diag = lib.solve_seq(pose, params, want_fk=True, want_diag=True)  # diagnostics always run staged
    assert np.array_equal(diag["angles"], staged["angles"])


def test_many_recordings_of_different_length_in_one_call(lib):
    """batch.run_ik_and_fk_many: recordings of 100, 57, 100 and 8 frames (bucketed; and padded to a common
    length) == one LegInvKinSeq.run_ik_and_fk per recording, bit for bit."""
    from seqikpy_amd import data
    from seqikpy_amd.batch import run_ik_and_fk_many
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.utils import calculate_body_size
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    kc = KinematicChainSeq(data.BOUNDS_LOCOMOTION, legs, calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs))
    cuts = [(0, 100), (300, 357), (500, 600), (900, 908)]
    recs = [{f"{l}_leg": z[f"{l}_pose"][a:b] for l in legs} | {"Neck": np.zeros((1, 1, 3))} for a, b in cuts]
    single = [LegInvKinSeq(r, kc, data.INITIAL_ANGLES_LOCOMOTION, log_level="ERROR").run_ik_and_fk(frame_parallel=False) for r in recs]
    for pad in (0, 64):
        many = run_ik_and_fk_many(recs, kc, data.INITIAL_ANGLES_LOCOMOTION, pad_to_multiple=pad, frame_parallel=False)
        assert len(many) == len(recs)
        for (ang, fk), (ang1, fk1) in zip(many, single):
            assert list(ang.keys()) == list(ang1.keys()) and list(fk.keys()) == list(fk1.keys())
            assert all(np.array_equal(ang[k], ang1[k]) for k in ang)
            assert all(np.array_equal(fk[k], fk1[k]) for k in fk)
    assert run_ik_and_fk_many([], kc) == []


def test_concurrent_calls_from_several_host_threads(lib):
    """The ABI is thread-safe for distinct buffers (per-thread constant-table cache and error string): four
    threads solving different legs / recordings at once get the same bits as one after the other; repeated
    stream open / close does not leak device state."""
    from concurrent.futures import ThreadPoolExecutor
    from seqikpy_amd.streaming import solve_streamed
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    jobs = []
    for i in range(8):
        sub = [legs[i % 6], legs[(i + 2) % 6]]
        pose = np.stack([z[f"{l}_pose"][40 * i:40 * i + 120] for l in sub])[None]
        jobs.append((pose, _params(lib, z, sub)))
    serial = [lib.solve_seq(p, prm, want_fk=True) for p, prm in jobs]
    with ThreadPoolExecutor(4) as ex:
        par = list(ex.map(lambda j: lib.solve_seq(j[0], j[1], want_fk=True), jobs))
    for a, b in zip(serial, par):
        assert np.array_equal(a["angles"], b["angles"]) and np.array_equal(a["fk"], b["fk"])
    with ThreadPoolExecutor(3) as ex:
        st = list(ex.map(lambda j: solve_streamed(np.concatenate([j[0]] * 5), j[1], slab_seq=2), jobs[:6]))
    for a, b in zip(serial, st):
        assert all(np.array_equal(b["angles"][k], a["angles"][0]) for k in range(5))
    for _ in range(20):
        solve_streamed(jobs[0][0], jobs[0][1], slab_seq=1, n_slots=2)


@pytest.mark.gpu
@pytest.mark.parametrize("want_fk", [False, True])
def test_launches_in_flight_on_several_streams_equal_serial(lib, want_fk):
    """The device entry point on three HIP streams back to back, no host synchronisation in between (the
    benchmark's launch pattern), different key points per launch: every launch returns the bits of the same call
    made alone.  (Each stream has its own stage hand-off workspace inside the library.)"""
    import torch
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    T, n_launch = 40, 18
    poses = [np.stack([z[f"{l}_pose"][(53 * i) % 950:(53 * i) % 950 + T] for l in legs])[None] for i in range(n_launch)]
    alone = [lib.solve_seq(p, params, want_fk=want_fk) for p in poses]
    streams = [torch.cuda.Stream() for _ in range(3)]
    d_pose = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in poses]
    d_ang = [torch.zeros((1, 6, T, 7), dtype=torch.float64, device="cuda") for _ in range(n_launch)]
    d_fk = [torch.zeros((1, 6, T, 9, 3), dtype=torch.float64, device="cuda") for _ in range(n_launch)]
    torch.cuda.synchronize()
    for i in range(n_launch):
        lib.solve_seq_device(d_pose[i].data_ptr(), 1, 6, T, params, d_ang[i].data_ptr(),
                             d_fk[i].data_ptr() if want_fk else 0, stream=streams[i % 3].cuda_stream)
    torch.cuda.synchronize()
    bad = [i for i in range(n_launch) if not np.array_equal(d_ang[i].cpu().numpy(), alone[i]["angles"])]
    assert not bad, bad
    if want_fk:
        assert all(np.array_equal(d_fk[i].cpu().numpy(), alone[i]["fk"]) for i in range(n_launch))
    lib.release_workspaces()
    again = lib.solve_seq(poses[0], params, want_fk=want_fk)  # workspaces come back on demand
    assert np.array_equal(again["angles"], alone[0]["angles"])


@pytest.mark.gpu
def test_eight_legs_and_single_frame_sequences(lib, oracle):
    """The ABI's leg-count limit (8 legs per call: six real ones + two repeated with other limits / seeds) and the
    shortest possible sequences (one frame each, 700 of them: every solve starts from the seeds); nine legs are
    refused before any launch."""
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    names = legs + ["RF", "LH"]
    rng = np.random.default_rng(11)
    params, arrays = [], []
    for i, l in enumerate(names):
        seg, b, seeds = z[f"{l}_seg"].copy(), z[f"{l}_bounds"].copy(), z[f"{l}_seeds"].copy()
        if i >= 6:   # a different morphology / different limits for the repeated legs
            seg *= 1.0 + 0.1 * rng.random(4)
            b[:, 0] -= 0.05
            b[:, 1] += 0.05
        params.append(lib.leg_params_from_arrays(seg, b, seeds))
        arrays.append((seg, b, seeds))
    pose = np.stack([z[f"{l}_pose"][:700] for l in names])            # (8, 700, 5, 3)
    seqs = np.ascontiguousarray(pose.transpose(1, 0, 2, 3)[:, :, None])   # (700, 8, 1, 5, 3): one frame per sequence
    got = lib.solve_seq(seqs, params, want_fk=True)
    for li in (0, 3, 6, 7):
        seg, b, seeds = arrays[li]
        for s in (0, 1, 350, 699):
            ref = oracle.seq_leg(seqs[s, li], seg, b, seeds)
            assert np.array_equal(got["angles"][s, li], ref["angles"]) and np.array_equal(got["fk"][s, li], ref["fk"])
    whole = lib.solve_seq(pose[None, :, :40], params, want_fk=False)      # the same 8 legs as ordinary recordings
    ref = oracle.seq_leg(pose[7, :40], *arrays[7])
    assert np.array_equal(whole["angles"][0, 7], ref["angles"])
    with pytest.raises(ValueError, match="n_legs"):
        lib.solve_seq(np.concatenate([pose, pose[:1]])[None, :, :4], params + params[:1])


@pytest.mark.gpu
def test_more_streams_than_remembered_workspaces(lib):
    """The library remembers a hand-off workspace for 64 streams; launches on 70 streams (two rounds, so that evicted
    streams come back) still return the bits of a launch made alone.  (pipeline=1: the lane-per-chain kernels are the ones
    that use the workspace.)"""
    import torch
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = np.stack([z[f"{l}_pose"][:30] for l in legs])[None]
    alone = lib.solve_seq(pose, params, want_fk=False)["angles"]
    d_pose = torch.from_numpy(np.ascontiguousarray(pose)).cuda()
    streams = [torch.cuda.Stream() for _ in range(70)]
    outs = [torch.zeros((1, 6, 30, 7), dtype=torch.float64, device="cuda") for _ in range(140)]
    torch.cuda.synchronize()
    for i in range(140):
        lib.solve_seq_device(d_pose.data_ptr(), 1, 6, 30, params, outs[i].data_ptr(), 0,
                             stream=streams[i % 70].cuda_stream, pipeline=1 if i % 2 else 0)
    torch.cuda.synchronize()
    assert all(np.array_equal(o.cpu().numpy(), alone) for o in outs)
    lib.release_workspaces()


def test_host_calls_do_not_strand_device_memory(lib):
    """host-buffer calls reuse one pooled context (stream + arena + hand-off workspace) instead of
    leaving a workspace behind per call; seqik_release_workspaces() gives everything back."""
    import threading
    import torch
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = [lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    base = np.stack([z[f"{l}_pose"][:64] for l in legs])
    pose = np.ascontiguousarray(np.broadcast_to(base, (512,) + base.shape))     # 196 608 leg-frames per call
    per_call = pose.shape[0] * 6 * 64 * (120 + 56 + 216 + 96)                   # arena + hand-off workspace, bytes
    ref = lib.solve_seq(pose[:2], params)
    torch.cuda.synchronize()
    lib.release_workspaces()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(40):
        out = lib.solve_seq(pose[: 512 - (i % 3)], params)
    assert np.array_equal(out["angles"][:2], ref["angles"])
    # the head / antenna entry point borrows the same pooled context (no hipMalloc / hipFree per call)
    zh = load_golden("anipose_head")
    head0 = lib.head_angles(zh["R_head"], zh["L_head"], zh["Neck"][:, 0], 0.1, 0.2)
    for i in range(20):
        head = lib.head_angles(zh["R_head"], zh["L_head"], zh["Neck"][:, 0], 0.1, 0.2)
    assert np.array_equal(head, head0)
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < 1.6 * per_call, (used, per_call)      # one context, not one workspace per call
    # concurrent callers get their own contexts and the same bits
    res = [None] * 4
    def work(k):
        res[k] = lib.solve_seq(pose[:128], params)["angles"]
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(np.array_equal(r[:2], ref["angles"]) for r in res)
    lib.release_workspaces()
    assert abs(free0 - torch.cuda.mem_get_info()[0]) <= 8 << 20
    # still works afterwards (everything is re-created on demand), and the caller's device is untouched
    dev = torch.cuda.current_device()
    assert np.array_equal(lib.solve_seq(pose[:2], params, device=0)["angles"], ref["angles"])
    assert torch.cuda.current_device() == dev
    lib.release_workspaces()


def test_device_division_and_square_root_equal_ieee_on_the_contract_range(lib):
    """DESIGN.md §2, floating-point contract: div_() / sqrt_() (the gfx950 expansions of `/` and `sqrt` without the
    range-scaling steps) == IEEE division / square root, bit for bit, for operands within [2^-767, 2^767] -- 4 M random
    pairs spread over the whole range, near-equal operands, powers of two, quotients close to rounding boundaries -- and
    give the IEEE results for zero, infinite and NaN operands."""
    rng = np.random.default_rng(11)
    n = 1 << 20

    def spread(lo, hi, size):   # random doubles with exponents uniform in [lo, hi], random sign
        return np.ldexp(1.0 + rng.random(size), rng.integers(lo, hi, size)) * rng.choice([-1.0, 1.0], size)

    a = np.concatenate([spread(-380, 380, n), spread(-60, 60, n), spread(-10, 10, n), np.ldexp(1.0, rng.integers(-300, 300, n))])
    b = np.concatenate([spread(-380, 380, n), spread(-60, 60, n), spread(-10, 10, n) , spread(-300, 300, n)])
    b[2 * n:2 * n + n // 2] = a[2 * n:2 * n + n // 2] * (1.0 + rng.integers(-4, 5, n // 2) * 2.0 ** -52)   # quotients next to 1
    q, r = lib.selftest_div_sqrt(a, b)
    with np.errstate(all="ignore"):
        assert np.array_equal(q, a / b)
        pos = np.abs(a)
        _, r = lib.selftest_div_sqrt(pos, b)
        assert np.array_equal(r, np.sqrt(pos))
        wide = spread(-766, 766, n)                     # the whole stated range
        q, r = lib.selftest_div_sqrt(wide, np.ones(n))
        assert np.array_equal(q, wide) and np.array_equal(lib.selftest_div_sqrt(np.abs(wide), np.ones(n))[1], np.sqrt(np.abs(wide)))
        sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 3.0, 1e-300, 1e300])
        aa, bb = [x.ravel() for x in np.meshgrid(sp, sp)]
        q, r = lib.selftest_div_sqrt(aa, bb)
        want = aa / bb
        assert np.array_equal(np.isnan(q), np.isnan(want)) and np.array_equal(q[~np.isnan(want)], want[~np.isnan(want)])
        assert np.array_equal(np.signbit(q[~np.isnan(want)]), np.signbit(want[~np.isnan(want)]))
        ws = np.sqrt(aa)
        assert np.array_equal(np.isnan(r), np.isnan(ws)) and np.array_equal(r[~np.isnan(ws)], ws[~np.isnan(ws)])


def test_coleman_li_square_root_at_limits_of_exactly_zero(lib):
    """Advisor finding of round 3.  The shipped limits contain exact zeros (CTr_pitch ub, FTi_pitch lb, TiTa_pitch ub): a
    trial point that lands on such a limit is put at next_toward(0, .) = +-2^-1074 by make_strictly_feasible(rstep = 0), and
    the Coleman-Li distance v = |x - bound| of the next pass is the smallest subnormal -- outside the [2^-767, 2^767] range
    the range-scaling-free sqrt_ / sqrt_pos_ are stated for.  What the kernels rely on, checked on the device:
      * sqrt_pos_ and sqrt_ of 2^-1074 and of every subnormal / tiny EVEN power of two are exact (the results are powers of
        two; the hardware seed v_rsq_f64 is exact there and the Newton corrections vanish);
      * the values that occur with the shipped limits -- lb_in - lb, ub - ub_in for every joint of every leg of both
        bounds tables, i.e. 2^-1074 next to a zero limit and one ulp of the limit otherwise -- come out as IEEE sqrt;
      * normal operands down to 2^-767 come out as IEEE sqrt (the stated range, select-free variant).
    Subnormals that are not powers of two cannot be produced by the solver from limits that are 0 or of ordinary
    magnitude (the next iterate after 2^-1074 is x + d p_h with d = sqrt(v) = 2^-537); seqik_validate_legs rejects limits
    whose magnitude is positive but below 2^-600, which is the only way to get there."""
    from seqikpy_amd import data
    with np.errstate(all="ignore"):
        even = np.ldexp(1.0, np.arange(-1074, -700, 2))
        assert np.array_equal(lib.selftest_sqrt_pos(even), np.sqrt(even))
        assert np.array_equal(lib.selftest_div_sqrt(even, np.ones_like(even))[1], np.sqrt(even))
        vals = []
        for table in (data.BOUNDS, data.BOUNDS_LOCOMOTION):
            for lb, ub in table.values():
                vals += [np.nextafter(lb, ub) - lb, ub - np.nextafter(ub, lb)]
        vals = np.array(sorted(set(vals)))
        assert vals.min() == 2.0 ** -1074           # the case the finding is about is in the shipped tables
        assert np.array_equal(lib.selftest_sqrt_pos(vals), np.sqrt(vals))
        rng = np.random.default_rng(5)
        wide = np.ldexp(1.0 + rng.random(1 << 18), rng.integers(-767, 767, 1 << 18))
        assert np.array_equal(lib.selftest_sqrt_pos(wide), np.sqrt(wide))
        # where the iteration stops being exact (documentation of the boundary, not a requirement): subnormal operands
        # that are not powers of two
        sub = np.ldexp(1.0 + rng.random(4096), rng.integers(-1074, -1023, 4096))
        sub = sub[sub > 0]
        frac_exact = float(np.mean(lib.selftest_sqrt_pos(sub) == np.sqrt(sub)))
        print(f"sqrt_pos_ on random subnormals: {100 * frac_exact:.1f} % equal to IEEE")


def test_pipeline_watchdog_is_reported(lib):
    """The stage pipeline's watchdog (run_stage, PIPED: a lane that sits out more than PIPE_SPIN_LIMIT passes) cannot trip
    by construction; if it ever does the result must not pass silently (the reference raises, never returns garbage:
    seqikpy/leg_inverse_kinematics.py:62-69 -> IKPy raises on scipy status -1).  A DIAGNOSTIC build of the same sources
    with the limit set to one pass (csrc/libseqik_hip_watchdog.so, built by __graft_entry__.build()) trips it at once:
    the blocking entry point returns SEQIK_ERR_HIP with a message, the asynchronous one leaves the fault for
    seqik_check_faults() / the next call, every launch still terminates, and the product build reports nothing."""
    from seqikpy_amd import _lib
    if not os.path.exists(_lib.WATCHDOG_LIB_PATH):
        _lib.build_watchdog_variant()
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from seqikpy_amd import _lib
z = np.load(%r)
legs = [str(l) for l in z["legs"]]
pose = np.stack([z[f"{l}_pose"][:40] for l in legs])[None]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
try:
    _lib.solve_seq(pose, params, pipeline=2)
    print("HOST: no error")
except _lib.SeqikLibraryError as e:
    print("HOST:", e)
_lib.check_faults()                       # the blocking call has consumed the fault
print("CLEARED")
d_pose = torch.from_numpy(pose).cuda(); d_ang = torch.zeros((1, len(legs), 40, 7), dtype=torch.float64, device="cuda")
_lib.solve_seq_device(d_pose.data_ptr(), 1, len(legs), 40, params, d_ang.data_ptr(), pipeline=2)
torch.cuda.synchronize()
print("NAN" if bool(torch.isnan(d_ang).any()) else "NO NAN")
try:
    _lib.check_faults()
    print("DEVICE: no error")
except _lib.SeqikLibraryError as e:
    print("DEVICE:", e)
out = _lib.solve_seq(pose, params, pipeline=1)   # lane-per-chain kernels have no pipeline: clean
print("SERIAL OK" if np.isfinite(out["angles"]).all() else "SERIAL BAD")
''' % (PKG_PARENT, ROOT, os.path.join(GOLDEN, "df3d_100.npz"))
    env = dict(os.environ, SEQIK_LIB=_lib.WATCHDOG_LIB_PATH)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "HOST: seqik: HIP error: seqik_solve_seq: stage pipeline watchdog" in out, out
    assert "CLEARED" in out and "NAN" in out and "NO NAN" not in out, out
    assert "DEVICE: seqik: HIP error: seqik_check_faults: stage pipeline watchdog" in out, out
    assert "SERIAL OK" in out, out
    # the product build: same calls, no fault
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    pose = np.stack([z[f"{l}_pose"][:40] for l in legs])[None]
    params = [lib.leg_params_from_arrays(*leg_arrays(z, l)[1:]) for l in legs]
    assert np.isfinite(lib.solve_seq(pose, params, pipeline=2)["angles"]).all()
    lib.check_faults()


def test_latency_build_of_the_pipeline_equals_the_plain_build(lib):
    """the 256-register LATENCY build of the stage pipeline (run_stage<..., LAT>: Jacobian, gradient, scaling
    carried in registers across rejected trials behind a wave-uniform gate, branch-free reflective step) exists only in the
    WPE == 2 pipeline kernels, which the host harness cannot instantiate.  SeqikOptions.reserved[3] = 4 / 5 runs the same
    call on the plain instantiation: bit equality on the shipped 6000-frame recordings (RF + LF: 2 chains) and on the six
    legs of the locomotion recording, lane pairs on (2 vs 4) and off (3 vs 5), against the lane-per-chain kernels, and on
    the frame-chunk path, whose speculative and repair kernels have their own LAT instantiations."""
    for name, legs in (("anipose_shipped", ["RF", "LF"]), ("df3d_1000", None)):
        z = load_golden(name)
        legs = legs or [str(l) for l in z["legs"]]
        pose, params = _stack(z, legs), _params(lib, z, legs)
        serial = lib.solve_seq(pose, params, want_fk=True, pipeline=1)
        runs = {p: lib.solve_seq(pose, params, want_fk=True, pipeline=p) for p in (2, 3, 4, 5)}
        for p, out in runs.items():
            assert np.array_equal(out["angles"], serial["angles"]), (name, p)
            assert np.array_equal(out["fk"], serial["fk"]), (name, p)
        chunk_lat = lib.solve_seq(pose, params, want_fk=True, frame_chunk=-1, pipeline=2)
        chunk_plain = lib.solve_seq(pose, params, want_fk=True, frame_chunk=-1, pipeline=4)
        assert np.array_equal(chunk_lat["angles"], chunk_plain["angles"]) and np.array_equal(chunk_lat["fk"], chunk_plain["fk"])
        assert chunk_lat["chunk_stats"] == chunk_plain["chunk_stats"]
    # replicated thin waves with several chains per wave (lanes 2..16), both builds
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    many = np.concatenate([_stack(z, legs)] * 40)                      # 240 chains of 100 frames
    params = _params(lib, z, legs)
    want = lib.solve_seq(many, params, want_fk=False, pipeline=1)["angles"]
    for lanes in (2, 5, 16):
        for p in (2, 4):
            got = lib.solve_seq(many, params, want_fk=False, pipeline=p, lanes_per_wave=lanes)["angles"]
            assert np.array_equal(got, want), (lanes, p)


def test_fault_words_are_per_stream(lib):
    """ADVICE r4 / ABI 6: a watchdog fault is reported to the word of the (device, stream) the launch was made on, so a host
    thread that enters the library on ANOTHER stream neither sees nor clears it.  Diagnostic build (watchdog limit of one
    pass), child process: a piped launch on stream A faults; a launch on stream B is accepted (ABI 5 refused it and cleared
    A's fault); check_faults(B) is clean; check_faults(A) raises; afterwards everything is clean."""
    from seqikpy_amd import _lib
    if not os.path.exists(_lib.WATCHDOG_LIB_PATH):
        _lib.build_watchdog_variant()
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from seqikpy_amd import _lib
z = np.load(%r)
legs = [str(l) for l in z["legs"]]
pose = np.stack([z[f"{l}_pose"][:40] for l in legs])[None]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
d_pose = torch.from_numpy(pose).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# (round-5 advice) a caller that POLLS transient streams must not use up the (device, stream) -> word table: the check paths only
# look a stream up; with the 63 slots gone A and B would share the last word and B's check below would take A's fault
for k in range(200):
    _lib.check_faults(0x7000 + 8 * k)
a_ang = torch.zeros((1, len(legs), 40, 7), dtype=torch.float64, device="cuda")
b_ang = torch.zeros_like(a_ang)
torch.cuda.synchronize()
_lib.solve_seq_device(d_pose.data_ptr(), 1, len(legs), 40, params, a_ang.data_ptr(), pipeline=2, stream=sa.cuda_stream)
sa.synchronize()
print("A NAN" if bool(torch.isnan(a_ang).any()) else "A CLEAN")
_lib.solve_seq_device(d_pose.data_ptr(), 1, len(legs), 40, params, b_ang.data_ptr(), pipeline=1, stream=sb.cuda_stream)
sb.synchronize()
print("B ACCEPTED", "B FINITE" if bool(torch.isfinite(b_ang).all()) else "B BAD")
_lib.check_faults(sb.cuda_stream)
print("B NO FAULT")
try:
    _lib.check_faults(sa.cuda_stream)
    print("A: no error")
except _lib.SeqikLibraryError as e:
    print("A:", e)
_lib.check_faults(sa.cuda_stream); _lib.check_faults()
print("ALL CLEAR")
''' % (PKG_PARENT, ROOT, os.path.join(GOLDEN, "df3d_100.npz"))
    env = dict(os.environ, SEQIK_LIB=_lib.WATCHDOG_LIB_PATH)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "A NAN" in out and "B ACCEPTED B FINITE" in out and "B NO FAULT" in out, out
    assert "A: seqik: HIP error: seqik_check_faults_stream: stage pipeline watchdog" in out, out
    assert "ALL CLEAR" in out, out


def test_chain_queue_of_the_fused_kernel_bit_for_bit(lib, oracle):
    """Round 6: SeqikOptions.reserved[0] = 128 ... 4096 -- a wavefront of the single-launch kernel owns a POOL of chains and a
    lane that has finished its chain takes the next one (run_stage<..., QUEUE>, seqik_fused_queue_kernel).  Which lane walks a
    chain must not enter its arithmetic: == the plain launch bit for bit (angles and FK) for pools of 128 / 192 / 1024 / 4096
    chains (more than a leg has: one pool per leg; ragged last pools: 1500 = 11 x 128 + 92 sequences), with and without FK,
    with per-chain warm starts, dense and planar layouts, RAW key points with the alignment fused; sampled chains == the
    oracle; options the queue cannot serve are refused, not ignored."""
    import torch
    from seqikpy_amd import data, synthetic, utils
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    S, N, L = 1500, 24, len(legs)
    pose = synthetic.synthetic_pose(S, N, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="iid", seed=606)
    params = [lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    plain = lib.solve_seq(pose, params, want_fk=True, pipeline=1)
    rng = np.random.default_rng(3)
    for s, li in zip(rng.integers(0, S, 8), rng.integers(0, L, 8)):
        seg, b, seeds = oracle.leg_params(legs[li], data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
        ref = oracle.seq_leg(pose[s, li], seg, b, seeds)
        assert np.array_equal(plain["angles"][s, li], ref["angles"]) and np.array_equal(plain["fk"][s, li], ref["fk"])
    for pool in (128, 192, 1024, 4096):
        q = lib.solve_seq(pose, params, want_fk=True, pipeline=1, lanes_per_wave=pool)
        assert np.array_equal(q["angles"], plain["angles"]) and np.array_equal(q["fk"], plain["fk"]), pool
    q = lib.solve_seq(pose, params, want_fk=False, pipeline=1, lanes_per_wave=256)
    assert np.array_equal(q["angles"], plain["angles"]) and q["fk"] is None
    # per-chain warm starts (init_angles) travel with the chain a lane takes
    init = plain["angles"][:, :, -1].copy()
    a = lib.solve_seq(pose, params, want_fk=True, pipeline=1, init_angles=init)
    b = lib.solve_seq(pose, params, want_fk=True, pipeline=1, init_angles=init, lanes_per_wave=128)
    assert np.array_equal(a["angles"], b["angles"]) and np.array_equal(a["fk"], b["fk"]) and not np.array_equal(a["angles"], plain["angles"])
    # RAW key points, alignment fused into the frame start
    scales, fixed = 1.0 + 0.3 * rng.random(L), rng.normal(0.0, 1.5, (L, 3))
    tcs = [np.asarray(data.TEMPLATE_NMF_LOCOMOTION[f"{l}_Coxa"], dtype=np.float64) for l in legs]
    affs = [lib.make_affine(fixed[i], scales[i], tcs[i]) for i in range(L)]
    raw = np.stack([(pose[:, i] - tcs[i]) / scales[i] + fixed[i] for i in range(L)], 1)
    fa = lib.solve_seq(raw, params, want_fk=True, pipeline=1, affine=affs)
    fb = lib.solve_seq(raw, params, want_fk=True, pipeline=1, affine=affs, lanes_per_wave=192)
    assert np.array_equal(fa["angles"], fb["angles"]) and np.array_equal(fa["fk"], fb["fk"])
    # the benchmark's path: planar device buffers, asynchronous entry point, two launches in flight on two streams
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [(torch.zeros((S, L, 7, N), dtype=torch.float64, device="cuda"), torch.zeros((S, L, N, 9, 3), dtype=torch.float64, device="cuda"))
            for _ in streams]
    torch.cuda.synchronize()      # the upload and the zero fills ran on the default stream: done before the side streams start
    for k, pool in enumerate((128, 256)):
        lib.solve_seq_device(d_pose.data_ptr(), S, L, N, params, outs[k][0].data_ptr(), outs[k][1].data_ptr(), stream=streams[k].cuda_stream,
                             layout=lib.planar_layout(N), lanes_per_wave=pool, pipeline=1)
    torch.cuda.synchronize()
    lib.check_faults()
    for d_ang, d_fk in outs:
        assert np.array_equal(d_ang.cpu().numpy().transpose(0, 1, 3, 2), plain["angles"]) and np.array_equal(d_fk.cpu().numpy(), plain["fk"])
    # refused, not ignored: not a multiple of 64, too large, one launch per stage, diagnostics, frame chunks, the stage pipeline
    for kw in (dict(lanes_per_wave=100), dict(lanes_per_wave=8192), dict(lanes_per_wave=128, staged=1), dict(lanes_per_wave=128, want_diag=True),
               dict(lanes_per_wave=128, frame_chunk=8), dict(lanes_per_wave=128, interleave_legs=1)):
        with pytest.raises(ValueError):
            lib.solve_seq(pose[:64], params, **({"pipeline": 1} | kw))
    with pytest.raises(ValueError):
        lib.solve_seq(pose[:64], params, lanes_per_wave=128)      # 384 chains: the library would take the stage pipeline
