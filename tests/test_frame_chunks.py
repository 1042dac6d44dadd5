"""Frame chunks (SeqikOptions.frame_chunk): one long recording cut into concurrently solved pieces, verified and
repaired on the device.  CPU tier: the device core's CHUNKED code (host harness) against the oracle-built model of the
launch sequence, bit for bit; the model against the serial walk (tolerance semantics).  GPU tier: the library itself."""
import os

import numpy as np
import pytest

from chunk_model import chunked_oracle
from conftest import leg_arrays, load_golden

CASES = [("df3d_1000", "RF", slice(0, 300), 16, 8), ("df3d_1000", "LH", slice(100, 420), 8, 4),
         ("anipose_shipped", "RF", slice(0, 500), 32, 8),
         ("anipose_shipped", "LF", slice(200, 420), 8, 8),   # through the kinematic-singularity episode: repairs
         ("anipose_shipped", "LF", slice(240, 400), 8, 2)]   # short run-in: many inconsistent chunks, cascades


@pytest.mark.parametrize("name,leg,sl,chunk,halo", CASES)
def test_chunked_core_equals_model(oracle, host_harness, name, leg, sl, chunk, halo):
    pose, seg, b, seeds = leg_arrays(load_golden(name), leg)
    pose = pose[sl]
    for rounds in (3, 1):
        m = chunked_oracle(oracle, pose, seg, b, seeds, chunk, halo, rounds=rounds)
        hh = host_harness.run_chunked(pose, seg, b, seeds, chunk, halo, rounds=rounds)
        assert np.array_equal(hh["angles"], m["angles"])
        assert np.array_equal(hh["fk"], m["fk"])
        assert np.array_equal(hh["stats"], m["stats"]) and np.array_equal(hh["flags"], m["flags"])


def test_plan_is_a_function_of_the_recording_length_alone(hiplib):
    """seqik_frame_chunk_plan (no GPU needed) == the restatement in chunk_model.plan; the automatic geometry for the
    BASELINE configs (config 1 / 2: 4 + 4, config 4: 8 + 8, one 1M-frame recording: 32 + 8)."""
    from chunk_model import plan
    for n in (1, 47, 48, 100, 1000, 1360, 1365, 1366, 6000, 262144, 300000, 1_000_000, 2_000_000, 10_000_000):
        for kw in (dict(), dict(frame_chunk=16), dict(frame_chunk=16, frame_halo=3), dict(frame_chunk=-1, frame_halo=6),
                   dict(frame_chunk=8, frame_lead=8), dict(frame_chunk=0)):
            assert hiplib.frame_chunk_plan(n, **kw) == plan(n, **{"frame_chunk": -1, **kw}) or "frame_chunk" in kw and \
                hiplib.frame_chunk_plan(n, **kw) == plan(n, **kw), (n, kw)
    assert hiplib.frame_chunk_plan(100) == (4, 4, 25) and hiplib.frame_chunk_plan(1000) == (4, 4, 250)
    assert hiplib.frame_chunk_plan(6000) == (8, 8, 750) and hiplib.frame_chunk_plan(1_000_000) == (32, 8, 31250)
    assert hiplib.frame_chunk_plan(40) == (0, 0, 0)


@pytest.mark.parametrize("name,leg,sl", [("anipose_shipped", "LF", slice(240, 400)), ("df3d_1000", "RF", slice(0, 300)),
                                         ("synthetic_iid", "LF", slice(0, 256))])
def test_guard_of_the_automatic_mode_in_core_and_model(oracle, host_harness, name, leg, sl):
    """A chain of which more than one chunk in eight fails the first verification is walked serially (bit for bit the
    serial walk); others keep their chunks.  Device core (host build) == model, statistics and report included."""
    if name == "synthetic_iid":   # random poses: several equivalent leg configurations, run-ins land in another one
        from oracle import c_oracle
        from seqikpy_amd import data, synthetic, utils
        body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, data.LEGS)
        pose = synthetic.synthetic_pose(4, 64, [leg], data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                        variant="iid").reshape(256, 5, 3)
        seg, b, seeds = c_oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
    else:
        pose, seg, b, seeds = leg_arrays(load_golden(name), leg)
    pose = pose[sl]
    serial = oracle.seq_leg(pose, seg, b, seeds)
    tripped = []
    for chunk, halo in ((8, 2), (8, 8), (4, 4)):
        m = chunked_oracle(oracle, pose, seg, b, seeds, chunk, halo, guard=True)
        hh = host_harness.run_chunked(pose, seg, b, seeds, chunk, halo, guard=True)
        assert np.array_equal(hh["angles"], m["angles"]) and np.array_equal(hh["fk"], m["fk"])
        assert np.array_equal(hh["stats"], m["stats"]) and np.array_equal(hh["flags"], m["flags"])
        if m["stats"][8]:
            assert m["stats"][7] * 8 > m["stats"][0] and (m["flags"] & 8).all() and m["stats"][3:7].sum() == 0
            assert np.array_equal(m["angles"], serial["angles"]) and np.array_equal(m["fk"], serial["fk"])
        else:
            assert m["stats"][7] * 8 <= m["stats"][0] and not (m["flags"] & 8).any()
        tripped.append(bool(m["stats"][8]))
    # random poses trip the guard whatever the geometry, the recordings never do
    assert all(tripped) if name == "synthetic_iid" else not any(tripped)


def test_slab_with_lead_and_resume_in_the_model(oracle, host_harness):
    """A slab of a recording (frame_lead): its chunk 0 starts from a run-in like the others; once the true state in front
    of the slab is known the resume step settles it.  With chunks on the same global grid the slab equals the same frames
    of the whole-recording call, bit for bit; the host-built device core agrees with the model."""
    from chunk_model import ChunkedChain
    pose, seg, b, seeds = leg_arrays(load_golden("df3d_1000"), "LM")
    C, h, a = 16, 8, 320
    whole = chunked_oracle(oracle, pose[:640], seg, b, seeds, C, h)
    slab = ChunkedChain(oracle, pose[a - h:640], seg, b, seeds, C, h, lead=h)
    slab.speculate()
    slab.settle()                                        # nothing known about the left neighbour yet
    slab.settle(init=whole["angles"][a - 1], resume=True)
    assert np.array_equal(slab.angles[h:], whole["angles"][a:]) and np.array_equal(slab.fk[h:], whole["fk"][a:])
    hh = host_harness.run_chunked(pose[a - h:640], seg, b, seeds, C, h, init=whole["angles"][a - 1], lead=h)
    assert np.array_equal(hh["angles"][h:], whole["angles"][a:]) and np.array_equal(hh["fk"][h:], whole["fk"][a:])
    # a wrong left state makes chunk 0 inconsistent: it is repaired from it, exactly as the serial continuation
    wrong = whole["angles"][a - 1] + 1e-3
    bad = ChunkedChain(oracle, pose[a - h:640], seg, b, seeds, C, h, lead=h)
    bad.speculate()
    bad.settle()
    bad.settle(init=wrong, resume=True)
    assert bad.flags[0] & 3 == 3 and bad.stats[7] >= 1
    assert np.array_equal(bad.angles[h:h + C], oracle.seq_leg(pose[a:a + C], seg, b, seeds, init=wrong.copy())["angles"])


def test_every_chunk_ends_up_consistent_and_repairs_happen(oracle):
    """After the sweep every chunk's warm start is within tol of its predecessor's last frame; on the LF episode
    chunks do get repaired (the test would be vacuous otherwise)."""
    pose, seg, b, seeds = leg_arrays(load_golden("anipose_shipped"), "LF")
    m = chunked_oracle(oracle, pose[240:400], seg, b, seeds, 8, 2, tol=1e-6, rounds=2)
    assert m["stats"][7] > 0 and m["stats"][3] > 0
    serial = oracle.seq_leg(pose[240:400], seg, b, seeds)
    assert np.abs(m["angles"][:40] - serial["angles"][:40]).max() < 1e-4  # in front of the episode: well-posed


def test_zero_tolerance_is_the_serial_walk(oracle, host_harness):
    """chunk_tol < 0 (exact): a chunk is accepted only if the run-in reproduced the true state bit for bit, everything
    else is re-solved from the true state -> the serial result, bit for bit."""
    pose, seg, b, seeds = leg_arrays(load_golden("df3d_1000"), "RM")
    pose = pose[:120]
    serial = oracle.seq_leg(pose, seg, b, seeds)
    for rounds in (0, 2):
        hh = host_harness.run_chunked(pose, seg, b, seeds, 8, 4, tol=0.0, rounds=rounds)
        assert np.array_equal(hh["angles"], serial["angles"]) and np.array_equal(hh["fk"], serial["fk"])


def test_chunked_equals_serial_within_noise_floor_on_recordings(oracle):
    """Default parameters on whole recordings: the chunked result stays within 2e-5 rad of the serial walk (measured
    max 1.1e-5; the reference's own run-to-run noise is ~5e-5) outside the LF singularity episode."""
    for name, legs, n in (("df3d_1000", ["RF", "RM", "LH"], 1000), ("anipose_shipped", ["RF"], 2000)):
        z = load_golden(name)
        for leg in legs:
            pose, seg, b, seeds = leg_arrays(z, leg)
            serial = oracle.seq_leg(pose[:n], seg, b, seeds)
            m = chunked_oracle(oracle, pose[:n], seg, b, seeds, 8, 8)
            assert np.abs(m["angles"] - serial["angles"]).max() < 2e-5, (name, leg)
            assert np.abs(m["fk"] - serial["fk"]).max() < 2e-5


def test_continuation_with_init(oracle, host_harness):
    pose, seg, b, seeds = leg_arrays(load_golden("df3d_1000"), "LM")
    first = oracle.seq_leg(pose[:50], seg, b, seeds)
    init = first["angles"][-1]
    m = chunked_oracle(oracle, pose[50:200], seg, b, seeds, 16, 8, init=init)
    hh = host_harness.run_chunked(pose[50:200], seg, b, seeds, 16, 8, init=init)
    assert np.array_equal(hh["angles"], m["angles"])
    assert np.array_equal(hh["angles"][:16], oracle.seq_leg(pose[50:66], seg, b, seeds, init=init)["angles"])


# ---------------------------------------------------------------------------------------------------------------
# GPU tier: the library's device-side implementation (csrc/seqik_hip.hip "Frame chunks") through the C ABI
# ---------------------------------------------------------------------------------------------------------------
def _params(hiplib, z, legs):
    return [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]


def _model_all(oracle, z, legs, sl, chunk, halo, **kw):
    ms = [chunked_oracle(oracle, z[f"{l}_pose"][sl], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"], chunk, halo, **kw)
          for l in legs]
    stats = np.sum([m["stats"] for m in ms], 0)
    stats[1:3] = ms[0]["stats"][1:3]
    return np.stack([m["angles"] for m in ms]), np.stack([m["fk"] for m in ms]), stats[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("name,legs,sl,chunk,halo,rounds", [
    ("df3d_1000", None, slice(0, 1000), 16, 8, 3),                  # config 2: six legs, no repair expected
    ("anipose_shipped", ["RF", "LF"], slice(0, 6000), 8, 8, 3),     # config 4: repairs in the LF episode
    ("anipose_shipped", ["LF", "RF"], slice(200, 420), 8, 2, 1),    # short run-in, one round: the sweep works too
    ("anipose_shipped", ["LF"], slice(240, 400), 8, 2, 8),          # cascades through many rounds
    ("df3d_1000", ["RM", "LH"], slice(3, 998), 64, 16, 3),          # ragged last chunk, long halo
])
def test_hip_chunks_equal_model(oracle, hiplib, name, legs, sl, chunk, halo, rounds):
    z = load_golden(name)
    legs = legs or [str(l) for l in z["legs"]]
    pose = np.stack([z[f"{l}_pose"][sl] for l in legs])[None]
    out = hiplib.solve_seq(pose, _params(hiplib, z, legs), frame_chunk=chunk, frame_halo=halo, chunk_rounds=rounds)
    ang, fk, stats = _model_all(oracle, z, legs, sl, chunk, halo, rounds=rounds)
    assert np.array_equal(out["angles"][0], ang)
    assert np.array_equal(out["fk"][0], fk)
    got = np.array([out["chunk_stats"][k] for k in hiplib.CHUNK_STATS_FIELDS])
    assert np.array_equal(got, stats), (got, stats)


@pytest.mark.gpu
def test_hip_chunks_several_sequences_planar_layout_and_init(oracle, hiplib):
    """Device entry point: S sequences x L legs, planar layout, init_angles for chunk 0, lanes-per-wave overrides."""
    import torch
    z = load_golden("df3d_1000")
    legs = ["LF", "RH", "LM"]
    S, T, C, H = 3, 200, 16, 4
    pose = np.stack([np.stack([z[f"{l}_pose"][100 + 250 * s:100 + 250 * s + T] for l in legs]) for s in range(S)])
    first = [[oracle.seq_leg(z[f"{l}_pose"][90 + 250 * s:100 + 250 * s], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"])
              ["angles"][-1] for l in legs] for s in range(S)]
    init = np.array(first)
    want = np.stack([np.stack([chunked_oracle(oracle, pose[s, li], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"], C, H,
                                              init=init[s, li])["angles"] for li, l in enumerate(legs)]) for s in range(S)])
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_init = torch.from_numpy(init).cuda()
    for lanes in (0, 1, 5, 64):
        d_ang = torch.zeros((S, len(legs), 7, T), dtype=torch.float64, device="cuda")
        d_fk = torch.zeros((S, len(legs), T, 9, 3), dtype=torch.float64, device="cuda")
        d_stats = torch.zeros(hiplib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
        hiplib.solve_seq_device(d_pose.data_ptr(), S, len(legs), T, _params(hiplib, z, legs), d_ang.data_ptr(),
                                d_fk.data_ptr(), layout=hiplib.planar_layout(T), d_init=d_init.data_ptr(),
                                frame_chunk=C, frame_halo=H, lanes_per_wave=lanes, d_chunk_stats=d_stats.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_ang.cpu().numpy().transpose(0, 1, 3, 2), want), lanes
        assert int(d_stats[0]) == S * len(legs) * 13


@pytest.mark.gpu
def test_hip_exact_tolerance_and_automatic_mode(oracle, hiplib):
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    params = _params(hiplib, z, legs)
    serial = hiplib.solve_seq(pose, params)
    assert all(v == 0 for v in serial["chunk_stats"].values())
    exact = hiplib.solve_seq(pose[:, :, :160], params, frame_chunk=8, frame_halo=4, chunk_tol=-1.0, chunk_rounds=2)
    assert np.array_equal(exact["angles"], serial["angles"][:, :, :160])  # bit for bit the serial walk
    auto = hiplib.solve_seq(pose, params, frame_chunk=-1)
    st = auto["chunk_stats"]
    # a recording this short (750 chunks of 8 would leave most SIMDs idle) is cut into chunks of 4 after a run-in of 4
    assert st["chunks"] == 6 * 250 and st["frames_per_chunk"] == 4 and st["run_in_frames"] == 4
    # ... whatever else is in the call: the geometry is a function of the recording's length alone
    big = hiplib.solve_seq(np.ascontiguousarray(np.broadcast_to(pose, (2,) + pose.shape[1:])), params, frame_chunk=-1)
    assert big["chunk_stats"]["chunks"] == 2 * 6 * 250 and big["chunk_stats"]["frames_per_chunk"] == 4
    assert np.array_equal(big["angles"][0], big["angles"][1]) and np.array_equal(big["angles"][0], auto["angles"][0])
    assert np.array_equal(big["fk"][0], auto["fk"][0])
    assert np.abs(auto["angles"] - serial["angles"]).max() < 2e-5
    assert np.abs(auto["fk"] - serial["fk"]).max() < 2e-5
    short = hiplib.solve_seq(pose[:, :, :40], params, frame_chunk=-1)  # too short: serial
    assert short["chunk_stats"]["chunks"] == 0 and np.array_equal(short["angles"], serial["angles"][:, :, :40])
    # stage subsets and diagnostics are walked serially whatever frame_chunk says
    diag = hiplib.solve_seq(pose[:, :, :100], params, want_diag=True, frame_chunk=8)
    assert np.array_equal(diag["angles"], serial["angles"][:, :, :100]) and diag["chunk_stats"]["chunks"] == 0


@pytest.mark.gpu
def test_python_api_serial_walk_is_the_oracle_and_the_default_is_within_parity_of_shipped_golden(oracle, hiplib):
    """run_ik_and_fk(frame_parallel=False) is the reference's own order (serial walk == oracle bit for bit); the DEFAULT
    (frame_parallel="auto" since round 6) solves a recording of 48 frames and more in verified frame chunks: within the parity
    budget of the shipped outputs, with a report of where the recording was hard; SEQIK_FRAME_PARALLEL=0 restores the serial
    walk for a process; recordings shorter than 48 frames are walked serially either way."""
    from conftest import LF_DEGENERATE
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    za = load_golden("anipose_shipped")
    kc = KinematicChainSeq(BOUNDS, ["RF", "LF"])
    cut = LegInvKinSeq({"RF_leg": za["RF_pose"][:300]}, KinematicChainSeq(BOUNDS, ["RF"]), INITIAL_ANGLES, log_level="ERROR")
    serial_ang, _ = cut.run_ik_and_fk(frame_parallel=False)
    serial_ang = {k: v.copy() for k, v in serial_ang.items()}
    assert cut.frame_chunk_stats["chunks"] == 0 and cut.frame_chunk_report == {}
    ref = oracle.seq_leg(za["RF_pose"][:300], za["RF_seg"], za["RF_bounds"], za["RF_seeds"])
    assert np.array_equal(np.stack([serial_ang[f"Angle_RF_{d}"] for d in hiplib.DOFS], 1), ref["angles"])
    old = os.environ.pop("SEQIK_FRAME_PARALLEL", None)
    try:
        os.environ["SEQIK_FRAME_PARALLEL"] = "0"
        env_serial, _ = cut.run_ik_and_fk()
        assert cut.frame_chunk_stats["chunks"] == 0 and all(np.array_equal(env_serial[k], serial_ang[k]) for k in serial_ang)
        del os.environ["SEQIK_FRAME_PARALLEL"]
        short = LegInvKinSeq({"RF_leg": za["RF_pose"][:40]}, KinematicChainSeq(BOUNDS, ["RF"]), INITIAL_ANGLES, log_level="ERROR")
        short_ang, _ = short.run_ik_and_fk()
        assert short.frame_chunk_stats["chunks"] == 0 and np.array_equal(short_ang["Angle_RF_ThC_yaw"], serial_ang["Angle_RF_ThC_yaw"][:40])
        ik = LegInvKinSeq({"RF_leg": za["RF_pose"], "LF_leg": za["LF_pose"]}, kc, INITIAL_ANGLES, log_level="ERROR")
        ang, fk = ik.run_ik_and_fk()                    # the default
    finally:
        if old is not None:
            os.environ["SEQIK_FRAME_PARALLEL"] = old
    assert ik.frame_chunk_stats["chunks"] == 2 * 750 and ik.frame_chunk_stats["chains_walked_serially"] == 0
    got = np.stack([ang[f"Angle_RF_{d}"] for d in hiplib.DOFS], 1)
    assert np.abs(got - za["RF_angles"]).max() < 1e-4
    ok = np.ones(6000, bool)
    ok[LF_DEGENERATE[0]:LF_DEGENERATE[1]] = False
    got = np.stack([ang[f"Angle_LF_{d}"] for d in hiplib.DOFS], 1)
    assert np.abs(got - za["LF_angles"])[ok].max() < 1e-4
    assert np.abs(fk["RF_leg"][za["fk_frames"]] - za["RF_fk_cut"]).max() < 1e-4
    assert np.abs(serial_ang["Angle_RF_ThC_yaw"] - ang["Angle_RF_ThC_yaw"][:300]).max() < 2e-5
    # the report points at the LF kinematic-singularity episode (frames 284-301) and nowhere near it on RF
    rep = ik.frame_chunk_report
    assert rep["LF"]["frames_per_chunk"] == 8 and not rep["LF"]["walked_serially"]
    assert any(LF_DEGENERATE[0] - 8 <= t < LF_DEGENERATE[1] + 8 for t in rep["LF"]["failed_first_check"])
    assert rep["LF"]["frames_repaired"] >= 8
    assert not any(LF_DEGENERATE[0] - 8 <= t < LF_DEGENERATE[1] + 8 for t in rep["RF"]["failed_first_check"])


@pytest.mark.gpu
def test_default_of_frame_parallel_is_decided_by_evidence(hiplib):
    """Round-5 review, item 8: the evidence behind the default, in the GPU tier.  Over the shipped 6000-frame recording (RF + LF,
    fixture = the reference's own shipped output of real IKPy) and the df3d recording (6 legs x 1000 frames, fixture = the
    reference's source run): `frame_parallel="auto"` stays below 1e-4 rad everywhere outside the LF singularity episode, every
    value above 5e-5 is listed and is ALSO above 5e-5 on the serial walk (the chunks add nothing to the tail), the two modes are
    within 1.5e-5 rad of each other, and auto is at least 5 x faster on the call a drop-in user makes."""
    import time
    from conftest import LF_DEGENERATE
    from seqikpy_amd import data
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.utils import calculate_body_size
    cases = []
    za = load_golden("anipose_shipped")
    cases.append(("anipose_shipped", za, ["RF", "LF"], KinematicChainSeq(data.BOUNDS, ["RF", "LF"]), data.INITIAL_ANGLES, True))
    zd = load_golden("df3d_1000")
    legs6 = [str(l) for l in zd["legs"]]
    cases.append(("df3d_1000", zd, legs6, KinematicChainSeq(data.BOUNDS_LOCOMOTION, legs6, calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs6)),
                  data.INITIAL_ANGLES_LOCOMOTION, False))
    for name, z, legs, kc, init, mask_lf in cases:
        aligned = {f"{l}_leg": z[f"{l}_pose"] for l in legs}
        ref = np.stack([z[f"{l}_angles"] for l in legs])                                # (L, N, 7)
        ok = np.ones(ref.shape[:2], bool)
        if mask_lf:
            ok[legs.index("LF"), LF_DEGENERATE[0]:LF_DEGENERATE[1]] = False
        res, ms = {}, {}
        for mode in (False, "auto"):
            best = float("inf")
            for _ in range(3):
                ik = LegInvKinSeq(aligned, kc, init, log_level="ERROR")
                t0 = time.perf_counter()
                ang, _ = ik.run_ik_and_fk(frame_parallel=mode)
                best = min(best, time.perf_counter() - t0)
            res[mode] = np.stack([np.stack([ang[f"Angle_{l}_{d}"] for d in hiplib.DOFS], 1) for l in legs])
            ms[mode] = best * 1e3
            if mode:
                assert ik.frame_chunk_stats["chunks"] > 0 and ik.frame_chunk_stats["chains_walked_serially"] == 0
        err = {m: np.where(ok[:, :, None], np.abs(res[m] - ref), 0.0) for m in res}
        assert err["auto"].max() < 1e-4 and err[False].max() < 1e-4, name
        over = np.argwhere(err["auto"] > 5e-5)
        listed = [(legs[i], int(t), hiplib.DOFS[j], float(err["auto"][i, t, j]), float(err[False][i, t, j])) for i, t, j in over]
        print(f"{name}: auto {ms['auto']:.2f} ms, serial {ms[False]:.2f} ms; max |d theta| auto {err['auto'].max():.3g} serial {err[False].max():.3g}; "
              f"values over 5e-5 (leg, frame, joint, auto, serial): {listed}")
        assert len(listed) <= 2 and all(e_serial > 5e-5 for *_, e_serial in listed), listed
        assert np.where(ok[:, :, None], np.abs(res["auto"] - res[False]), 0.0).max() < 1.5e-5, name
        assert ms["auto"] * 5 < ms[False], (name, ms)


@pytest.mark.gpu
def test_python_api_explicit_chunk_parameters(hiplib):
    """run_ik_and_fk(frame_parallel=dict(chunk=..., halo=...)) on the shipped grooming recording (config 4): within the
    parity bar of the shipped outputs; unknown options and stage subsets are refused."""
    from conftest import LF_DEGENERATE
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    za = load_golden("anipose_shipped")
    ik = LegInvKinSeq({"RF_leg": za["RF_pose"], "LF_leg": za["LF_pose"]}, KinematicChainSeq(BOUNDS, ["RF", "LF"]),
                      INITIAL_ANGLES, log_level="ERROR")
    ang, fk = ik.run_ik_and_fk(frame_parallel=dict(chunk=32, halo=16))
    got = np.stack([ang[f"Angle_RF_{d}"] for d in hiplib.DOFS], 1)
    assert np.abs(got - za["RF_angles"]).max() < 1e-4
    ok = np.ones(6000, bool)
    ok[LF_DEGENERATE[0]:LF_DEGENERATE[1]] = False
    got = np.stack([ang[f"Angle_LF_{d}"] for d in hiplib.DOFS], 1)
    assert np.abs(got - za["LF_angles"])[ok].max() < 1e-4
    assert ik.frame_chunk_stats["chunks"] == 2 * 188 and ik.frame_chunk_report["RF"]["frames_per_chunk"] == 32
    with pytest.raises(ValueError):
        ik.run_ik_and_fk(frame_parallel=dict(chunks=3))
    with pytest.raises(ValueError):
        ik.run_ik_and_fk(frame_parallel=True, stages=[1, 2])


@pytest.mark.gpu
def test_a_recording_gives_the_same_bits_alone_in_a_batch_and_in_a_longer_batch(hiplib):
    """Round-2 review item 3: under frame_parallel="auto" one recording alone == the same recording inside
    run_ik_and_fk_many == inside a batch twice as long == through pipeline.run_body_ik, bit for bit."""
    from seqikpy_amd import data
    from seqikpy_amd.batch import run_ik_and_fk_many
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.pipeline import run_body_ik
    from seqikpy_amd.utils import calculate_body_size
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    kc = KinematicChainSeq(data.BOUNDS_LOCOMOTION, legs, calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs))
    cuts = [(0, 400), (300, 700), (600, 1000), (100, 500), (250, 650), (5, 405)]
    recs = [{f"{l}_leg": z[f"{l}_pose"][a:b] for l in legs} for a, b in cuts]
    alone = LegInvKinSeq(recs[1], kc, data.INITIAL_ANGLES_LOCOMOTION, log_level="ERROR")
    ang1, fk1 = alone.run_ik_and_fk(frame_parallel="auto")
    assert alone.frame_chunk_stats["chunks"] == 6 * 100
    reports = []
    for batch in (recs[:3], recs):
        many = run_ik_and_fk_many(batch, kc, data.INITIAL_ANGLES_LOCOMOTION, frame_parallel="auto", reports=reports)
        ang, fk = many[1]
        assert all(np.array_equal(ang[k], ang1[k]) for k in ang1) and all(np.array_equal(fk[k], fk1[k]) for k in fk1)
        assert reports[1] == alone.frame_chunk_report
    body, fkb = run_body_ik(recs[1], kc, data.TEMPLATE_NMF_LOCOMOTION, data.INITIAL_ANGLES_LOCOMOTION, frame_parallel="auto")
    assert all(np.array_equal(body[k], ang1[k]) for k in ang1) and all(np.array_equal(fkb[k], fk1[k]) for k in fk1)
    # the DEFAULT of all three entry points is that mode ...
    d1, _ = LegInvKinSeq(recs[1], kc, data.INITIAL_ANGLES_LOCOMOTION, log_level="ERROR").run_ik_and_fk()
    d2 = run_ik_and_fk_many(recs[:3], kc, data.INITIAL_ANGLES_LOCOMOTION)[1][0]
    d3, _ = run_body_ik(recs[1], kc, data.TEMPLATE_NMF_LOCOMOTION, data.INITIAL_ANGLES_LOCOMOTION)
    assert all(np.array_equal(d1[k], ang1[k]) and np.array_equal(d2[k], ang1[k]) and np.array_equal(d3[k], ang1[k]) for k in ang1)
    # ... and the serial walk of all three agrees as well
    s1, _ = LegInvKinSeq(recs[1], kc, data.INITIAL_ANGLES_LOCOMOTION, log_level="ERROR").run_ik_and_fk(frame_parallel=False)
    s1 = {k: v.copy() for k, v in s1.items()}
    s2 = run_ik_and_fk_many(recs[:3], kc, data.INITIAL_ANGLES_LOCOMOTION, frame_parallel=False)[1][0]
    s3, _ = run_body_ik(recs[1], kc, data.TEMPLATE_NMF_LOCOMOTION, data.INITIAL_ANGLES_LOCOMOTION, frame_parallel=False)
    assert all(np.array_equal(s1[k], s2[k]) and np.array_equal(s1[k], s3[k]) for k in s1)


# ---------------------------------------------------------------------------------------------------------------
# Stage pipeline (SeqikOptions.reserved[3]): four wavefronts per group of chains, hand-off through LDS
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 1, 3, 5, 7, 12, 20, 32, 33, 64])
def test_stage_pipeline_equals_oracle_bit_for_bit(oracle, hiplib, lanes):
    """Serial walk on the stage pipeline == lane-per-chain kernel == C oracle, whatever the number of chains per
    workgroup (every lane has its own ring slots and counters).  Up to 32 chains per wavefront every chain runs on
    two or more lanes and neighbouring lanes split the passes between them (lane pairs): same bits, also against the
    pipeline without pairs (pipeline = 3)."""
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(hiplib, z, legs)
    pose = np.stack([np.stack([z[f"{l}_pose"][o:o + 90] for l in legs]) for o in (0, 200, 411, 640, 900)])  # S = 5
    piped = hiplib.solve_seq(pose, params, pipeline=2, lanes_per_wave=lanes)
    plain = hiplib.solve_seq(pose, params, pipeline=1)
    assert np.array_equal(piped["angles"], plain["angles"]) and np.array_equal(piped["fk"], plain["fk"])
    unpaired = hiplib.solve_seq(pose, params, pipeline=3, lanes_per_wave=lanes)
    assert np.array_equal(piped["angles"], unpaired["angles"]) and np.array_equal(piped["fk"], unpaired["fk"])
    for s, o in enumerate((0, 200, 411, 640, 900)):
        for li, l in enumerate(legs):
            ref = oracle.seq_leg(z[f"{l}_pose"][o:o + 90], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"])
            assert np.array_equal(piped["angles"][s, li], ref["angles"]) and np.array_equal(piped["fk"][s, li], ref["fk"])


@pytest.mark.gpu
def test_stage_pipeline_edge_cases(oracle, hiplib):
    """One frame, two frames (shorter than the ring), init_angles, no FK, nasty made-up legs, the LF episode."""
    from conftest import random_leg_case
    z = load_golden("anipose_shipped")
    params = _params(hiplib, z, ["RF", "LF"])
    pose = np.stack([z["RF_pose"][200:420], z["LF_pose"][200:420]])[None]
    ser = hiplib.solve_seq(pose, params, pipeline=1)
    for n in (1, 2, 3, 220):
        p = hiplib.solve_seq(pose[:, :, :n], params, pipeline=2, want_fk=(n != 3))
        assert np.array_equal(p["angles"], ser["angles"][:, :, :n])
    init = ser["angles"][:, :, 99]
    cont = hiplib.solve_seq(pose[:, :, 100:], params, pipeline=2, init_angles=init)
    assert np.array_equal(cont["angles"], ser["angles"][:, :, 100:]) and np.array_equal(cont["fk"], ser["fk"][:, :, 100:])
    rng = np.random.default_rng(5)
    for _ in range(24):
        kp, seg, b, seeds = random_leg_case(rng, 40)
        lp = hiplib.leg_params_from_arrays(seg, b, seeds)
        got = hiplib.solve_seq(kp[None, None], [lp], pipeline=2)
        ref = oracle.seq_leg(kp, seg, b, seeds)
        assert np.array_equal(got["angles"][0, 0], ref["angles"]) and np.array_equal(got["fk"][0, 0], ref["fk"])


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,halo,rounds", [(8, 8, 3), (8, 2, 1), (16, 4, 3)])
def test_chunks_on_the_stage_pipeline_equal_model(oracle, hiplib, chunk, halo, rounds):
    z = load_golden("anipose_shipped")
    legs = ["LF", "RF"]
    sl = slice(200, 520)
    pose = np.stack([z[f"{l}_pose"][sl] for l in legs])[None]
    out = hiplib.solve_seq(pose, _params(hiplib, z, legs), frame_chunk=chunk, frame_halo=halo, chunk_rounds=rounds, pipeline=2)
    ang, fk, stats = _model_all(oracle, z, legs, sl, chunk, halo, rounds=rounds)
    assert np.array_equal(out["angles"][0], ang) and np.array_equal(out["fk"][0], fk)
    assert np.array_equal(np.array([out["chunk_stats"][k] for k in hiplib.CHUNK_STATS_FIELDS]), stats)


@pytest.mark.gpu
def test_automatic_chunks_give_way_to_the_serial_walk_per_chain_when_speculation_fails(oracle, hiplib):
    """Random poses that span several equivalent leg configurations: half of the run-ins end in another configuration than
    the serial walk.  The automatic mode notices PER CHAIN, on the device (more than one chunk in eight inconsistent at
    the first verification), and walks those chains serially, bit for bit -- host and device entry points alike, and
    without touching the well-behaved chains of the same call; explicit chunk parameters are honoured as given."""
    import torch
    from seqikpy_amd import data, synthetic, utils
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    pose = synthetic.synthetic_pose(4, 64, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="iid")
    rnd = np.ascontiguousarray(pose.transpose(1, 0, 2, 3, 4).reshape(6, 256, 5, 3))      # one random recording per leg
    z = load_golden("df3d_1000")
    real = np.stack([z[f"{l}_pose"][:256] for l in legs])
    rec = np.stack([rnd, real, rnd[:, ::-1]])                                            # sequences 0 and 2 are hopeless
    params = [hiplib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    serial = hiplib.solve_seq(rec, params)
    auto = hiplib.solve_seq(rec, params, frame_chunk=-1, want_chunk_flags=True)
    st, fl = auto["chunk_stats"], auto["chunk_flags"]
    assert st["chunks"] == 3 * 6 * 64 and st["frames_per_chunk"] == 4
    walked = (fl & hiplib.CHUNK_FLAG_SERIAL).any(-1)                                     # (3, 6) chains walked serially
    assert walked[0].all() and walked[2].all() and not walked[1].any()
    assert st["chains_walked_serially"] == 12 and st["chunks_of_those_chains"] == 12 * 64
    assert np.array_equal(auto["angles"][[0, 2]], serial["angles"][[0, 2]]) and np.array_equal(auto["fk"][[0, 2]], serial["fk"][[0, 2]])
    alone = hiplib.solve_seq(rec[1:2], params, frame_chunk=-1)                           # the real recording: as if alone
    assert np.array_equal(auto["angles"][1], alone["angles"][0]) and np.array_equal(auto["fk"][1], alone["fk"][0])
    assert alone["chunk_stats"]["chains_walked_serially"] == 0
    for leg_i, l in enumerate(legs):                                                     # == the model, per chain
        m = chunked_oracle(oracle, rec[0, leg_i], *[z[f"{l}_{k}"] for k in ("seg", "bounds", "seeds")], 4, 4, guard=True)
        assert m["stats"][8] == 1 and np.array_equal(auto["angles"][0, leg_i], m["angles"])
        assert np.array_equal(fl[0, leg_i], m["flags"])
    # the device entry point has the same guard (it used to live in the host entry point only)
    d_pose = torch.from_numpy(rec).cuda()
    d_ang = torch.zeros(rec.shape[:3] + (7,), dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(hiplib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
    hiplib.solve_seq_device(d_pose.data_ptr(), 3, 6, 256, params, d_ang.data_ptr(), frame_chunk=-1, d_chunk_stats=d_stats.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_ang.cpu().numpy(), auto["angles"]) and int(d_stats[8]) == 12
    forced = hiplib.solve_seq(rec[:1], params, frame_chunk=8)
    assert forced["chunk_stats"]["chunks"] == 6 * 32 and forced["chunk_stats"]["chains_walked_serially"] == 0
    # and on a real recording the automatic mode keeps its chunks
    lg = [str(l) for l in z["legs"]]
    ok = hiplib.solve_seq(np.stack([z[f"{l}_pose"] for l in lg])[None], _params(hiplib, z, lg), frame_chunk=-1)
    assert ok["chunk_stats"]["chunks"] == 1500 and ok["chunk_stats"]["chains_walked_serially"] == 0


@pytest.mark.gpu
def test_slab_with_lead_and_resume_on_the_device(oracle, hiplib):
    """frame_lead / chunk_states / chunk_resume through the C ABI == the model (tests/chunk_model.py): a slab of a
    recording whose chunk 0 is settled in a second call once the true state in front of it is known."""
    import torch
    from chunk_model import ChunkedChain
    z = load_golden("df3d_1000")
    legs = ["LM", "RH"]
    C, h, a, b = 16, 8, 320, 650
    params = _params(hiplib, z, legs)
    pose = np.stack([z[f"{l}_pose"][a - h:b] for l in legs])[None]
    n = pose.shape[2]
    K = hiplib.frame_chunk_plan(n, C, h, h)[2]
    for wrong in (0.0, 1e-3):
        models = []
        for l in legs:
            m = ChunkedChain(oracle, z[f"{l}_pose"][a - h:b], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"], C, h, lead=h)
            m.speculate()
            m.settle()
            models.append(m)
        d_pose = torch.from_numpy(pose).cuda()
        d_ang = torch.zeros((1, 2, n, 7), dtype=torch.float64, device="cuda")
        d_fk = torch.zeros((1, 2, n, 9, 3), dtype=torch.float64, device="cuda")
        d_states = torch.zeros((1, 2, K, 7), dtype=torch.float64, device="cuda")
        d_flags = torch.zeros((1, 2, K), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros(hiplib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
        kw = dict(frame_chunk=C, frame_halo=h, frame_lead=h, d_chunk_states=d_states.data_ptr(), d_chunk_flags=d_flags.data_ptr(),
                  d_chunk_stats=d_stats.data_ptr())
        hiplib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), **kw)
        torch.cuda.synchronize()
        for li, m in enumerate(models):
            assert np.array_equal(d_ang[0, li, h:].cpu().numpy(), m.angles[h:]) and np.array_equal(d_states[0, li].cpu().numpy(), m.ss)
        whole = [oracle.seq_leg(z[f"{l}_pose"][:a], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"])["angles"][-1] + wrong for l in legs]
        d_init = torch.from_numpy(np.stack(whole)[None]).cuda()
        hiplib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), d_init=d_init.data_ptr(),
                                chunk_resume=1, **kw)
        torch.cuda.synchronize()
        for li, m in enumerate(models):
            m.settle(init=whole[li], resume=True)
            assert np.array_equal(d_ang[0, li, h:].cpu().numpy(), m.angles[h:]), (wrong, li)
            assert np.array_equal(d_fk[0, li, h:].cpu().numpy(), m.fk[h:])
            assert np.array_equal(d_flags[0, li].cpu().numpy(), m.flags)
        assert int(d_stats[7]) == sum(int(m.stats[7]) for m in models)
        assert (int(d_stats[7]) > 0) == (wrong > 0)


@pytest.mark.gpu
def test_chunks_with_fused_alignment_and_several_recordings(hiplib):
    """Frame chunks of RAW key points with AlignPose.align_leg fused into the kernels == frame chunks of the pre-aligned
    key points, for several recordings in one call and on both kernels."""
    from seqikpy_amd import data
    from seqikpy_amd.alignment import AlignPose
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    raw = {f"{l}_leg": z[f"{l}_raw"] for l in legs}
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    affs = [hiplib.make_affine(*al.leg_affine(raw[f"{l}_leg"], l)) for l in legs]
    params = _params(hiplib, z, legs)
    cuts = [(0, 300), (350, 650), (690, 990)]
    pose_raw = np.stack([np.stack([z[f"{l}_raw"][a:b] for l in legs]) for a, b in cuts])
    pose_al = np.stack([np.stack([z[f"{l}_pose"][a:b] for l in legs]) for a, b in cuts])
    for pl in (1, 2):
        fused = hiplib.solve_seq(pose_raw, params, affine=affs, frame_chunk=16, frame_halo=8, pipeline=pl)
        plain = hiplib.solve_seq(pose_al, params, frame_chunk=16, frame_halo=8, pipeline=pl)
        assert fused["chunk_stats"]["chunks"] == 3 * 6 * 19
        assert np.array_equal(fused["angles"], plain["angles"]) and np.array_equal(fused["fk"], plain["fk"])
    serial = hiplib.solve_seq(pose_al, params)
    assert np.abs(plain["angles"] - serial["angles"]).max() < 2e-5
