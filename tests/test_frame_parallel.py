"""Frame-parallel (verified chunk) mode: equals the serial solve to ~tol; the CPU tier exercises the
orchestration with the oracle standing in for the library, the GPU tier runs the real thing."""
import numpy as np
import pytest

from conftest import LF_DEGENERATE, load_golden

from seqikpy_amd import _lib, frame_parallel


def _oracle_solve_seq(oracle, z, legs):
    def solve(pose, legs_params, want_fk=True, affine=None, device=0, init_angles=None, **_):
        S, L, N = pose.shape[:3]
        ang = np.zeros((S, L, N, 7))
        fk = np.zeros((S, L, N, 9, 3))
        for s in range(S):
            for li, leg in enumerate(legs):
                r = oracle.seq_leg(pose[s, li], z[f"{leg}_seg"], z[f"{leg}_bounds"], z[f"{leg}_seeds"],
                                   init=None if init_angles is None else init_angles[s, li])
                ang[s, li], fk[s, li] = r["angles"], r["fk"]
        return dict(angles=ang, fk=fk)
    return solve


def test_orchestration_with_oracle_backend(oracle, monkeypatch):
    z = load_golden("df3d_1000")
    legs = ["RF", "LM", "RH"]
    monkeypatch.setattr(_lib, "solve_seq", _oracle_solve_seq(oracle, z, legs))
    pose = np.stack([z[f"{l}_pose"][:330] for l in legs])[None]
    serial = _lib.solve_seq(pose, legs)
    for chunk, halo in ((64, 16), (50, 8), (330, 16), (400, 0)):
        stats = {}
        out = frame_parallel.solve_frame_parallel(pose, legs, chunk=chunk, halo=halo, tol=1e-6, stats=stats)
        assert np.abs(out["angles"] - serial["angles"]).max() < 2e-5, (chunk, halo, stats)
        assert np.abs(out["fk"] - serial["fk"]).max() < 2e-5
        assert stats["chunks"] == -(-330 // chunk)
    # halo 0: every chunk fails verification and is repaired from the true state -> bit-identical
    stats = {}
    out = frame_parallel.solve_frame_parallel(pose, legs, chunk=64, halo=0, stats=stats)
    assert np.array_equal(out["angles"], serial["angles"]) and np.array_equal(out["fk"], serial["fk"])
    assert stats["repaired"] == 5 and stats["rounds"] == 5
    # tol 0: nothing is accepted -> also exact
    out = frame_parallel.solve_frame_parallel(pose, legs, chunk=100, halo=10, tol=0.0)
    assert np.array_equal(out["angles"], serial["angles"])


def test_basin_jumps_are_caught_by_verification(oracle, monkeypatch):
    """Grooming data: a seed restart can sit in another local minimum ('warm start is
    load-bearing').  LF frame 352 of the shipped recording is such a point for an 8-frame halo (the
    restarted TiTa_pitch sits on its -150 deg bound, 2.6 rad away); verification must reject the chunk
    that starts there and the repair must bring it back onto the serial trajectory."""
    z = load_golden("anipose_shipped")
    legs = ["LF"]
    monkeypatch.setattr(_lib, "solve_seq", _oracle_solve_seq(oracle, z, legs))
    pose = z["LF_pose"][None, None, 304:304 + 192]          # chunks of 48 frames: boundary at frame 352
    serial = _lib.solve_seq(pose, legs)
    unverified = frame_parallel.solve_frame_parallel(pose, legs, chunk=48, halo=8, tol=10.0)
    assert np.abs(unverified["angles"] - serial["angles"]).max() > 1.0   # the hazard is real
    stats = {}
    out = frame_parallel.solve_frame_parallel(pose, legs, chunk=48, halo=8, tol=1e-6, stats=stats)
    assert np.abs(out["angles"] - serial["angles"]).max() < 1e-5
    assert stats["repaired"] >= 1


@pytest.mark.gpu
def test_frame_parallel_on_gpu(hiplib):
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    serial = hiplib.solve_seq(pose, params, want_fk=True)
    stats = {}
    out = frame_parallel.solve_frame_parallel(pose, params, chunk=64, halo=16, stats=stats)
    assert np.abs(out["angles"] - serial["angles"]).max() < 2e-5
    assert np.abs(out["fk"] - serial["fk"]).max() < 2e-5
    exact = frame_parallel.solve_frame_parallel(pose, params, chunk=128, halo=0)
    assert np.array_equal(exact["angles"], serial["angles"])
    # Python API on the shipped grooming recording (config 4): within the parity bar of the shipped golden
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
    assert ik.frame_chunk_stats["chunks"] == 2 * 188  # 6000 frames / 32 per chunk, 2 legs (device-side chunks)
