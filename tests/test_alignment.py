"""Alignment row: host AlignPose vs the reference's (fixture df3d_1000.npz holds the
un-aligned key points and the output of the reference's AlignPose.align_pose), and the fused
kernel prologue vs aligning first (bit for bit)."""
import numpy as np
import pytest

from conftest import leg_arrays, load_golden

from seqikpy_amd import data
from seqikpy_amd.alignment import AlignPose


@pytest.fixture(scope="module")
def df3d():
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    raw = {f"{l}_leg": z[f"{l}_raw"] for l in legs}
    return z, legs, raw


def test_host_alignment_reproduces_the_reference_bit_for_bit(df3d):
    z, legs, raw = df3d
    al = AlignPose(raw, legs_list=legs, include_claw=False, body_template=data.TEMPLATE_NMF_LOCOMOTION,
                   log_level="ERROR")
    aligned = al.align_pose()
    for leg in legs:
        assert np.array_equal(aligned[f"{leg}_leg"], z[f"{leg}_pose"]), leg
        assert aligned[f"{leg}_leg"].shape == (1000, 5, 3)
    fixed, scale, tc = al.leg_affine(raw["RF_leg"], "RF")
    assert fixed.shape == (3,) and scale > 0 and np.array_equal(tc, data.TEMPLATE_NMF_LOCOMOTION["RF_Coxa"])


def test_include_claw_changes_the_scale(df3d):
    _, legs, raw = df3d
    a = AlignPose(raw, legs, include_claw=False, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    b = AlignPose(raw, legs, include_claw=True, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    assert a.leg_affine(raw["LM_leg"], "LM")[1] != b.leg_affine(raw["LM_leg"], "LM")[1]


def test_fused_alignment_equals_align_then_solve_on_host(df3d, oracle, host_harness):
    """Device core (host build) with the affine fused == oracle on the host-aligned key points."""
    z, legs, raw = df3d
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    for leg in ("RF", "LH"):
        _, seg, b, seeds = leg_arrays(z, leg)
        aff = al.leg_affine(raw[f"{leg}_leg"], leg)
        fused = host_harness.run(raw[f"{leg}_leg"][:300], seg, b, seeds, affine=aff)
        ref = oracle.seq_leg(z[f"{leg}_pose"][:300], seg, b, seeds)
        assert np.array_equal(fused["angles"], ref["angles"])
        assert np.array_equal(fused["fk"], ref["fk"])
        assert np.array_equal(fused["nfev"], ref["nfev"])


@pytest.mark.gpu
def test_fused_alignment_on_gpu(df3d, hiplib, oracle):
    """Config 5 in small: RAW key points + SeqikAffine through the C ABI == align first, then solve."""
    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.utils import calculate_body_size
    z, legs, raw = df3d
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    affs = [hiplib.make_affine(*al.leg_affine(raw[f"{l}_leg"], l)) for l in legs]
    pose_raw = np.stack([raw[f"{l}_leg"] for l in legs])[None]
    pose_al = np.stack([z[f"{l}_pose"] for l in legs])[None]
    fused = hiplib.solve_seq(pose_raw, params, want_fk=True, affine=affs)
    plain = hiplib.solve_seq(pose_al, params, want_fk=True)
    assert np.array_equal(fused["angles"], plain["angles"])
    assert np.array_equal(fused["fk"], plain["fk"])
    ref = oracle.seq_leg(*leg_arrays(z, "RM"))
    assert np.array_equal(fused["angles"][0, legs.index("RM")], ref["angles"])
    # Python API: LegInvKinSeq(raw, ..., leg_affine=AlignPose.leg_affines())
    body = calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    ik = LegInvKinSeq(raw, KinematicChainSeq(data.BOUNDS_LOCOMOTION, legs, body), data.INITIAL_ANGLES_LOCOMOTION,
                      log_level="ERROR", leg_affine=al.leg_affines())
    ang, fk = ik.run_ik_and_fk(frame_parallel=False)  # serial walk: bit-comparable with the oracle
    assert np.array_equal(ang["Angle_RM_FTi_pitch"], ref["angles"][:, 5])
    assert np.array_equal(fk["RM_leg"], ref["fk"])


def test_full_alignment_incl_antennae_bit_identical_to_reference():
    """anipose_raw_cut.npz: un-aligned legs / antennae / thorax and the reference's align_pose output."""
    from seqikpy_amd.data import NMF_TEMPLATE
    z = load_golden("anipose_raw_cut")
    raw = {str(k): z[f"raw_{k}"] for k in z["segments"]}
    out = AlignPose(raw, ["RF", "LF"], body_template=NMF_TEMPLATE, log_level="ERROR").align_pose()
    assert list(out.keys()) == ["R_head", "RF_leg", "L_head", "LF_leg", "Neck"]
    for k, v in out.items():
        assert np.array_equal(v, z[f"aligned_{k}"]), k
    assert out["Neck"].shape == (1, 1, 3)


def test_converters(tmp_path):
    from seqikpy_amd.alignment import (convert_from_anipose_to_dict, convert_from_df3d_to_dict,
                                       convert_from_df3dpp_to_dict)
    from seqikpy_amd.data import PTS2ALIGN
    rng = np.random.default_rng(3)
    names = sorted({kp for kps in PTS2ALIGN.values() for kp in kps})
    ani = {f"{kp}_{ax}": rng.normal(size=11) for kp in names for ax in "xyz"}
    d = convert_from_anipose_to_dict(ani, PTS2ALIGN)
    assert list(d.keys()) == list(PTS2ALIGN.keys()) and d["RF_leg"].shape == (11, 5, 3) and d["R_head"].shape == (11, 2, 3)
    assert np.array_equal(d["LF_leg"][:, 3, 1], ani["tibia_tarsus_L_y"])
    arr = rng.normal(size=(7, 38, 3))
    d = convert_from_df3d_to_dict(arr, {"RF_leg": np.arange(0, 5), "LH_leg": np.arange(29, 34)})
    assert np.array_equal(d["LH_leg"], arr[:, 29:34])
    pp = {"RM_leg": {kp: {"raw_pos_aligned": rng.normal(size=(9, 3))} for kp in ["Coxa", "Femur", "Tibia", "Tarsus", "Claw"]}}
    d = convert_from_df3dpp_to_dict(pp)
    assert d["RM_leg"].shape == (9, 5, 3) and np.array_equal(d["RM_leg"][:, 4], pp["RM_leg"]["Claw"]["raw_pos_aligned"])
    with pytest.raises(FileNotFoundError):
        AlignPose.from_file_path(tmp_path, file_name="pose3d.*", legs_list=["RF"])
    import pickle
    with open(tmp_path / "pose3d.h5", "wb") as f:
        pickle.dump(ani, f)
    al = AlignPose.from_file_path(tmp_path, file_name="pose3d.*", convert_func=convert_from_anipose_to_dict,
                                  legs_list=["RF", "LF"], log_level="ERROR")
    assert al.pose_data_dict["Thorax"].shape == (11, 3, 3)


class _FakeAlignStats:
    """numpy stand-in for _lib.AlignStats (CPU tier: checks the finishing formulas, not the GPU sort)."""

    def __init__(self, n_legs, capacity, device=0):
        self.pose = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def add(self, pose, **_):
        self.pose.append(pose)

    def finish(self, ranks):
        pose = np.concatenate([p.transpose(1, 0, 2, 3, 4).reshape(p.shape[1], -1, 5, 3) for p in self.pose], axis=1)
        out = np.zeros((pose.shape[0], 7, len(ranks)))
        for li in range(pose.shape[0]):
            series = [pose[li, :, 0, a] for a in range(3)]
            lengths = np.linalg.norm(np.diff(pose[li], axis=1), axis=2)
            series += [lengths[:, i] for i in range(4)]
            for j, v in enumerate(series):
                out[li, j] = np.sort(v)[np.asarray(ranks)]
        return out


def test_gpu_statistics_path_finishing_formulas_equal_numpy(df3d, monkeypatch):
    """leg_affines(on_gpu=True) applies numpy's quantile interpolation to exact order statistics: with a numpy
    sort standing in for the GPU the constants equal the host path bit for bit (several lengths: the
    interpolation weight changes with N)."""
    from seqikpy_amd import _lib
    monkeypatch.setattr(_lib, "AlignStats", _FakeAlignStats)
    z, legs, raw = df3d
    for n in (1000, 999, 37, 12):
        cut = {k: v[:n] for k, v in raw.items()}
        al = AlignPose(cut, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
        host, dev = al.leg_affines(), al.leg_affines(on_gpu=True)
        for leg in legs:
            assert np.array_equal(host[leg][0], dev[leg][0]) and host[leg][1] == dev[leg][1], (n, leg)


@pytest.mark.gpu
def test_alignment_statistics_on_gpu(df3d, hiplib):
    """seqik_align_stats_*: the GPU's order statistics give bit-identical affine constants; slabs added in two
    pieces and a device-resident planar slab give the same answer."""
    import torch
    z, legs, raw = df3d
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    host, dev = al.leg_affines(), al.leg_affines(on_gpu=True)
    for leg in legs:
        assert np.array_equal(host[leg][0], dev[leg][0]) and host[leg][1] == dev[leg][1], leg
    pose = np.stack([raw[f"{l}_leg"] for l in legs])[None]          # (1, 6, 1000, 5, 3)
    ranks = [0, 449, 450, 549, 550, 999]
    with hiplib.AlignStats(6, 1000) as st:
        st.add(pose)
        whole = st.finish(ranks)
        st.reset()
        st.add(np.ascontiguousarray(pose[:, :, 600:]))
        st.add(np.ascontiguousarray(pose[:, :, :600]))
        assert np.array_equal(st.finish(ranks), whole)
        st.reset()
        # 10 "sequences" of 100 frames in the planar device layout of the streaming path
        seqs = np.ascontiguousarray(pose[0].reshape(6, 10, 100, 5, 3).transpose(1, 0, 3, 2, 4))   # (10, 6, 5, 100, 3)
        d = torch.from_numpy(seqs).cuda()
        st.add(d.data_ptr(), n_seq=10, n_frames=100, layout=hiplib.planar_layout(100), on_device=True)
        torch.cuda.synchronize()
        assert np.array_equal(st.finish(ranks), whole)
    srt = np.sort(raw["RM_leg"][:, 0, 1])
    assert np.array_equal(whole[legs.index("RM"), 1], srt[ranks])
    lengths = np.sort(np.linalg.norm(np.diff(raw["LH_leg"], axis=1), axis=2)[:, 2])
    assert np.array_equal(whole[legs.index("LH"), 5], lengths[ranks])


def test_linear_quantile_index_equals_numpy_for_every_n():
    """the GPU path must interpolate between the same two order statistics with the same gamma as
    np.quantile.  On the series 0, 1, ..., n-1 the quantile is the virtual index itself, so numpy's own result exposes
    both: checked for every n up to 3000 and for large recordings.  (The installed numpy 2.2 evaluates the "linear"
    method as (n - 1) q; the helper asks numpy's own method table, so another numpy's expression is followed too.)"""
    from seqikpy_amd.alignment import linear_quantile_index

    def lerp(a, b, g):  # numpy.lib._function_base_impl._lerp
        r = a + (b - a) * g
        return b - (b - a) * (1 - g) if g >= 0.5 else r

    ns = list(range(1, 3001)) + [6000, 99991, 1_000_000, 10_000_019, 2**31 - 1]
    for n in ns:
        for q in (0.45, 0.55, 0.5 - 0.05, 0.5 + 0.05, 0.0, 1.0):
            lo, hi, g = linear_quantile_index(n, q)
            assert 0 <= lo <= hi <= n - 1 and 0.0 <= g < 1.0
            if n <= 100_000:
                want = np.quantile(np.arange(n, dtype=np.float64), q)
                assert lerp(float(lo), float(hi), g) == want, (n, q)


def test_reference_held_pins_six_legs_notebook_scale_factors_and_shipped_alignment():
    """Pins the reference itself holds for the alignment row on ALL SIX legs with the locomotion template (nothing here
    was produced by the build container): the six `find_scale_leg` values printed in the stored output of
    examples/seqikpy_locomotion.ipynb cell 6 (seqikpy/alignment.py:417-423) and the aligned poses that cell exported,
    shipped as data/df3d_pose_result__210902_PR_Fly1/pose3d_aligned.pkl (alignment.py:392-434, 436-487).  Fixture:
    df3d_align_pins.npz (oracle/gen_golden.py::gen_df3d_align_pins)."""
    z = load_golden("df3d_align_pins")
    legs = [str(l) for l in z["legs"]]
    raw = {f"{l}_leg": z[f"{l}_raw"] for l in legs}
    al = AlignPose(raw, legs_list=legs, include_claw=False, body_template=data.TEMPLATE_NMF_LOCOMOTION,
                   body_size=None, log_level="ERROR")
    for leg, printed in zip(legs, z["printed_scale_factors"]):
        scale = al.find_scale_leg(leg, al.get_mean_length(raw[f"{leg}_leg"], segment_is_leg=True))
        assert float(scale) == float(printed), (leg, scale, printed)      # repr round trip: every printed digit
        assert al.leg_affine(raw[f"{leg}_leg"], leg)[1] == float(printed)
    aligned = al.align_pose()
    assert list(aligned.keys()) == [f"{l}_leg" for l in legs]
    for leg in legs:
        assert np.array_equal(aligned[f"{leg}_leg"], z[f"{leg}_shipped_aligned"]), leg   # max |delta| = 0
    # and the shipped file is the input of the df3d_100 leg fixture: the chain of custody of config 2's short cut
    z100 = load_golden("df3d_100")
    for leg in legs:
        assert np.array_equal(z100[f"{leg}_pose"], z[f"{leg}_shipped_aligned"]), leg
