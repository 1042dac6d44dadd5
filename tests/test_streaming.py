"""Streaming row (BASELINE config 5): slabs pushed through `seqik_stream_*` give the same bits as one
`seqik_solve_seq` call -- with a ragged last slab, with the alignment fused (RAW key points), in the
planar layout from pinned memory, and "in time" (carry) against the unsplit recording."""
import numpy as np
import pytest

from conftest import DOFS, leg_arrays, load_golden

from seqikpy_amd import data
from seqikpy_amd.alignment import AlignPose


def _params(lib, z, legs):
    return [lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]


def _windows(z, legs, key, offs, T):
    base = np.stack([z[f"{l}_{key}"] for l in legs])  # (L, 1000, 5, 3)
    return np.stack([base[:, o:o + T] for o in offs])


# ---------------------------------------------------------------- CPU tier: argument checks (no GPU needed)
def test_stream_open_rejects_bad_arguments(hiplib):
    from seqikpy_amd.streaming import SeqikStream
    z = load_golden("df3d_100")
    _, seg, b, seeds = leg_arrays(z, "RF")
    good = hiplib.leg_params_from_arrays(seg, b, seeds)
    with pytest.raises(ValueError, match="bad sizes"):
        SeqikStream([good], slab_seq=4, n_frames=8, n_slots=0)
    with pytest.raises(ValueError, match="bad sizes"):
        SeqikStream([good], slab_seq=0, n_frames=8)
    bad = seeds.copy()
    bad[1] = 4.0
    with pytest.raises(ValueError, match="outside of provided bounds"):
        SeqikStream([hiplib.leg_params_from_arrays(seg, b, bad)], slab_seq=4, n_frames=8)
    lib = hiplib.load()
    assert lib.seqik_stream_submit(None, None, 1, None, None) == hiplib.ERR_ARG
    assert lib.seqik_stream_wait(None) == hiplib.ERR_ARG
    assert lib.seqik_stream_close(None) == hiplib.SEQIK_OK
    assert lib.seqik_host_register(None, 8) == hiplib.ERR_ARG


# ---------------------------------------------------------------- GPU tier
@pytest.fixture(scope="module")
def lib(hiplib):
    if hiplib.load().seqik_device_count() < 1:
        pytest.fail("GPU tier needs a GPU: the HIP path must not be skipped silently")
    return hiplib


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [1, 3])
def test_slabs_of_sequences_equal_one_call(lib, n_slots):
    from seqikpy_amd.streaming import solve_streamed
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = _windows(z, legs, "pose", [(37 * i) % 950 for i in range(23)], 48)  # 23 sequences: ragged vs slab 5
    whole = lib.solve_seq(pose, params, want_fk=True)
    st = solve_streamed(pose, params, slab_seq=5, n_slots=n_slots)
    assert np.array_equal(st["angles"], whole["angles"])
    assert np.array_equal(st["fk"], whole["fk"])
    no_fk = solve_streamed(pose, params, slab_seq=7, want_fk=False, n_slots=n_slots)
    assert no_fk["fk"] is None and np.array_equal(no_fk["angles"], whole["angles"])


@pytest.mark.gpu
def test_carried_slabs_in_time_equal_the_unsplit_recording(lib, oracle):
    """6 legs x 1000 frames in slabs of 96 frames (10 carried slabs + a 40-frame remainder) == one call
    == the oracle's serial frame loop."""
    from seqikpy_amd.streaming import solve_streamed_in_time
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    whole = lib.solve_seq(pose, params, want_fk=True)
    st = solve_streamed_in_time(pose, params, slab_frames=96)
    assert np.array_equal(st["angles"], whole["angles"])
    assert np.array_equal(st["fk"], whole["fk"])
    ref = oracle.seq_leg(*leg_arrays(z, "LM"))
    assert np.array_equal(st["angles"][0, legs.index("LM")], ref["angles"])


@pytest.mark.gpu
def test_streamed_raw_key_points_with_fused_alignment_planar_pinned(lib):
    """Config 5 in small: RAW key points in pinned memory, planar layout, SeqikAffine fused, 3 slots."""
    from seqikpy_amd.streaming import PinnedArray, SeqikStream
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    raw = {f"{l}_leg": z[f"{l}_raw"] for l in legs}
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    affs = [lib.make_affine(*al.leg_affine(raw[f"{l}_leg"], l)) for l in legs]
    params = _params(lib, z, legs)
    T, slab = 32, 4
    offs = [(53 * i) % 960 for i in range(12)]
    pose_raw = _windows(z, legs, "raw", offs, T)
    pose_al = _windows(z, legs, "pose", offs, T)
    ref = lib.solve_seq(pose_al, params, want_fk=True)
    L = len(legs)
    n_slabs = len(offs) // slab
    bufs = [(PinnedArray((slab, L, 5, T, 3)), PinnedArray((slab, L, 7, T)), PinnedArray((slab, L, T, 9, 3)))
            for _ in range(n_slabs)]
    with SeqikStream(params, slab, T, affine=affs, layout=lib.planar_layout(T), want_fk=True, n_slots=3) as st:
        for k, (p, a, f) in enumerate(bufs):
            p.array[...] = pose_raw[k * slab:(k + 1) * slab].transpose(0, 1, 3, 2, 4)
            st.submit(p.array, a.array, f.array)
        st.wait()
    for k, (p, a, f) in enumerate(bufs):
        sl = slice(k * slab, (k + 1) * slab)
        assert np.array_equal(a.array.transpose(0, 1, 3, 2), ref["angles"][sl])
        assert np.array_equal(f.array, ref["fk"][sl])
        p.free(); a.free(); f.free()


@pytest.mark.gpu
def test_stream_slot_reuse_and_reset_carry(lib):
    """More slabs than slots (slot reuse blocks correctly); reset_carry starts new recordings."""
    from seqikpy_amd.streaming import SeqikStream
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]  # (1, 6, 100, 5, 3)
    whole = lib.solve_seq(pose, params, want_fk=False)
    T = 20
    outs = []
    with SeqikStream(params, 1, T, want_fk=False, n_slots=2, carry=True) as st:
        for rep in range(2):
            for k in range(5):
                a = np.empty((1, 6, T, 7))
                st.submit(np.ascontiguousarray(pose[:, :, k * T:(k + 1) * T]), a)
                outs.append(a)
            st.wait()
            st.reset_carry()
    for rep in range(2):
        got = np.concatenate(outs[rep * 5:(rep + 1) * 5], axis=2)
        assert np.array_equal(got, whole["angles"])
    with SeqikStream(params, 1, T, want_fk=True, n_slots=2) as st:
        with pytest.raises(ValueError):
            st.submit(np.ascontiguousarray(pose[:, :, :T]), np.empty((1, 6, T, 7)))  # fk missing
        with pytest.raises(ValueError):
            st.submit(np.ascontiguousarray(np.concatenate([pose[:, :, :T]] * 2)), np.empty((2, 6, T, 7)),
                      np.empty((2, 6, T, 9, 3)))  # n_seq exceeds the slab size


@pytest.mark.gpu
def test_generic_chains_through_the_stream(lib):
    """generic=True: LegInvKinGeneric chains streamed in slabs (and carried in time) == seqik_solve_generic."""
    from seqikpy_amd.streaming import SeqikStream
    zg = load_golden("generic_rf_100")
    legs = ["RF", "LF"]
    params = [lib.leg_params_from_arrays(zg[f"{l}_seg"], zg[f"{l}_bounds"], zg[f"{l}_seeds"]) for l in legs]
    pose = np.stack([zg[f"{l}_pose"][:60] for l in legs])[None]                  # (1, 2, 60, 5, 3)
    whole = lib.solve_generic(pose, params)
    outs = []
    with SeqikStream(params, 1, 20, want_fk=True, n_slots=2, carry=True, generic=True) as st:
        for k in range(3):
            a, f = np.empty((1, 2, 20, 7)), np.empty((1, 2, 20, 9, 3))
            st.submit(np.ascontiguousarray(pose[:, :, 20 * k:20 * (k + 1)]), a, f)
            outs.append((a, f))
        st.wait()
    assert np.array_equal(np.concatenate([o[0] for o in outs], axis=2), whole["angles"])
    assert np.array_equal(np.concatenate([o[1] for o in outs], axis=2), whole["fk"])
    seqs = np.concatenate([pose[:, :, :30], pose[:, :, 30:]])                     # 2 independent sequences of 30
    ref = lib.solve_generic(seqs, params)
    with SeqikStream(params, 1, 30, want_fk=False, generic=True) as st:
        a0, a1 = np.empty((1, 2, 30, 7)), np.empty((1, 2, 30, 7))
        st.submit(np.ascontiguousarray(seqs[:1]), a0)
        st.submit(np.ascontiguousarray(seqs[1:]), a1)
        st.wait()
    assert np.array_equal(a0, ref["angles"][:1]) and np.array_equal(a1, ref["angles"][1:])


@pytest.mark.gpu
def test_one_recording_streamed_in_time_slabs_with_frame_chunks(hiplib):
    """BASELINE config 5 read literally: ONE long recording streamed in time slabs (carried warm start), every slab cut
    into frame chunks on the device.  Equal to the serial walk to the chunk tolerance's noise floor, and the chunked
    slabs really are chunked (a slab's first chunk continues bit-identically from the carried state)."""
    from conftest import load_golden
    from seqikpy_amd.streaming import solve_streamed_in_time
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]           # (1, 6, 1000, 5, 3)
    serial = hiplib.solve_seq(pose, params)
    exact = solve_streamed_in_time(pose, params, slab_frames=256)   # 3 slabs of 256 + a rest of 232: bit-exact
    assert np.array_equal(exact["angles"], serial["angles"]) and np.array_equal(exact["fk"], serial["fk"])
    chunked = solve_streamed_in_time(pose, params, slab_frames=256, frame_chunk=-1)
    assert np.abs(chunked["angles"] - serial["angles"]).max() < 2e-5
    assert np.abs(chunked["fk"] - serial["fk"]).max() < 2e-5
    assert np.array_equal(chunked["angles"][:, :, :8], serial["angles"][:, :, :8])        # chunk 0 of slab 0
    assert not np.array_equal(chunked["angles"], serial["angles"])                         # ... and the rest is chunked


@pytest.mark.gpu
def test_stream_slots_follow_a_padded_layout_and_bad_layouts_are_rejected(lib):
    """the device slots of a stream are sized from the caller's layout (chain stride may exceed the dense
    size), and a layout whose key points / angles would reach outside a chain's stride is refused at open time."""
    from seqikpy_amd.streaming import SeqikStream
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = _params(lib, z, legs)
    T, S, L = 24, 3, len(legs)
    pose = _windows(z, legs, "pose", [0, 30, 61], T)                      # (S, L, T, 5, 3)
    ref = lib.solve_seq(pose, params, want_fk=True)
    pad_p, pad_a = 15 * T + 40, 7 * T + 24                                # padded chain strides (doubles)
    lay = lib.SeqikLayout(pad_p, 3 * T, 3, pad_a, T, 1)                   # planar inside a padded chain block
    hp = np.full((S * L, pad_p), np.nan)
    hp[:, :15 * T] = pose.transpose(0, 1, 3, 2, 4).reshape(S * L, 15 * T)
    hp[:, 15 * T:] = 0.0
    ha = np.full((S * L, pad_a), -7.0)
    hf = np.empty((S, L, T, 9, 3))
    with SeqikStream(params, S, T, layout=lay, want_fk=True, n_slots=2) as st:
        for _ in range(3):                                                # slot reuse with the padded size
            st.submit(hp, ha, hf)
        st.wait()
    got = ha[:, :7 * T].reshape(S, L, 7, T).transpose(0, 1, 3, 2)
    assert np.array_equal(got, ref["angles"]) and np.array_equal(hf, ref["fk"])
    for bad in (lib.SeqikLayout(15 * T - 1, 3 * T, 3, 7 * T, T, 1),      # last key point outside the chain stride
                lib.SeqikLayout(15 * T, 3 * T, 3, 7 * T - 1, T, 1),      # last angle outside
                lib.SeqikLayout(15 * T, 0, 3, 7 * T, T, 1)):             # zero stride
        with pytest.raises(ValueError):
            SeqikStream(params, S, T, layout=bad)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_config5_at_its_full_size_ten_million_frames_streamed(lib, oracle):
    """BASELINE config 5 at its LITERAL size (round-5 review, item 5): 10 M frames x 6 legs -- 20 slabs of 7 813 sequences of 64
    frames = 10 000 640 frames, 60 M leg-frames -- streamed from pinned host memory over 3 slots, RAW key points, the alignment
    (reference: seqikpy/alignment.py:436-487) fused into the kernel prologue, warm start from the previous frame
    (leg_inverse_kinematics.py:272), 7 angles + 9 x 3 FK back in pinned memory.  One distinct slab is generated and re-submitted
    (what `scripts/stream_config5.py --unique 1` and bench.py --detail time; the kernels cannot tell), so size-independent
    properties take the place of an oracle run over 60 M leg-frames:
      * EVERY one of the 20 slabs comes back, bit for bit the same result (the output slots are poisoned with NaN before each use);
      * finite everywhere, every angle within its limits, FK == forward kinematics of the returned angles (independent numpy FK);
      * sampled chains == the C oracle on the host-aligned key points, bit for bit (angles and FK);
      * streamed == one direct blocking call on a sampled piece of the slab, bit for bit."""
    import time
    from seqikpy_amd import synthetic, utils
    from seqikpy_amd.streaming import PinnedArray, SeqikStream
    legs = data.LEGS
    L, T, S, n_slabs, n_slots = len(legs), 64, 7813, 20, 3
    assert n_slabs * S * T >= 10_000_000
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    params = [lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    rng = np.random.default_rng(5)
    scales, fixed = 1.0 + 0.4 * rng.random(L), rng.normal(0.0, 2.0, (L, 3))          # a made-up camera frame per leg
    tcs = [np.asarray(data.TEMPLATE_NMF_LOCOMOTION[f"{l}_Coxa"], dtype=np.float64) for l in legs]
    affs = [lib.make_affine(fixed[i], scales[i], tcs[i]) for i in range(L)]
    al = synthetic.synthetic_pose(S, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="iid",
                                  seed=synthetic.SEED_BASE + 5)
    raw = np.empty_like(al)
    for i in range(L):
        raw[:, i] = (al[:, i] - tcs[i]) / scales[i] + fixed[i]
    slab = PinnedArray((S, L, 5, T, 3))
    slab.array[...] = raw.transpose(0, 1, 3, 2, 4)
    outs = [(PinnedArray((S, L, 7, T)), PinnedArray((S, L, T, 9, 3))) for _ in range(n_slots)]
    first, seconds = None, 0.0
    with SeqikStream(params, S, T, affine=affs, layout=lib.planar_layout(T), want_fk=True, n_slots=n_slots) as st:
        done = 0
        while done < n_slabs:
            group = min(n_slots, n_slabs - done)
            for k in range(group):
                outs[k][0].array.fill(np.nan)
                outs[k][1].array.fill(np.nan)
            t0 = time.perf_counter()
            for k in range(group):
                st.submit(slab.array, outs[k][0].array, outs[k][1].array)
            st.wait()
            seconds += time.perf_counter() - t0
            if first is None:
                first = (outs[0][0].array.copy(), outs[0][1].array.copy())
            for k in range(group):
                assert np.array_equal(outs[k][0].array, first[0]) and np.array_equal(outs[k][1].array, first[1]), done + k
            done += group
    lib.check_faults()
    ang, fk = first[0].transpose(0, 1, 3, 2), first[1]                                  # (S, L, T, 7), (S, L, T, 9, 3)
    assert np.isfinite(ang).all() and np.isfinite(fk).all()
    for li, leg in enumerate(legs):
        lb = np.array([data.BOUNDS_LOCOMOTION[f"{leg}_{d}"][0] for d in DOFS])
        ub = np.array([data.BOUNDS_LOCOMOTION[f"{leg}_{d}"][1] for d in DOFS])
        assert (ang[:, li] >= lb).all() and (ang[:, li] <= ub).all()
        seg = [body[f"{leg}_{s}"] for s in data.SEGMENTS]
        kp = synthetic.leg_forward_kinematics(ang[:, li], seg) + tcs[li]      # fused alignment: the origin is the template's coxa
        assert np.abs(kp - fk[:, li][:, :, [0, 4, 6, 7, 8]]).max() < 1e-12
    # the oracle gets what AlignPose.align_leg makes of the RAW key points (numpy, three separately rounded operations)
    pick = np.random.default_rng(11)
    for s, li in zip(pick.integers(0, S, 18), pick.integers(0, L, 18)):
        aligned = (raw[s, li] - fixed[li]) * scales[li] + tcs[li]
        seg, b, seeds = oracle.leg_params(legs[li], data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
        ref = oracle.seq_leg(aligned, seg, b, seeds)
        assert np.array_equal(ang[s, li], ref["angles"]) and np.array_equal(fk[s, li], ref["fk"]), (s, li)
    direct = lib.solve_seq(raw[1000:1512], params, want_fk=True, affine=affs)
    assert np.array_equal(ang[1000:1512], direct["angles"]) and np.array_equal(fk[1000:1512], direct["fk"])
    units = n_slabs * S * L * T
    print(f"config 5 at full size: {units} leg-frames in {seconds:.2f} s (groups of {n_slots} slabs, PCIe-inclusive) = {units / seconds:.3g} /s")
    assert units / seconds > 1e6        # the north star's target rate, PCIe and fused alignment included
