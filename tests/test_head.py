"""Head / antenna row (BASELINE config 4): closed-form angles vs the shipped
head_joint_angles.pkl and the reference's HeadInverseKinematics re-run here (anipose_head.npz)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import head_oracle  # noqa: E402

# acos is ill-conditioned near 0 / pi; libm, numpy and ocml differ in the last ulp of their inputs
TOL = 1e-6


@pytest.fixture(scope="module")
def z():
    return load_golden("anipose_head")


def test_head_oracle_vs_shipped_golden(z):
    out = head_oracle.head_angles(z["R_head"], z["L_head"], z["Neck"][:, 0], z["rest_head_pitch"][0],
                                  z["rest_antenna_pitch"][0])
    assert np.abs(out.T - z["shipped"]).max() < 1e-9
    assert np.abs(out.T - z["ref_run"]).max() < 1e-9


def test_head_device_code_on_host_vs_golden(z, host_harness):
    out = host_harness.head_angles(z["R_head"], z["L_head"], z["Neck"][:, 0], float(z["rest_head_pitch"][0]),
                                   float(z["rest_antenna_pitch"][0]))
    assert np.abs(out.T - z["shipped"]).max() < TOL


def test_head_device_code_on_host_given_roll_and_single_point_records(z, host_harness):
    """What the reference's per-quantity methods allow beyond compute_head_angles (head_inverse_kinematics.py:26, :242, :278):
    a head roll handed in by the caller, and head records with one key point per side (no antenna angles)."""
    args = (z["R_head"], z["L_head"], z["Neck"][:, 0], float(z["rest_head_pitch"][0]), float(z["rest_antenna_pitch"][0]))
    own = host_harness.head_angles(*args)
    # the frames' own roll handed back in: sin / cos of the angle instead of the normalised components, same to ~1e-12 after the acos
    again = host_harness.head_angles(*args, head_roll=own[0])
    assert np.array_equal(again[:3], own[:3]) and np.abs(again - own).max() < 1e-10
    rng = np.random.default_rng(3)
    for roll in (0.3, -1.1, rng.uniform(-np.pi, np.pi, 6000)):
        got = host_harness.head_angles(*args, head_roll=roll)
        want = head_oracle.head_angles(*args, head_roll=roll)
        assert np.array_equal(got[:3], own[:3])
        assert np.abs(got - want).max() < TOL
        assert np.abs(got[3:] - own[3:]).max() > 1e-3   # ... and it is a different derotation
    one = host_harness.head_angles(z["R_head"][:, :1], z["L_head"][:, :1], *args[2:], compute_ant=False)
    assert np.array_equal(one[:3], own[:3]) and not one[3:].any()
    three = host_harness.head_angles(np.concatenate([z["R_head"], z["R_head"][:, :1] + 7.0], 1),
                                     np.concatenate([z["L_head"], z["L_head"][:, :1] - 7.0], 1), *args[2:])
    assert np.array_equal(three, own)


def test_signed_angle_device_code_on_host(host_harness):
    """angle_between_segments (:163-182) for general vectors and axis."""
    rng = np.random.default_rng(4)
    v1, v2 = rng.normal(size=(5000, 3)), rng.normal(size=(5000, 3))
    for axis in (np.eye(3)[0], np.eye(3)[1], np.eye(3)[2], rng.normal(size=3)):
        want = head_oracle.signed_angle(v1, v2, axis)
        assert np.abs(host_harness.signed_angles(v1, v2, axis) - want).max() < 1e-7
    # one vector against many (the reference tiles it with get_plane), and the sign convention at det == 0
    want = head_oracle.signed_angle(np.eye(3)[1], v2, np.eye(3)[0])
    assert np.abs(host_harness.signed_angles(np.eye(3)[1], v2, np.eye(3)[0]) - want).max() < 1e-7
    assert host_harness.signed_angles([1.0, 0, 0], [0.0, 1, 0], [0.0, 0, 1])[0] == pytest.approx(np.pi / 2, abs=1e-15)
    assert host_harness.signed_angles([1.0, 0, 0], [0.0, 1, 0], [1.0, 0, 0])[0] == pytest.approx(-np.pi / 2, abs=1e-15)


def test_rest_angles_from_template(z):
    from seqikpy_amd.data import NMF_TEMPLATE
    from seqikpy_amd.head_inverse_kinematics import HeadInverseKinematics
    hk = HeadInverseKinematics({"R_head": z["R_head"], "L_head": z["L_head"], "Neck": z["Neck"]}, NMF_TEMPLATE,
                               log_level="ERROR")
    assert hk.rest_head_pitch == pytest.approx(z["rest_head_pitch"][0], abs=1e-15)
    assert hk.rest_antenna_pitch == pytest.approx(z["rest_antenna_pitch"][0], abs=1e-15)
    with pytest.raises(ValueError):
        HeadInverseKinematics({"R_head": z["R_head"]}, NMF_TEMPLATE)


@pytest.mark.gpu
def test_head_angles_on_gpu(z, hiplib, tmp_path):
    from seqikpy_amd.data import NMF_TEMPLATE
    from seqikpy_amd.head_inverse_kinematics import ANGLE_NAMES, HeadInverseKinematics
    hk = HeadInverseKinematics({"R_head": z["R_head"], "L_head": z["L_head"], "Neck": z["Neck"]}, NMF_TEMPLATE,
                               log_level="ERROR")
    ang = hk.compute_head_angles(export_path=tmp_path)
    assert list(ang.keys()) == ANGLE_NAMES == [str(n) for n in z["names"]]
    got = np.stack([ang[n] for n in ANGLE_NAMES], 1)
    assert got.shape == (6000, 7)
    assert np.abs(got - z["shipped"]).max() < TOL
    assert os.path.exists(tmp_path / "head_joint_angles.pkl")
    only_head = hk.compute_head_angles(compute_ant_angles=False)
    assert list(only_head.keys()) == ANGLE_NAMES[:3]
    assert np.array_equal(only_head["Angle_head_yaw"], ang["Angle_head_yaw"])
    # per-frame neck and empty input
    neck_n = np.repeat(z["Neck"], 6000, axis=0)
    hk2 = HeadInverseKinematics({"R_head": z["R_head"], "L_head": z["L_head"], "Neck": neck_n}, NMF_TEMPLATE,
                                log_level="ERROR")
    assert np.array_equal(hk2.compute_head_angles()["Angle_head_pitch"], ang["Angle_head_pitch"])
    assert hiplib.head_angles(z["R_head"][:0], z["L_head"][:0], z["Neck"][:, 0], 0.1, 0.2).shape == (7, 0)


@pytest.mark.gpu
def test_head_per_quantity_methods_on_gpu(z, hiplib):
    """Every public method of the reference's HeadInverseKinematics (:144-339): the three head angles one by one, the
    antenna angles with the head roll as an ARGUMENT, angle_between_segments, the array helpers; records with a single
    key point per side."""
    from scipy.spatial.transform import Rotation
    from seqikpy_amd.data import NMF_TEMPLATE
    from seqikpy_amd.head_inverse_kinematics import ANGLE_NAMES, Axes, HeadInverseKinematics
    pos = {"R_head": z["R_head"], "L_head": z["L_head"], "Neck": z["Neck"]}
    hk = HeadInverseKinematics(pos, NMF_TEMPLATE, log_level="ERROR")
    both = hk.compute_head_angles()
    col = {n: z["shipped"][:, i] for i, n in enumerate(ANGLE_NAMES)}
    roll = hk.compute_head_roll()
    for name, got in (("Angle_head_roll", roll), ("Angle_head_pitch", hk.compute_head_pitch()),
                      ("Angle_head_yaw", hk.compute_head_yaw())):
        assert np.array_equal(got, both[name]) and np.abs(got - col[name]).max() < TOL
    for side in ("L", "R", "l"):
        for kind, fn in (("yaw", hk.compute_antenna_yaw), ("pitch", hk.compute_antenna_pitch)):
            name = f"Angle_antenna_{kind}_{side.upper()}"
            got = fn(side=side, head_roll=roll)
            assert np.abs(got - both[name]).max() < 1e-10 and np.abs(got - col[name]).max() < TOL
    with pytest.raises(ValueError, match="Side should be either R or L"):
        hk.compute_antenna_yaw("X", roll)
    # a roll that is NOT the frames' own (the reference derotates by whatever it is handed)
    args = (z["R_head"], z["L_head"], z["Neck"][:, 0], hk.rest_head_pitch, hk.rest_antenna_pitch)
    for other in (roll + 0.3, np.full(6000, -0.7)):
        want = head_oracle.head_angles(*args, head_roll=other)
        assert np.abs(hk.compute_antenna_yaw("L", other) - want[3]).max() < TOL
        assert np.abs(hk.compute_antenna_pitch("L", other) - want[4]).max() < TOL
        assert np.abs(hk.compute_antenna_yaw("R", other) - want[5]).max() < TOL
        assert np.abs(hk.compute_antenna_pitch("R", other) - want[6]).max() < TOL
    dev = hiplib.head_angles(*args, head_roll=roll + 0.3)
    assert np.array_equal(dev[:3], np.stack([both[n] for n in ANGLE_NAMES[:3]]))
    # array helpers and attributes
    assert np.array_equal(hk.head_vector_mid, (z["R_head"][:, 0] + z["L_head"][:, 0]) * 0.5 - z["Neck"][:, 0])
    assert np.array_equal(hk.head_vector_horizontal, z["L_head"][:, 0] - z["R_head"][:, 0])
    assert np.array_equal(hk.get_head_vector("R"), z["Neck"][:, 0] - z["R_head"][:, 0])
    assert np.array_equal(hk.get_ant_vector("L"), z["L_head"][:, 1] - z["L_head"][:, 0])
    assert hk.get_plane(Axes.X_AXIS, 5).shape == (5, 3)
    v = hk.get_ant_vector("R")[:50]
    assert np.abs(hk.derotate_vector(0.4, v) - Rotation.from_euler("x", -0.4).apply(v)).max() < 1e-15
    # angle_between_segments as the reference uses it for the head roll, and for general operands
    hv = hk.head_vector_horizontal.copy()
    hv[:, 0] = 0
    got = HeadInverseKinematics.angle_between_segments(v1=hk.get_plane(Axes.Y_AXIS, 6000), v2=hv, rot_axis=Axes.X_AXIS)
    assert np.abs(got - roll).max() < 1e-10
    rng = np.random.default_rng(8)
    v1, v2, axis = rng.normal(size=(4097, 3)), rng.normal(size=(4097, 3)), rng.normal(size=3)
    assert np.abs(HeadInverseKinematics.angle_between_segments(v1, v2, axis) - head_oracle.signed_angle(v1, v2, axis)).max() < 1e-7
    assert HeadInverseKinematics.angle_between_segments(v1[:0], v2[:0], axis).shape == (0,)
    # one key point per side (e.g. a bristle): head angles only (reference docstring :26)
    one = HeadInverseKinematics({"R_head": z["R_head"][:, :1], "L_head": z["L_head"][:, :1], "Neck": z["Neck"]},
                                NMF_TEMPLATE, log_level="ERROR")
    only = one.compute_head_angles(compute_ant_angles=False)
    assert list(only) == ANGLE_NAMES[:3] and all(np.array_equal(only[n], both[n]) for n in only)
    assert np.array_equal(one.compute_head_yaw(), both["Angle_head_yaw"])
    with pytest.raises(IndexError):
        one.compute_head_angles()
    lib = hiplib.load()
    assert lib.seqik_head_angles_ex(None, None, 0, 0, None, 0, 0.0, 0.0, 0, None, None, None) == hiplib.ERR_ARG


@pytest.mark.gpu
def test_head_kernel_streaming_rate_and_linearity(z, hiplib):
    """Config 4 at scale: 4 M frames through the device entry point; results must equal the small run
    tiled (elementwise kernel: frame t depends on frame t only)."""
    import torch
    reps = 700
    r = torch.from_numpy(np.tile(z["R_head"], (reps, 1, 1))).cuda()
    l = torch.from_numpy(np.tile(z["L_head"], (reps, 1, 1))).cuda()
    neck = torch.from_numpy(z["Neck"][0, 0].copy()).cuda()
    n = r.shape[0]
    out = torch.zeros((7, n), dtype=torch.float64, device="cuda")
    lib = hiplib.load()
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.seqik_head_angles_device(r.data_ptr(), l.data_ptr(), n, neck.data_ptr(), 0,
                                      float(z["rest_head_pitch"][0]), float(z["rest_antenna_pitch"][0]), 1,
                                      out.data_ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    small = hiplib.head_angles(z["R_head"], z["L_head"], z["Neck"][:, 0], z["rest_head_pitch"][0],
                               z["rest_antenna_pitch"][0])
    assert np.array_equal(out.cpu().numpy().reshape(7, reps, 6000), np.broadcast_to(small[:, None], (7, reps, 6000)))
    # records that are only 8-byte aligned take the per-lane loads instead of the staged 16-byte ones: same bits;
    # so does a run without the antenna angles (three rows written, the others untouched)
    m = 100_003
    r_off = torch.zeros(m * 6 + 1, dtype=torch.float64, device="cuda")
    l_off = torch.zeros(m * 6 + 1, dtype=torch.float64, device="cuda")
    r_off[1:] = r[:m].reshape(-1)
    l_off[1:] = l[:m].reshape(-1)
    out2 = torch.full((7, m), 7.0, dtype=torch.float64, device="cuda")
    assert (r_off.data_ptr() + 8) % 16 == 8
    rc = lib.seqik_head_angles_device(r_off.data_ptr() + 8, l_off.data_ptr() + 8, m, neck.data_ptr(), 0,
                                      float(z["rest_head_pitch"][0]), float(z["rest_antenna_pitch"][0]), 1,
                                      out2.data_ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out2, out[:, :m])
    out3 = torch.full((7, m), 7.0, dtype=torch.float64, device="cuda")
    rc = lib.seqik_head_angles_device(r.data_ptr(), l.data_ptr(), m, neck.data_ptr(), 0,
                                      float(z["rest_head_pitch"][0]), float(z["rest_antenna_pitch"][0]), 0,
                                      out3.data_ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out3[:3], out[:3, :m]) and bool((out3[3:] == 7.0).all())


def test_derotate_vector_broadcasts_like_the_reference():
    """the reference's own callers hand `derotate_vector` an (N,) roll with an (N, 3) vector through
    np.vectorize(signature="(m),(m,n)->(m,n)") (head_inverse_kinematics.py:253, :290), and Rotation.from_euler builds one
    rotation per row.  Scalar / (3,), scalar / (M, 3) and (M,) / (M, 3) all against scipy's Rotation (host helper: no GPU)."""
    from scipy.spatial.transform import Rotation
    from seqikpy_amd.head_inverse_kinematics import HeadInverseKinematics
    hk = object.__new__(HeadInverseKinematics)      # the helper uses no state of the object
    rng = np.random.default_rng(11)
    v = rng.normal(size=(40, 3))
    roll = rng.uniform(-3.0, 3.0, size=40)
    assert np.abs(hk.derotate_vector(0.4, v[0]) - Rotation.from_euler("x", -0.4).apply(v[0])).max() < 1e-15
    assert hk.derotate_vector(0.4, v[0]).shape == (3,)
    assert np.abs(hk.derotate_vector(0.4, v) - Rotation.from_euler("x", -0.4).apply(v)).max() < 1e-15
    got = hk.derotate_vector(roll, v)
    assert got.shape == (40, 3) and np.abs(got - Rotation.from_euler("x", -roll).apply(v)).max() < 1e-15
    vec = np.vectorize(hk.derotate_vector, signature="(m),(m,n)->(m,n)")     # exactly how the reference calls it
    assert np.array_equal(vec(roll, v), got)
    assert np.abs(hk.derotate_vector(roll, v[0]) - Rotation.from_euler("x", -roll).apply(v[0])).max() < 1e-15
    with pytest.raises(ValueError):
        hk.derotate_vector(roll[:7], v)
