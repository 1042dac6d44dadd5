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
