"""ORACLE / TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's closed-form head and
antenna angles (seqikpy/head_inverse_kinematics.py:103-339), vectorised over frames.

  angle_between_segments   :163-178   acos of the normalised dot product, sign from det([axis, v1, v2])
  compute_head_roll/pitch/yaw :180-226
  compute_antenna_pitch/yaw   :228-291 (after derotate_vector :330-333 = Rx(-head_roll))
Pinned by tests/test_head.py against the shipped head_joint_angles.pkl (fixture anipose_head.npz)."""
import numpy as np

X, Y, Z = np.eye(3)


def signed_angle(v1, v2, axis):
    v1 = np.atleast_2d(v1).astype(np.float64)
    v2 = np.atleast_2d(v2).astype(np.float64)
    n = max(len(v1), len(v2))
    v1 = np.broadcast_to(v1, (n, 3))
    v2 = np.broadcast_to(v2, (n, 3))
    c = np.einsum("ij,ij->i", v1 / np.linalg.norm(v1, axis=1)[:, None], v2 / np.linalg.norm(v2, axis=1)[:, None])
    det = np.einsum("j,ij->i", axis, np.cross(v1, v2))
    return np.arccos(c) * np.where(det > 0, 1.0, -1.0)


def derotate(roll, v):
    """Rotation about X by -roll (what scipy's Rotation.from_euler('x', -roll).apply does)."""
    c, s = np.cos(roll), np.sin(roll)
    return np.stack([v[:, 0], c * v[:, 1] + s * v[:, 2], -s * v[:, 1] + c * v[:, 2]], axis=1)


def head_angles(r_head, l_head, neck, rest_head_pitch, rest_antenna_pitch, head_roll=None, compute_ant=True):
    """(N, K, 3), (N, K, 3), neck (1 or N, 3) -> (7, N) in the reference's dict order (3 rows without the antennae).
    head_roll: the roll compute_antenna_pitch / _yaw are handed (:242, :278); None = the frames' own."""
    rb, lb = r_head[:, 0], l_head[:, 0]
    hor = lb - rb
    mid = (rb + lb) * 0.5 - neck
    v = hor.copy(); v[:, 0] = 0
    roll = signed_angle(Y, v, X)
    v = mid.copy(); v[:, 1] = 0
    pitch = signed_angle(X, v, Y) + rest_head_pitch
    v = hor.copy(); v[:, 2] = 0
    yaw = signed_angle(Y, v, Z)
    out = [roll, pitch, yaw]
    if not compute_ant:
        return np.stack(out)
    if head_roll is not None:
        roll = np.broadcast_to(np.asarray(head_roll, dtype=np.float64).reshape(-1), roll.shape)
    hor_d = derotate(roll, hor)
    for side, head in (("L", l_head), ("R", r_head)):
        ant = derotate(roll, head[:, 1] - head[:, 0])
        hv = derotate(roll, neck - head[:, 0])
        a1 = ant.copy(); a1[:, 0] = 0
        h1 = hor_d.copy(); h1[:, 0] = 0
        ayaw = signed_angle(a1, h1, X)
        if side == "R":
            ayaw = np.pi - ayaw
        a2 = ant.copy(); a2[:, 1] = 0
        h2 = hv.copy(); h2[:, 1] = 0
        apitch = signed_angle(h2, a2, Y) - rest_antenna_pitch
        out += [ayaw, apitch]
    return np.stack(out)
