"""Synthetic in-workspace key points for benchmarks and property tests.

Per (sequence, frame, leg) joint angles are drawn inside the joint bounds, pushed through the
stage-4 forward kinematics of the leg (the chain of ``KinematicChainSeq.create_leg_chain_stage_4``,
reference ``seqikpy/kinematic_chain.py:338-421``) to get the five key points, and perturbed with
Gaussian key-point noise.  Two variants:

* ``"iid"``    -- every frame independent (no temporal continuity; the warm start of a frame is the
  answer of an unrelated previous frame);
* ``"smooth"`` -- angles follow a band-limited random walk inside the bounds (realistic).

This is data generation only (numpy on the host); it is not part of the solve path.
"""
from typing import Dict, List, Sequence

import numpy as np

from .data import DOFS, SEGMENTS

SEED_BASE = 20241022


def _rot(axis: str, a: np.ndarray) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    o, z = np.ones_like(a), np.zeros_like(a)
    if axis == "x":
        m = [[o, z, z], [z, c, -s], [z, s, c]]
    elif axis == "y":
        m = [[c, z, s], [z, o, z], [-s, z, c]]
    else:
        m = [[c, -s, z], [s, c, z], [z, z, o]]
    return np.stack([np.stack(r, axis=-1) for r in m], axis=-2)


def leg_forward_kinematics(theta: np.ndarray, seg: Sequence[float]) -> np.ndarray:
    """Key points (..., 5, 3) relative to the Thorax-Coxa joint for angles (..., 7) in ``DOFS`` order."""
    yaw, pitch, roll, ctr_pitch, ctr_roll, fti, tita = [theta[..., i] for i in range(7)]
    down = np.array([0.0, 0.0, -1.0])
    m = _rot("x", yaw) @ _rot("y", pitch) @ _rot("z", roll) @ _rot("y", ctr_pitch)
    p0 = np.zeros(theta.shape[:-1] + (3,))
    # the coxa hangs below the three ThC rotations; CTr_pitch rotates after the translation
    m_thc = _rot("x", yaw) @ _rot("y", pitch) @ _rot("z", roll)
    p1 = p0 + (m_thc @ down) * seg[0]
    m2 = m @ _rot("z", ctr_roll)
    p2 = p1 + (m2 @ down) * seg[1]
    m3 = m2 @ _rot("y", fti)
    p3 = p2 + (m3 @ down) * seg[2]
    m4 = m3 @ _rot("y", tita)
    p4 = p3 + (m4 @ down) * seg[3]
    return np.stack([p0, p1, p2, p3, p4], axis=-2)


def _smooth_walk(rng, shape_sn, n_dof, smoothness=25):
    """Band-limited noise in [0, 1]: moving average of white noise along the frame axis, rescaled."""
    s, n = shape_sn
    w = rng.standard_normal((s, n + smoothness, n_dof))
    kernel = np.hanning(smoothness + 1)
    kernel /= kernel.sum()
    out = np.empty((s, n, n_dof))
    for d in range(n_dof):
        for i in range(s):
            out[i, :, d] = np.convolve(w[i, :, d], kernel, mode="valid")[:n]
    out /= (np.sqrt((kernel ** 2).sum()) * 3.0)  # ~unit variance -> +-1 at 3 sigma
    return np.clip(0.5 + 0.5 * out, 0.0, 1.0)


def synthetic_pose(n_seq: int, n_frames: int, legs: List[str], bounds_dof: Dict[str, tuple],
                   body_size: Dict[str, float], template: Dict[str, np.ndarray], variant: str = "iid",
                   noise: float = 0.01, seed: int = SEED_BASE, margin: float = 0.05,
                   return_theta: bool = False):
    """Key points ``(n_seq, n_legs, n_frames, 5, 3)`` (float64) for the given legs.

    ``rng = default_rng(seed + leg_index)`` per leg; origin = template ``<leg>_Coxa``."""
    if variant not in ("iid", "smooth"):
        raise ValueError("variant must be 'iid' or 'smooth'")
    pose = np.empty((n_seq, len(legs), n_frames, 5, 3))
    thetas = np.empty((n_seq, len(legs), n_frames, 7))
    for li, leg in enumerate(legs):
        rng = np.random.default_rng(seed + li)
        lb = np.array([bounds_dof[f"{leg}_{d}"][0] for d in DOFS])
        ub = np.array([bounds_dof[f"{leg}_{d}"][1] for d in DOFS])
        lo = lb + margin * (ub - lb)
        hi = ub - margin * (ub - lb)
        if variant == "iid":
            u = rng.random((n_seq, n_frames, 7))
        else:
            u = _smooth_walk(rng, (n_seq, n_frames), 7)
        theta = lo + u * (hi - lo)
        seg = [body_size[f"{leg}_{s}"] for s in SEGMENTS]
        kp = leg_forward_kinematics(theta, seg)
        kp = kp + template[f"{leg}_Coxa"]
        kp[..., 1:, :] += noise * rng.standard_normal(kp[..., 1:, :].shape)
        pose[:, li] = kp
        thetas[:, li] = theta
    if return_theta:
        return pose, thetas
    return pose
