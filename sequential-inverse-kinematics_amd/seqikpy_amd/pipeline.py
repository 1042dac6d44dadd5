"""Legs + head / antennae of one recording in ONE submission (BASELINE config 4).

The reference runs ``HeadInverseKinematics.compute_head_angles()`` and ``LegInvKinSeq.run_ik_and_fk()`` one after
the other and merges the two dictionaries into ``body_joint_angles.pkl``
(``examples/example_entire_pipeline.py:72-101``).  Here both go to the GPU together: the key points are uploaded
once, the leg kernel and the head / antenna kernel are enqueued on two HIP streams (device entry points of the C
ABI), one synchronisation, one download.  Results are the bits the two classes return separately.

torch is used for device memory and streams only.
"""
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from .data import DOFS, INITIAL_ANGLES
from .head_inverse_kinematics import ANGLE_NAMES, HeadInverseKinematics
from .kinematic_chain import KinematicChainSeq


def run_body_ik(aligned_pos: Dict[str, np.ndarray], kinematic_chain_class: KinematicChainSeq,
                body_template: Dict[str, np.ndarray], initial_angles: Optional[Dict] = None, device: int = -1,
                frame_parallel=None, stats: Optional[dict] = None) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Returns ``(body_joint_angles, forward_kinematics)``: the 7 head / antenna angles (when ``R_head``, ``L_head``
    and ``Neck`` are present) + 7 angles per leg, and the ``"<leg>_leg" -> (N, 9, 3)`` joint positions.
    ``frame_parallel``: as ``LegInvKinSeq.run_ik_and_fk`` -- None / ``"auto"`` (default: verified frame chunks; the same
    device-side per-leg guard as every other entry point) or ``False`` (the reference's serial walk).  ``stats``: a dict
    that receives the chunk statistics of the launch."""
    import torch
    from .leg_inverse_kinematics import default_frame_parallel
    if frame_parallel is None:
        frame_parallel = default_frame_parallel()
    if initial_angles is None:
        initial_angles = INITIAL_ANGLES
    kc = kinematic_chain_class
    segs = [(name, name.split("_")[0]) for name in aligned_pos
            if "leg" in name.lower() and f"{name.split('_')[0]}_Coxa" in kc.body_size]
    if not segs:
        raise ValueError("no leg of aligned_pos is covered by the kinematic chain's body_size")
    legs = [_lib.make_leg_params(leg, kc.bounds_dof, kc.body_size, initial_angles) for _, leg in segs]
    pose = np.stack([np.asarray(aligned_pos[name], dtype=np.float64)[:, :5, :] for name, _ in segs])[None]
    _lib._check_finite(pose)
    n = pose.shape[2]
    with_head = all(k in aligned_pos for k in ("R_head", "L_head", "Neck"))
    lib = _lib.load()
    with torch.cuda.device(device):
        leg_stream, head_stream = torch.cuda.Stream(), torch.cuda.Stream()
        d_pose = torch.from_numpy(pose).cuda(non_blocking=True)
        d_ang = torch.zeros((1, len(segs), n, 7), dtype=torch.float64, device="cuda")
        d_fk = torch.zeros((1, len(segs), n, 9, 3), dtype=torch.float64, device="cuda")
        # (allocated and zero-filled on the current stream BEFORE leg_stream waits for it: the kernels of the chunked call
        # write these statistics on leg_stream)
        d_stats = torch.zeros(_lib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
        cur = torch.cuda.current_stream()
        leg_stream.wait_stream(cur)
        _lib.solve_seq_device(d_pose.data_ptr(), 1, len(segs), n, legs, d_ang.data_ptr(), d_fk.data_ptr(),
                              stream=leg_stream.cuda_stream, frame_chunk=-1 if frame_parallel else 0,
                              d_chunk_stats=d_stats.data_ptr())
        if with_head:
            hk = HeadInverseKinematics(aligned_pos, body_template, log_level="ERROR")
            r = np.ascontiguousarray(aligned_pos["R_head"], dtype=np.float64)
            l_ = np.ascontiguousarray(aligned_pos["L_head"], dtype=np.float64)
            neck = np.ascontiguousarray(np.asarray(aligned_pos["Neck"], dtype=np.float64)[:, 0, :])
            nh = r.shape[0]
            d_r, d_l, d_n = (torch.from_numpy(a).cuda(non_blocking=True) for a in (r, l_, neck))
            d_head = torch.zeros((7, nh), dtype=torch.float64, device="cuda")
            head_stream.wait_stream(cur)
            rc = lib.seqik_head_angles_device(d_r.data_ptr(), d_l.data_ptr(), nh, d_n.data_ptr(),
                                              3 if (neck.shape[0] == nh and nh > 1) else 0, hk.rest_head_pitch,
                                              hk.rest_antenna_pitch, 1, d_head.data_ptr(), head_stream.cuda_stream)
            if rc != _lib.SEQIK_OK:
                _lib._raise(rc)
        leg_stream.synchronize()
        head_stream.synchronize()
        _lib.check_faults()   # the device entry points do not synchronise: a kernel fault is reported here
        ang, fk = d_ang.cpu().numpy(), d_fk.cpu().numpy()
        head = d_head.cpu().numpy() if with_head else None
        if stats is not None:
            stats.update(_lib.chunk_stats_dict(d_stats.cpu().numpy()))
    body = {}
    if with_head:
        body.update({name: head[i].copy() for i, name in enumerate(ANGLE_NAMES)})
    fk_dict = {}
    for li, (name, leg) in enumerate(segs):
        for d, dof in enumerate(DOFS):
            body[f"Angle_{leg}_{dof}"] = ang[0, li, :, d].copy()
        fk_dict[name] = fk[0, li].copy()
    return body, fk_dict
