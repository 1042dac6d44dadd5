"""Head and antenna joint angles -- host-side mirror of the reference's
``seqikpy/head_inverse_kinematics.py`` (``HeadInverseKinematics`` :53-339).

Same constructor, ``compute_head_angles(export_path=None, compute_ant_angles=True)`` and output
dictionary (``Angle_head_roll`` ... ``Angle_antenna_pitch_R`` -> ``(N,)``); the per-frame closed
forms run in one HIP kernel (``csrc/seqik_head.hpp``).  The two zero-pose constants of the body
template are evaluated here on the host (they are scalars)."""
import logging
from pathlib import Path
from typing import Dict, Literal, Optional, Union

import numpy as np

from . import _lib
from .utils import save_file

ANGLE_NAMES = ["Angle_head_roll", "Angle_head_pitch", "Angle_head_yaw", "Angle_antenna_yaw_L",
               "Angle_antenna_pitch_L", "Angle_antenna_yaw_R", "Angle_antenna_pitch_R"]


def _signed_angle(v1, v2, axis):
    """Angle between two 3-vectors, negative when det([axis, v1, v2]) <= 0 (reference :163-178)."""
    v1 = np.asarray(v1, dtype=np.float64)
    v2 = np.asarray(v2, dtype=np.float64)
    c = np.dot(v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2))
    sign = 1.0 if np.dot(axis, np.cross(v1, v2)) > 0 else -1.0
    return float(np.arccos(c) * sign)


class HeadInverseKinematics:
    """Calculates the head DOFs (3) and the antennae DOFs (2 per side).

    ``aligned_pos`` must hold ``R_head``, ``L_head`` ((N, 2, 3): antenna base, antenna tip) and ``Neck``
    ((1, 1, 3) or (N, 1, 3)); ``body_template`` the template joint positions (``data.NMF_TEMPLATE``)."""

    def __init__(self, aligned_pos: Dict[str, np.ndarray], body_template: Dict[str, np.ndarray],
                 log_level: Literal["DEBUG", "INFO", "WARNING", "ERROR"] = "INFO") -> None:
        self.aligned_pos = aligned_pos
        self.body_template = body_template
        if not all(key in self.aligned_pos for key in ["R_head", "L_head", "Neck"]):
            raise ValueError(
                """self.aligned_pos must have R_head, L_head, Neck as keys,
                at least one of them is missing in the current data"""
            )
        self.rest_head_pitch = self.get_rest_head_pitch()
        self.rest_antenna_pitch = self.get_rest_antenna_pitch()
        self.logger = logging.getLogger(self.__class__.__name__)
        self.logger.setLevel(getattr(logging, log_level.upper(), None))
        self.device = -1  # HIP device ordinal; -1 = the calling thread's current device

    def get_rest_antenna_pitch(self) -> float:
        """Antenna pitch at the zero pose of the biomechanical model."""
        head = np.array(self.body_template["Neck"] - self.body_template["R_Antenna_base"], dtype=np.float64)
        head[1] = 0
        ant = np.array(self.body_template["R_Antenna_edge"] - self.body_template["R_Antenna_base"], dtype=np.float64)
        ant[1] = 0
        return _signed_angle(head, ant, np.array([0.0, 1.0, 0.0]))

    def get_rest_head_pitch(self) -> float:
        """Head pitch at the zero pose of the biomechanical model."""
        head = np.array((self.body_template["R_Antenna_base"] + self.body_template["L_Antenna_base"]) * 0.5
                        - self.body_template["Neck"], dtype=np.float64)
        head[1] = 0
        return _signed_angle(head, np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))

    def compute_head_angles(self, export_path: Union[str, Path] = None,
                            compute_ant_angles: Optional[bool] = True) -> Dict[str, np.ndarray]:
        """Head roll, pitch, yaw and (optionally) antenna yaw / pitch per side, one value per frame."""
        neck = np.asarray(self.aligned_pos["Neck"], dtype=np.float64)[:, 0, :]
        out = _lib.head_angles(self.aligned_pos["R_head"], self.aligned_pos["L_head"], neck,
                               self.rest_head_pitch, self.rest_antenna_pitch, compute_ant=bool(compute_ant_angles),
                               device=self.device)
        head_angles = {name: out[i].copy() for i, name in enumerate(ANGLE_NAMES[: out.shape[0]])}
        if export_path is not None:
            save_file(Path(export_path) / "head_joint_angles.pkl", head_angles)
            self.logger.info("Head joint angles are saved at %s!", export_path)
        return head_angles
