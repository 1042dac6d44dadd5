"""Head and antenna joint angles -- host-side mirror of the reference's
``seqikpy/head_inverse_kinematics.py`` (``HeadInverseKinematics`` :53-339).

Same constructor, ``compute_head_angles(export_path=None, compute_ant_angles=True)`` and output
dictionary (``Angle_head_roll`` ... ``Angle_antenna_pitch_R`` -> ``(N,)``); the per-frame closed
forms run in one HIP kernel (``csrc/seqik_head.hpp``).  The two zero-pose constants of the body
template are evaluated here on the host (they are scalars).

The reference's per-quantity methods are here too, each one a launch of the same kernel:
``compute_head_roll / _pitch / _yaw`` (:185-240), ``compute_antenna_pitch / _yaw(side, head_roll)`` (:242-307, the
``head_roll`` argument is honoured: the kernel derotates by the roll it is given), ``angle_between_segments``
(:163-182, general vectors and axis, its own small kernel), and the array helpers ``get_head_vector*``,
``get_ant_vector``, ``get_plane``, ``derotate_vector``, ``Axes`` (:38-44, :144-161, :330-339), which only
re-arrange inputs.  ``compute_head_angles`` does not go through them: one launch gives all seven rows."""
import logging
from collections import namedtuple
from pathlib import Path
from typing import Dict, Literal, Optional, Union

import numpy as np

from . import _lib
from .utils import save_file

AxesTuple = namedtuple("AxesTuple", "X_AXIS Y_AXIS Z_AXIS")
Axes = AxesTuple(X_AXIS=np.array([1, 0, 0]), Y_AXIS=np.array([0, 1, 0]), Z_AXIS=np.array([0, 0, 1]))

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
        self.head_vector_mid = self.get_head_vector_mid()
        self.head_vector_horizontal = self.get_head_vector_horizontal()
        assert self.head_vector_mid.shape[1] == 3 and self.head_vector_horizontal.shape[1] == 3, f"""
                One of head vectors
                (mid: {self.head_vector_mid.shape}, horizontal: {self.head_vector_horizontal.shape})
                does not have the right shape (N,3).
                """
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

    # ---- input re-arrangements (:144-161, :337-339): plain array views / differences, nothing is solved here ----------
    def get_head_vector(self, side: Literal["R", "L"]) -> np.ndarray:
        """(N, 3) vector from the <side> antenna base (or any head key point) to the neck."""
        return self.aligned_pos["Neck"][:, 0, :] - self.aligned_pos[f"{side}_head"][:, 0, :]

    def get_head_vector_mid(self) -> np.ndarray:
        """(N, 3) vector from the neck to the middle of the two antenna bases."""
        return (self.aligned_pos["R_head"][:, 0, :] + self.aligned_pos["L_head"][:, 0, :]) * 0.5 \
            - self.aligned_pos["Neck"][:, 0, :]

    def get_head_vector_horizontal(self) -> np.ndarray:
        """(N, 3) vector from the right antenna base to the left one."""
        return self.aligned_pos["L_head"][:, 0, :] - self.aligned_pos["R_head"][:, 0, :]

    def get_ant_vector(self, side: Literal["R", "L"]) -> np.ndarray:
        """(N, 3) vector from the antenna base to the antenna edge."""
        return self.aligned_pos[f"{side}_head"][:, 1, :] - self.aligned_pos[f"{side}_head"][:, 0, :]

    def get_plane(self, row: np.ndarray, n_row: int) -> np.ndarray:
        """``row`` repeated ``n_row`` times."""
        return np.tile(row, (n_row, 1))

    def derotate_vector(self, head_roll_angle, vector_to_derotate: np.ndarray) -> np.ndarray:
        """``vector_to_derotate`` ((3,) or (M, 3)) rotated about the x axis by ``-head_roll_angle`` (:330-335; the
        kernel does this in registers for the antenna angles -- this is the stand-alone helper).  The angle may be a
        scalar (one rotation for every row) or an (M,) array (one rotation per row), as ``Rotation.from_euler`` /
        ``Rotation.apply`` broadcast in the reference, whose own callers pass ``(N,)`` angles with ``(N, 3)`` vectors
        (:253, :290)."""
        a = np.asarray(head_roll_angle, dtype=np.float64)
        v = np.asarray(vector_to_derotate, dtype=np.float64)
        if a.ndim > 1 or v.ndim > 2 or v.shape[-1] != 3:
            raise ValueError(f"derotate_vector: angle {a.shape} / vector {v.shape}: expected () or (M,) and (3,) or (M, 3)")
        if a.ndim == 1 and v.ndim == 2 and a.shape[0] not in (1, v.shape[0]):
            raise ValueError(f"derotate_vector: {a.shape[0]} angles for {v.shape[0]} vectors")
        c, s_ = np.cos(a), np.sin(a)
        out = np.stack(np.broadcast_arrays(v[..., 0] + 0.0 * c, c * v[..., 1] + s_ * v[..., 2],
                                           -s_ * v[..., 1] + c * v[..., 2]), axis=-1)
        return out

    # ---- per-quantity methods (:163-307), each one launch of the head kernel --------------------------------------------
    @staticmethod
    def angle_between_segments(v1: np.ndarray, v2: np.ndarray, rot_axis: np.ndarray) -> np.ndarray:
        """Angle (rad) between the rows of ``v1`` and ``v2`` from the cosine formula, negated where
        det([rot_axis, v1, v2]) is not positive."""
        return _lib.signed_angles(v1, v2, rot_axis)

    def _rows(self, compute_ant: bool, head_roll=None) -> np.ndarray:
        neck = np.asarray(self.aligned_pos["Neck"], dtype=np.float64)[:, 0, :]
        return _lib.head_angles(self.aligned_pos["R_head"], self.aligned_pos["L_head"], neck, self.rest_head_pitch,
                                self.rest_antenna_pitch, compute_ant=compute_ant, device=self.device, head_roll=head_roll)

    def compute_head_pitch(self) -> np.ndarray:
        """Head pitch (rad): antero-posterior axis to the mid head vector on the sagittal plane, plus the rest pitch.
        Higher = head lowered more."""
        return self._rows(False)[1].copy()

    def compute_head_roll(self) -> np.ndarray:
        """Head roll (rad): horizontal axis to the horizontal head vector on the transverse plane.  Positive = to the
        right in fly coordinates."""
        return self._rows(False)[0].copy()

    def compute_head_yaw(self) -> np.ndarray:
        """Head yaw (rad): horizontal axis to the horizontal head vector on the frontal plane.  Positive = to the left."""
        return self._rows(False)[2].copy()

    @staticmethod
    def _side(side: str) -> str:
        side = side.upper()
        if side not in {"R", "L"}:
            raise ValueError("Side should be either R or L")
        return side

    def compute_antenna_pitch(self, side: Literal["R", "L"], head_roll: np.ndarray) -> np.ndarray:
        """Antenna pitch (rad) of one side after derotating by ``head_roll``, minus the rest antenna pitch.
        Higher = antenna lifted more."""
        side = self._side(side)
        return self._rows(True, head_roll)[ANGLE_NAMES.index(f"Angle_antenna_pitch_{side}")].copy()

    def compute_antenna_yaw(self, side: Literal["R", "L"], head_roll: np.ndarray) -> np.ndarray:
        """Antenna yaw (rad) of one side after derotating by ``head_roll``.  Higher = closer to the midline."""
        side = self._side(side)
        return self._rows(True, head_roll)[ANGLE_NAMES.index(f"Angle_antenna_yaw_{side}")].copy()

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
