"""Leg alignment -- host-side mirror of the leg part of the reference's ``AlignPose``
(``seqikpy/alignment.py:228-487``: ``align_pose``, ``get_fixed_pos``, ``get_mean_length``,
``find_scale_leg``, ``align_leg``).

Aligning a leg is one per-frame affine map, ``(raw - fixed_coxa) * scale + template_coxa``, whose
three constants are quantile statistics of the whole recording.  The statistics are computed
here on the host (numpy, O(N log N), once per recording); the per-frame map is either applied on
the host by ``align_pose()`` -- bit-identical to the reference -- or handed to the HIP kernels via
``leg_affine()`` / ``LegInvKinSeq(..., leg_affine=...)`` so that RAW key points go straight to the
GPU and the map is fused into the solve prologue (``include/seqik.h``: ``SeqikAffine``).

Head / antenna alignment (``align_head``) and the anipose / df3d converters are not part of this
round.
"""
import logging
from pathlib import Path
from typing import Dict, List, Literal, Optional, Tuple, Union

import numpy as np

from .data import NMF_TEMPLATE
from .utils import calculate_body_size, save_file


def _mean_quantile(vector: np.ndarray, quantile_diff: float = 0.05) -> float:
    """Mean of the 0.45 and 0.55 quantiles (reference ``_get_mean_quantile``, alignment.py:83-87)."""
    return 0.5 * (np.quantile(vector, q=0.5 - quantile_diff) + np.quantile(vector, q=0.5 + quantile_diff))


class AlignPose:
    """Aligns the 3D leg key points to the template (coxa position and leg size).

    Parameters mirror the reference: ``pose_data_dict`` (``"<leg>_leg" -> (N, 5, 3)``), ``legs_list``,
    ``include_claw``, ``body_template``, ``body_size``, ``log_level``.
    """

    def __init__(self, pose_data_dict: Dict[str, np.ndarray], legs_list: List[str],
                 include_claw: Optional[bool] = False, body_template: Optional[Dict[str, np.ndarray]] = None,
                 body_size: Optional[Dict[str, float]] = None,
                 log_level: Literal["DEBUG", "INFO", "WARNING", "ERROR"] = "INFO") -> None:
        self.pose_data_dict = pose_data_dict
        self.include_claw = include_claw
        self.body_template = NMF_TEMPLATE if body_template is None else body_template
        self.body_size = calculate_body_size(self.body_template, legs_list) if body_size is None else body_size
        self.logger = logging.getLogger(self.__class__.__name__)
        self.logger.setLevel(getattr(logging, log_level.upper(), None))

    # -- statistics (host) ---------------------------------------------------------------
    @staticmethod
    def get_fixed_pos(points_3d: np.ndarray) -> np.ndarray:
        """Per-axis mean of the 0.45 / 0.55 quantiles of a key point over the recording."""
        return np.array([_mean_quantile(points_3d[:, 0]), _mean_quantile(points_3d[:, 1]),
                         _mean_quantile(points_3d[:, 2])])

    def get_mean_length(self, segment_array: np.ndarray, segment_is_leg: bool = True) -> Dict[str, float]:
        lengths = np.linalg.norm(np.diff(segment_array, axis=1), axis=2)
        names = ["coxa", "femur", "tibia", "tarsus"] if segment_is_leg else ["antenna"]
        return {name: _mean_quantile(lengths[:, i]) for i, name in enumerate(names)}

    def find_scale_leg(self, leg_name: str, mean_length: Dict[str, float]) -> float:
        model = self.body_size[leg_name] if self.include_claw else (
            self.body_size[leg_name] - self.body_size[f"{leg_name}_Tarsus"])
        fly = mean_length["coxa"] + mean_length["femur"] + mean_length["tibia"]
        fly += mean_length["tarsus"] if self.include_claw else 0
        return model / fly

    def leg_affine(self, leg_array: np.ndarray, leg_name: str) -> Tuple[np.ndarray, float, np.ndarray]:
        """``(fixed_coxa, scale, template_coxa)`` of one leg -- the constants of ``align_leg``."""
        fixed_coxa = AlignPose.get_fixed_pos(leg_array[:, 0, :])
        scale = self.find_scale_leg(leg_name, self.get_mean_length(leg_array, segment_is_leg=True))
        self.logger.info("Scale factor for %s leg: %s", leg_name, scale)
        return fixed_coxa, float(scale), np.asarray(self.body_template[f"{leg_name}_Coxa"], dtype=np.float64)

    def leg_affines(self) -> Dict[str, Tuple[np.ndarray, float, np.ndarray]]:
        """Affine constants of every ``*_leg`` entry, keyed by leg name (for the fused GPU path)."""
        return {seg[:2]: self.leg_affine(arr, seg[:2]) for seg, arr in self.pose_data_dict.items() if "leg" in seg}

    # -- host application (reference-identical) ------------------------------------------
    def align_leg(self, leg_array: np.ndarray, leg_name: str) -> np.ndarray:
        fixed_coxa, scale, template_coxa = self.leg_affine(leg_array, leg_name)
        aligned = np.empty_like(leg_array)
        aligned[:, 0, :] = np.zeros_like(leg_array[:, 0, :]) + template_coxa
        for i in range(1, 5):
            aligned[:, i, :] = (leg_array[:, i, :] - fixed_coxa).reshape(-1, 3) * scale + template_coxa
        return aligned

    def align_pose(self, export_path: Optional[Union[str, Path]] = None) -> Dict[str, np.ndarray]:
        aligned_pose = {}
        for segment, segment_array in self.pose_data_dict.items():
            if "leg" in segment:
                aligned_pose[segment] = self.align_leg(segment_array, segment[:2])
            elif "head" in segment:
                raise NotImplementedError("head alignment is not part of this build yet (SURVEY.md 8f-2)")
            else:
                self.logger.debug("%s is not aligned", segment)
        if "Neck" in self.body_template:
            aligned_pose["Neck"] = self.body_template["Neck"].reshape((-1, 1, 3))
        if export_path is not None:
            save_file(Path(export_path) / "pose3d_aligned.pkl", aligned_pose)
            self.logger.info("Aligned pose is saved at %s", export_path)
        return aligned_pose
