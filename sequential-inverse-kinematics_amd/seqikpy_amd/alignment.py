"""Leg alignment -- host-side mirror of the leg part of the reference's ``AlignPose``
(``seqikpy/alignment.py:228-487``: ``align_pose``, ``get_fixed_pos``, ``get_mean_length``,
``find_scale_leg``, ``align_leg``).

Aligning a leg is one per-frame affine map, ``(raw - fixed_coxa) * scale + template_coxa``, whose
three constants are quantile statistics of the whole recording.  The statistics are computed
here on the host (numpy, O(N log N), once per recording); the per-frame map is either applied on
the host by ``align_pose()`` -- bit-identical to the reference -- or handed to the HIP kernels via
``leg_affine()`` / ``LegInvKinSeq(..., leg_affine=...)`` so that RAW key points go straight to the
GPU and the map is fused into the solve prologue (``include/seqik.h``: ``SeqikAffine``).

Also here: the antenna alignment (``align_head``, host only -- it feeds the closed-form head kernel)
and the three input converters (anipose, DeepFly3D, DeepFly3DPostProcessing -> the segment dictionary
every class of this package consumes), reference ``seqikpy/alignment.py:103-226``.
"""
import logging
from pathlib import Path
from typing import Dict, List, Literal, Optional, Tuple, Union

import numpy as np

import pickle

from .data import NMF_TEMPLATE, PTS2ALIGN
from .utils import calculate_body_size, dict_to_nparray_pose, save_file


def convert_from_anipose_to_dict(pose_3d: Dict[str, np.ndarray], pts2align: Dict[str, List[str]]) -> Dict[str, np.ndarray]:
    """anipose columns ``"<keypoint>_x|_y|_z" -> (N,)`` to ``{segment: (N, n_key_points, 3)}``."""
    out = {}
    for segment, names in pts2align.items():
        arr = np.empty((np.asarray(pose_3d[f"{names[0]}_x"]).shape[0], len(names), 3))
        for i, kp in enumerate(names):
            for a, axis in enumerate("xyz"):
                arr[:, i, a] = pose_3d[f"{kp}_{axis}"]
        out[segment] = arr
    return out


def convert_from_df3d_to_dict(pose_3d: np.ndarray, pts2align: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """DeepFly3D ``(N, n_key_points, 3)`` array + ``{segment: key point indices}``."""
    return {segment: pose_3d[:, idx, :].copy() for segment, idx in pts2align.items()}


def convert_from_df3dpp_to_dict(pose_3d: Dict[str, Dict[str, np.ndarray]],
                                pts2align: Optional[List[str]] = None) -> Dict[str, np.ndarray]:
    """DeepFly3DPostProcessing nested dictionary -> ``{"<leg>_leg": (N, 5, 3)}``."""
    segments = list(pose_3d.keys()) if pts2align is None else pts2align
    return {segment: dict_to_nparray_pose(pose_3d[segment], claw_is_end_effector=True) for segment in segments}


def _mean_quantile(vector: np.ndarray, quantile_diff: float = 0.05) -> float:
    """Mean of the 0.45 and 0.55 quantiles (reference ``_get_mean_quantile``, alignment.py:83-87)."""
    return 0.5 * (np.quantile(vector, q=0.5 - quantile_diff) + np.quantile(vector, q=0.5 + quantile_diff))


def linear_quantile_index(n: int, q: float) -> Tuple[int, int, float]:
    """``(previous index, next index, gamma)`` of numpy's default ("linear") quantile of ``n`` values, with the
    floating-point expression of the numpy that is installed: its own method table when it can be reached
    (``_QuantileMethods["linear"]``; ``(n - 1) q`` in numpy 1.22-2.x, which is NOT bit-identical to the general
    ``n q + (alpha + q (1 - alpha - beta)) - 1`` form other methods use), the same expression otherwise; neighbours
    and gamma as ``_get_indexes`` / ``_get_gamma``.  tests/test_alignment.py checks it against ``np.quantile`` itself
    for every n up to 3000."""
    vi = None
    try:
        from numpy.lib import _function_base_impl as _impl
        vi = float(_impl._QuantileMethods["linear"]["get_virtual_index"](n, q))
    except Exception:  # other numpy layout: the documented expression of the "linear" method
        vi = float((n - 1) * q)
    lo = int(np.floor(vi))
    hi = lo + 1
    if vi >= n - 1:       # numpy: indexes above the bounds -> both neighbours are the last element
        lo = hi = n - 1
    if vi < 0:
        lo = hi = 0
    return lo, hi, float(vi - np.floor(vi))


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

    @classmethod
    def from_file_path(cls, main_dir: Union[str, Path], file_name: Optional[str] = "pose3d.*",
                       convert_func=None, pts2align: Optional[Dict[str, List[str]]] = None, **kwargs):
        """Loads a pickled 3D pose (the newest match of ``file_name`` under ``main_dir``), optionally
        converts it with ``convert_func(pose, pts2align)``; ``FileNotFoundError`` if nothing matches."""
        paths = list(Path(main_dir).rglob(file_name))
        if not paths:
            raise FileNotFoundError(f"{file_name} does not exits in {main_dir}")
        with open(paths[-1].as_posix(), "rb") as f:
            pose_3d = pickle.load(f)
        if convert_func is not None:
            return cls(convert_func(pose_3d, PTS2ALIGN if pts2align is None else pts2align), **kwargs)
        return cls(pose_3d, **kwargs)

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

    def leg_affines(self, on_gpu: bool = False, device: int = -1) -> Dict[str, Tuple[np.ndarray, float, np.ndarray]]:
        """Affine constants of every ``*_leg`` entry, keyed by leg name (for the fused GPU path).

        ``on_gpu=True`` computes the whole-recording reductions (seven quantile pairs per leg) on the MI355X
        (``seqik_align_stats_*``): the GPU returns the exact order statistics, numpy's own interpolation / mean /
        scale formulas are applied here, so the constants equal the host path's bit for bit -- at a fraction of
        its cost for long recordings (host: ~5 s per million frames and six legs)."""
        segs = [(seg, arr) for seg, arr in self.pose_data_dict.items() if "leg" in seg]
        if not on_gpu:
            return {seg[:2]: self.leg_affine(arr, seg[:2]) for seg, arr in segs}
        from . import _lib
        n = {np.asarray(arr).shape[0] for _, arr in segs}
        if len(n) != 1:
            raise ValueError("on_gpu=True needs the same number of frames for every leg")
        n = n.pop()
        pose = np.stack([np.asarray(arr, dtype=np.float64)[:, :5, :] for _, arr in segs])[None]   # (1, L, N, 5, 3)
        if not np.isfinite(pose).all():
            # numpy's quantile of a series that holds a NaN is NaN; a radix sort would just push the NaN to the end
            return {seg[:2]: self.leg_affine(arr, seg[:2]) for seg, arr in segs}
        qs = (0.5 - 0.05, 0.5 + 0.05)
        ranks, gammas = [], []
        for q in qs:
            lo, hi, gamma = linear_quantile_index(n, q)
            ranks += [lo, hi]
            gammas.append(gamma)
        with _lib.AlignStats(len(segs), n, device=device) as st:
            st.add(pose)
            order = st.finish(ranks)                                                                  # (L, 7, 4)

        def mean_quantile(v4):
            vals = []
            for j, g in enumerate(gammas):  # numpy.lib._function_base_impl._lerp
                a, b = v4[2 * j], v4[2 * j + 1]
                diff = b - a
                r = a + diff * g
                if g >= 0.5:
                    r = b - diff * (1 - g)
                vals.append(r)
            return 0.5 * (vals[0] + vals[1])

        out = {}
        for li, (seg, _) in enumerate(segs):
            leg = seg[:2]
            fixed = np.array([mean_quantile(order[li, a]) for a in range(3)])
            mean_length = {name: mean_quantile(order[li, 3 + i]) for i, name in enumerate(["coxa", "femur", "tibia", "tarsus"])}
            scale = self.find_scale_leg(leg, mean_length)
            self.logger.info("Scale factor for %s leg: %s", leg, scale)
            out[leg] = (fixed, float(scale), np.asarray(self.body_template[f"{leg}_Coxa"], dtype=np.float64))
        return out

    # -- host application (reference-identical) ------------------------------------------
    def align_leg(self, leg_array: np.ndarray, leg_name: str) -> np.ndarray:
        fixed_coxa, scale, template_coxa = self.leg_affine(leg_array, leg_name)
        aligned = np.empty_like(leg_array)
        aligned[:, 0, :] = np.zeros_like(leg_array[:, 0, :]) + template_coxa
        for i in range(1, 5):
            aligned[:, i, :] = (leg_array[:, i, :] - fixed_coxa).reshape(-1, 3) * scale + template_coxa
        return aligned

    # -- antenna alignment (host) -----------------------------------------------------------
    @property
    def thorax_mid_pts(self) -> np.ndarray:
        assert "Thorax" in self.pose_data_dict, "To align the head, you need to have a `Thorax` key point"
        thorax = self.pose_data_dict["Thorax"]
        return 0.5 * (thorax[:, 0, :] + thorax[:, -1, :])

    def find_stationary_indices(self, array: np.ndarray, threshold: Optional[float] = 5e-5) -> np.ndarray:
        """Frames where the second difference of ``array`` is below ``threshold``."""
        return np.where(np.diff(np.diff(array)) < threshold)[0]

    def align_head(self, head_array: np.ndarray, side: str) -> np.ndarray:
        """Scales / translates antenna base and tip to the template (reference :489-555): the base by the
        antenna-base-to-thorax distance, the tip by the antenna length, both about a fixed antenna origin
        estimated on the frames where that distance is stationary."""
        base_to_thorax = np.linalg.norm(head_array[:, 0, :] - self.thorax_mid_pts, axis=1)
        ant_len = np.linalg.norm(np.diff(head_array, axis=1), axis=2)[:, 0]
        if not (self.body_size.get("Antenna_mid_thorax") and self.body_size.get("Antenna")):
            raise KeyError("body_size must hold <Antenna_mid_thorax> and <Antenna>")
        stat = self.find_stationary_indices(base_to_thorax)
        origin = AlignPose.get_fixed_pos(head_array[stat, 0, :])
        scale_base = self.body_size["Antenna_mid_thorax"] / _mean_quantile(base_to_thorax[stat])
        scale_tip = self.body_size["Antenna"] / _mean_quantile(ant_len)
        self.logger.info("Scale factor antenna base %s: %s, ant itself: %s", side, scale_base, scale_tip)
        aligned = np.empty_like(head_array)
        tmpl = self.body_template[f"{side}_Antenna_base"]
        aligned[:, 0, :] = (head_array[:, 0, :] - origin) * scale_base + tmpl
        aligned[:, 1, :] = (head_array[:, 1, :] - origin) * scale_tip + tmpl
        return aligned

    def align_pose(self, export_path: Optional[Union[str, Path]] = None) -> Dict[str, np.ndarray]:
        aligned_pose = {}
        for segment, segment_array in self.pose_data_dict.items():
            if "leg" in segment:
                aligned_pose[segment] = self.align_leg(segment_array, segment[:2])
            elif "head" in segment:
                aligned_pose[segment] = self.align_head(segment_array, segment[0])
            else:
                self.logger.debug("%s is not aligned", segment)
        if "Neck" in self.body_template:
            aligned_pose["Neck"] = self.body_template["Neck"].reshape((-1, 1, 3))
        if export_path is not None:
            save_file(Path(export_path) / "pose3d_aligned.pkl", aligned_pose)
            self.logger.info("Aligned pose is saved at %s", export_path)
        return aligned_pose
