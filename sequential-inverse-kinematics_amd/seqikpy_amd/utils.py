"""Host helpers on either side of the leg-IK path: body sizes, the pickle formats, and the converters between the
pose / angle containers the reference's callers hold (DeepFly3D dictionaries) and the ``(N, key points, 3)`` /
``(N, dofs)`` arrays the solvers take and give, and the resampling of the resulting joint-angle series -- the
data-format part of the reference's ``seqikpy/utils.py`` (:89-123, :235-245, :293-362).  Not here: the stimulus /
video / DeepLabCut helpers of that file (not part of the path), its duplicates of AlignPose's private reductions, and
``from_anipose_to_array`` / ``df_to_nparray`` (:248-290), which assign a ``(3, N)`` block into an ``(N, 3)`` slot and so
only run for N == 3 -- nothing in the reference calls them."""
import pickle
from typing import Dict, List

import numpy as np

_ALL_LEGS = ["RF", "LF", "RM", "LM", "RH", "LH"]


def calculate_body_size(body_template: Dict[str, np.ndarray], legs_list: List[str] = None) -> Dict[str, float]:
    """Segment lengths from template joint positions (reference: ``seqikpy/utils.py:89-123``).

    ``body_size["<leg>_<segment>"]`` is the distance between consecutive template joints,
    ``body_size["<leg>"]`` the sum over the four segments.
    """
    legs_list = list(_ALL_LEGS) if legs_list is None else legs_list
    if set(legs_list).difference(_ALL_LEGS):
        raise NameError(
            f"""
            legs_list could only contain ["RF", "LF", "RM", "LM", "RH", "LH"],
            currently, it contains {legs_list}
            """
        )
    joints = ["Coxa", "Femur", "Tibia", "Tarsus", "Claw"]
    body_size = {}
    for i, segment in enumerate(joints[:-1]):
        for leg in legs_list:
            body_size[f"{leg}_{segment}"] = np.linalg.norm(
                body_template[f"{leg}_{segment}"] - body_template[f"{leg}_{joints[i + 1]}"]
            )
    for leg in legs_list:
        body_size[leg] = (body_size[f"{leg}_Coxa"] + body_size[f"{leg}_Femur"]
                          + body_size[f"{leg}_Tibia"] + body_size[f"{leg}_Tarsus"])
    if "R_Antenna_base" in body_template:
        body_size["Antenna"] = np.linalg.norm(body_template["R_Antenna_base"] - body_template["R_Antenna_edge"])
        body_size["Antenna_mid_thorax"] = np.linalg.norm(body_template["R_Antenna_base"] - body_template["Thorax_mid"])
    return body_size


def save_file(out_fname, data):
    """Pickle ``data`` (same on-disk format as the reference, ``seqikpy/utils.py:235-238``)."""
    with open(out_fname, "wb") as f:
        pickle.dump(data, f)


def load_file(output_fname):
    with open(output_fname, "rb") as f:
        return pickle.load(f)


def dict_to_nparray_pose(pose_dict, claw_is_end_effector: bool):
    """DeepFly3DPostProcessing leg dictionary (``{"Coxa": {"raw_pos_aligned": (N, 3)}, ...}``) ->
    ``(N, 4 or 5, 3)`` key-point array (reference ``seqikpy/utils.py:293-310``)."""
    key_points = ["Coxa", "Femur", "Tibia", "Tarsus"] + (["Claw"] if claw_is_end_effector else [])
    n = np.asarray(pose_dict["Coxa"]["raw_pos_aligned"]).shape[0]
    out = np.empty((n, len(key_points), 3))
    for i, kp in enumerate(key_points):
        out[:, i, :] = np.array(pose_dict[kp]["raw_pos_aligned"])
    return out


def dict_to_nparray_angle(angle_dict, leg, claw_is_end_effector):
    """DeepFly3DPostProcessing angle dictionary (``{"RF_leg": {"ThC_roll": (N,), ...}}``) -> ``(N, 7 or 6)`` in THAT
    format's column order -- roll, yaw, pitch, then CTr pitch / roll, FTi, (TiTa) (``seqikpy/utils.py:313-329``)."""
    dofs = ["ThC_roll", "ThC_yaw", "ThC_pitch", "CTr_pitch", "CTr_roll", "FTi_pitch"] + \
        (["TiTa_pitch"] if claw_is_end_effector else [])
    return np.stack([np.asarray(angle_dict[f"{leg}_leg"][d], dtype=np.float64) for d in dofs], axis=1)


def interpolate_signal(signal, original_ts, new_ts):
    """Resamples one series from time step ``original_ts`` to ``new_ts`` with a shape-preserving cubic (PCHIP) over
    ``[0, N * original_ts)`` (``seqikpy/utils.py:332-349``).  As there: if the interpolation fails, infinities and the
    last sample are zeroed IN the caller's array and it is tried once more."""
    from scipy.interpolate import pchip_interpolate
    total = signal.shape[0] * original_ts
    x_old, x_new = np.arange(0, total, original_ts), np.arange(0, total, new_ts)
    try:
        return np.array(pchip_interpolate(x_old, signal, x_new))
    except BaseException:  # noqa: B036 -- the reference's own breadth
        signal[np.isinf(signal)] = 0
        signal[-1] = 0
        return np.array(pchip_interpolate(x_old, signal, x_new))


def interpolate_joint_angles(joint_angles_dict, **kwargs):
    """``interpolate_signal`` over every series of a joint-angle dictionary (``run_ik_and_fk``'s first result);
    ``original_ts`` / ``new_ts`` as keyword arguments (``seqikpy/utils.py:352-360``)."""
    return {dof: interpolate_signal(signal=series, **kwargs) for dof, series in joint_angles_dict.items()}
