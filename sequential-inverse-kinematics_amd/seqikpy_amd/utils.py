"""Host helpers the leg-IK path needs (subset of the reference's ``seqikpy/utils.py``)."""
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
