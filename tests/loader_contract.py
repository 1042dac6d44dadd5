"""TEST INFRASTRUCTURE -- the contract the reference's consumers place on the pipeline's output files, restated
from how they are read (not copied):

  * ``seqikpy/utils.py:235-245``  ``load_file`` = ``pickle.load`` of the whole file;
  * ``seqikpy/visualization.py:191-213``  ``load_grid_plot_data(data_path)``: ``body_joint_angles.pkl`` if present,
    else ``head_joint_angles.pkl`` merged with ``leg_joint_angles.pkl`` (optional), plus ``pose3d_aligned.pkl``;
  * ``seqikpy/visualization.py:443-492``  ``plot_3d_points(ax, points3d, t=...)``: ``points3d`` maps a segment name to
    an array indexed ``[t, :, 0..2]``; segments with more than 3 points are drawn as lines, the others need an entry
    in ``marker_types`` (``R_head``, ``L_head``, ``Neck`` by default); names containing "R" / "L" pick the colour map;
  * joint-angle consumers index ``joint_angles[name][t]`` with names ``Angle_<leg>_<dof>`` / head angle names.
"""
import os
import pickle

import numpy as np

DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
HEAD_ANGLES = ["Angle_head_roll", "Angle_head_pitch", "Angle_head_yaw", "Angle_antenna_yaw_L", "Angle_antenna_pitch_L",
               "Angle_antenna_yaw_R", "Angle_antenna_pitch_R"]
DEFAULT_MARKERS = {"R_head", "L_head", "Neck"}


def load_file(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def load_grid_plot_data(data_path):
    """Same files, same precedence as the reference's loader."""
    p = lambda n: os.path.join(str(data_path), n)  # noqa: E731
    if os.path.isfile(p("body_joint_angles.pkl")):
        joint_angles = load_file(p("body_joint_angles.pkl"))
    else:
        head = load_file(p("head_joint_angles.pkl"))
        legs = load_file(p("leg_joint_angles.pkl")) if os.path.isfile(p("leg_joint_angles.pkl")) else {}
        joint_angles = {**head, **legs}
    return joint_angles, load_file(p("pose3d_aligned.pkl"))


def check_joint_angles(joint_angles, legs, n_frames, with_head):
    assert isinstance(joint_angles, dict)
    want = ([n for n in HEAD_ANGLES] if with_head else []) + [f"Angle_{l}_{d}" for l in legs for d in DOFS]
    assert set(want) <= set(joint_angles), sorted(set(want) - set(joint_angles))
    for k in want:
        a = joint_angles[k]
        assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.shape == (n_frames,), (k, a.dtype, a.shape)
        assert np.isfinite(a).all(), k
        float(a[0]), float(a[n_frames - 1])  # indexable per frame


def check_points3d(points3d, n_frames, leg_points):
    """What plot_3d_points needs from a pose / forward-kinematics dictionary."""
    assert isinstance(points3d, dict) and points3d
    for name, arr in points3d.items():
        assert isinstance(name, str) and isinstance(arr, np.ndarray) and arr.ndim == 3 and arr.shape[2] == 3, name
        assert arr.shape[0] in (1, n_frames), (name, arr.shape)   # Neck is stored once
        order = arr.shape[1]
        if order <= 3:
            assert name in DEFAULT_MARKERS, f"{name}: {order} points would need its own marker type"
        if name.endswith("_leg"):
            assert order == leg_points and arr.shape[0] == n_frames, (name, arr.shape)
            assert np.isfinite(arr).all(), name
            arr[0, :, 0], arr[n_frames - 1, :, 2]
