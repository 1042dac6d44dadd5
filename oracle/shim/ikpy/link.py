"""ORACLE ONLY.  Stand-in for ``ikpy.link`` (see package docstring).

Call sites in the reference that define the needed surface:
``seqikpy/kinematic_chain.py:170-198`` (``OriginLink()``,
``URDFLink(name=, origin_translation=, origin_orientation=, rotation=,
joint_type=, bounds=)``).
"""
import numpy as np


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rpy_matrix(roll, pitch, yaw):
    """IKPy geometry.rpy_matrix: Rz(yaw) . Ry(pitch) . Rx(roll)."""
    return _rz(yaw) @ (_ry(pitch) @ _rx(roll))


def axis_rotation_matrix(axis, theta):
    """IKPy geometry.axis_rotation_matrix: un-normalised Rodrigues form."""
    x, y, z = axis
    c, s = np.cos(theta), np.sin(theta)
    return np.array([
        [x ** 2 + (1 - x ** 2) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
        [x * y * (1 - c) + z * s, y ** 2 + (1 - y ** 2) * c, y * z * (1 - c) - x * s],
        [x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z ** 2 + (1 - z ** 2) * c],
    ])


def _homogeneous(rot):
    out = np.eye(4)
    out[:3, :3] = rot
    return out


class Link:
    def __init__(self, name, bounds=None):
        self.name = name
        if bounds is None or tuple(bounds) == (None, None):
            bounds = (-np.inf, np.inf)
        self.bounds = tuple(bounds)

    def get_link_frame_matrix(self, theta):  # pragma: no cover
        raise NotImplementedError


class OriginLink(Link):
    def __init__(self):
        super().__init__(name="Base link", bounds=(-np.inf, np.inf))
        self.joint_type = "fixed"

    def get_link_frame_matrix(self, theta):
        return np.eye(4)


class URDFLink(Link):
    def __init__(self, name, origin_translation, origin_orientation, rotation=None,
                 translation=None, bounds=None, joint_type="revolute", **_ignored):
        super().__init__(name=name, bounds=bounds)
        self.origin_translation = np.asarray(origin_translation, dtype=float)
        self.origin_orientation = np.asarray(origin_orientation, dtype=float)
        self.joint_type = joint_type
        self.has_rotation = (rotation is not None) and joint_type == "revolute"
        self.rotation = None if rotation is None else np.asarray(rotation, dtype=float)
        # Constant part T(origin_translation) . RPY(origin_orientation)
        base = np.eye(4)
        base[:3, 3] = self.origin_translation
        self._base = base @ _homogeneous(rpy_matrix(*self.origin_orientation))

    def get_link_frame_matrix(self, theta):
        if self.has_rotation:
            return self._base @ _homogeneous(axis_rotation_matrix(self.rotation, theta))
        return self._base.copy()
