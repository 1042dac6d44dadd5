"""Kinematic chains of the sequential leg IK -- host-side mirror of the reference's
``seqikpy/kinematic_chain.py`` (``KinematicChainBase`` :27-74, ``KinematicChainSeq`` :77-421,
``KinematicChainGeneric`` :424-532).

The reference builds an ``ikpy.chain.Chain`` per stage (and, for stages 2-4, per frame).
Here a chain is a plain description -- link names, axes, translations, bounds -- that the
HIP library turns into per-(leg, stage) constants once; nothing symbolic, nothing per frame.
The public surface is kept: class names, constructor arguments, ``body_size`` /
``bounds_dof`` attributes, ``create_leg_chain(leg_name, angles=, stage=, t=)`` and the
``ValueError`` conditions; returned chains expose ``.name`` and ``.links[i].name``.
"""
from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Sequence

import numpy as np

from .data import NMF_TEMPLATE
from .utils import calculate_body_size

LEG_NAMES = ["RF", "LF", "RM", "LM", "RH", "LH"]

X_AXIS = (1, 0, 0)
Y_AXIS = (0, 1, 0)
Z_AXIS = (0, 0, 1)


class Link:
    """One link of a chain: ``T(origin_translation) . RPY(origin_orientation) . R(rotation, theta)``."""

    def __init__(self, name: str, origin_translation=(0, 0, 0), origin_orientation=(0, 0, 0),
                 rotation: Optional[Sequence[float]] = None, joint_type: str = "revolute",
                 bounds=(-np.inf, np.inf)):
        self.name = name
        self.origin_translation = np.asarray(origin_translation, dtype=np.float64)
        self.origin_orientation = np.asarray(origin_orientation, dtype=np.float64)
        self.rotation = None if rotation is None else np.asarray(rotation, dtype=np.float64)
        self.joint_type = joint_type
        self.bounds = (float(bounds[0]), float(bounds[1]))

    @property
    def has_rotation(self) -> bool:
        return self.joint_type == "revolute" and self.rotation is not None

    def __repr__(self):
        return f"Link(name={self.name!r}, joint_type={self.joint_type!r}, bounds={self.bounds})"


def OriginLink() -> Link:
    return Link("Base link", joint_type="fixed")


class Chain:
    """Ordered list of links.  ``spec`` records how the HIP library should solve it."""

    def __init__(self, name: str, links: List[Link], spec: Optional[dict] = None):
        self.name = name
        self.links = links
        self.spec = spec or {}

    def __len__(self):
        return len(self.links)

    def __repr__(self):
        return f"Chain(name={self.name!r}, links={[l.name for l in self.links]})"


class KinematicChainBase(ABC):
    """Abstract class to create kinematic chains for the legs.

    Parameters
    ----------
    bounds_dof : Dict[str, tuple]
        Bounds of the joint degrees of freedom, ``"<leg>_<dof>" -> (lb, ub)``.
    legs_list : List[str]
        Legs for which chains are created.
    body_size : Dict[str, float], optional
        Segment sizes; computed from ``NMF_TEMPLATE`` when ``None``.
    """

    def __init__(self, bounds_dof: Dict[str, np.ndarray], legs_list: List[str],
                 body_size: Dict[str, float] = None) -> None:
        self.body_size = calculate_body_size(NMF_TEMPLATE, legs_list) if body_size is None else body_size
        self.bounds_dof = bounds_dof
        self.legs_list = list(legs_list)

    def __call__(self):
        print("Base kinematic chain is called.")

    @abstractmethod
    def create_leg_chain(self, leg_name: str, **kwargs) -> Chain:
        raise NotImplementedError


def _angle(angles, key, t):
    return float(np.asarray(angles[key])[t])


class KinematicChainSeq(KinematicChainBase):
    """Sequential kinematic chain: one chain per stage (yaw-pitch-roll order)."""

    def __call__(self):
        print("Sequential kinematic chain is called.")

    def create_leg_chain(self, leg_name: str, **kwargs) -> Chain:
        angles = kwargs.get("angles", None)
        stage = kwargs.get("stage", 1)
        t = kwargs.get("t", 0)
        if leg_name not in LEG_NAMES:
            raise ValueError(f"Unknown leg name ({leg_name}) is provided!")
        if not 1 <= stage <= 4:
            raise ValueError(f"Unknown stage number ({stage}) number is provided!")
        if stage == 1:
            return self.create_leg_chain_stage_1(leg_name)
        if stage == 2:
            return self.create_leg_chain_stage_2(leg_name, angles=angles, t=t)
        if stage == 3:
            return self.create_leg_chain_stage_3(leg_name, angles=angles, t=t)
        return self.create_leg_chain_stage_4(leg_name, angles=angles, t=t)

    # -- helpers ---------------------------------------------------------------------
    def _b(self, leg, dof):
        return self.bounds_dof[f"{leg}_{dof}"]

    def _rev(self, leg, dof, axis, segment=None):
        tz = 0.0 if segment is None else -self.body_size[f"{leg}_{segment}"]
        return Link(f"{leg}_{dof}", (0, 0, tz), (0, 0, 0), axis, "revolute", self._b(leg, dof))

    def _fix(self, leg, dof, axis, angles, t, segment=None):
        tz = 0.0 if segment is None else -self.body_size[f"{leg}_{segment}"]
        a = _angle(angles, f"Angle_{leg}_{dof}", t)
        rpy = {X_AXIS: (a, 0, 0), Y_AXIS: (0, a, 0), Z_AXIS: (0, 0, a)}[axis]
        return Link(f"{leg}_{dof}", (0, 0, tz), rpy, None, "fixed", self._b(leg, dof))

    def _spec(self, leg, stage, angles, t):
        prior = None
        if stage > 1:
            prior = np.zeros(7)
            dofs = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch"][: 2 * (stage - 1)]
            for i, dof in enumerate(dofs):
                prior[i] = _angle(angles, f"Angle_{leg}_{dof}", t)
        return dict(kind="seq", leg=leg, stage=stage, prior_angles=prior, factory=self)

    # -- stages ----------------------------------------------------------------------
    def create_leg_chain_stage_1(self, leg_name: str) -> Chain:
        """Thorax/coxa yaw and pitch; contains the coxa only."""
        links = [
            OriginLink(),
            self._rev(leg_name, "ThC_yaw", X_AXIS),
            self._rev(leg_name, "ThC_pitch", Y_AXIS),
            self._rev(leg_name, "CTr_pitch", Y_AXIS, "Coxa"),
        ]
        return Chain("chain_stage_1", links, self._spec(leg_name, 1, None, 0))

    def create_leg_chain_stage_2(self, leg_name: str, angles: Dict[str, np.ndarray], t: int) -> Chain:
        """Thorax/coxa roll and coxa/trochanter pitch; coxa + femur."""
        links = [
            OriginLink(),
            self._fix(leg_name, "ThC_yaw", X_AXIS, angles, t),
            self._fix(leg_name, "ThC_pitch", Y_AXIS, angles, t),
            self._rev(leg_name, "ThC_roll", Z_AXIS),
            self._rev(leg_name, "CTr_pitch", Y_AXIS, "Coxa"),
            self._rev(leg_name, "FTi_pitch", Y_AXIS, "Femur"),
        ]
        return Chain("chain_stage_2", links, self._spec(leg_name, 2, angles, t))

    def create_leg_chain_stage_3(self, leg_name: str, angles: Dict[str, np.ndarray], t: int) -> Chain:
        """Coxa/trochanter roll and femur/tibia pitch; coxa + femur + tibia."""
        links = [
            OriginLink(),
            self._fix(leg_name, "ThC_yaw", X_AXIS, angles, t),
            self._fix(leg_name, "ThC_pitch", Y_AXIS, angles, t),
            self._fix(leg_name, "ThC_roll", Z_AXIS, angles, t),
            self._fix(leg_name, "CTr_pitch", Y_AXIS, angles, t, "Coxa"),
            self._rev(leg_name, "CTr_roll", Z_AXIS),
            self._rev(leg_name, "FTi_pitch", Y_AXIS, "Femur"),
            self._rev(leg_name, "TiTa_pitch", Y_AXIS, "Tibia"),
        ]
        return Chain("chain_stage_3", links, self._spec(leg_name, 3, angles, t))

    def create_leg_chain_stage_4(self, leg_name: str, angles: Dict[str, np.ndarray], t: int) -> Chain:
        """Tibia/tarsus pitch; the entire leg."""
        links = [
            OriginLink(),
            self._fix(leg_name, "ThC_yaw", X_AXIS, angles, t),
            self._fix(leg_name, "ThC_pitch", Y_AXIS, angles, t),
            self._fix(leg_name, "ThC_roll", Z_AXIS, angles, t),
            self._fix(leg_name, "CTr_pitch", Y_AXIS, angles, t, "Coxa"),
            self._fix(leg_name, "CTr_roll", Z_AXIS, angles, t),
            self._fix(leg_name, "FTi_pitch", Y_AXIS, angles, t, "Femur"),
            self._rev(leg_name, "TiTa_pitch", Y_AXIS, "Tibia"),
            Link(f"{leg_name}_Claw", (0, 0, -self.body_size[f"{leg_name}_Tarsus"]), (0, 0, 0), (0, 0, 0),
                 "revolute", (-np.pi, np.pi)),
        ]
        return Chain("chain_stage_4", links, self._spec(leg_name, 4, angles, t))


class KinematicChainGeneric(KinematicChainBase):
    """Generic kinematic chain: one 9-link chain for the entire leg."""

    def __call__(self):
        print("Generic kinematic chain is called.")

    def create_leg_chain(self, leg_name: str, **kwargs) -> Chain:
        if leg_name not in LEG_NAMES:
            raise ValueError(f"Unknown leg name ({leg_name}) is provided!")
        b = self.bounds_dof
        size = self.body_size

        def rev(dof, axis, segment=None):
            tz = 0.0 if segment is None else -size[f"{leg_name}_{segment}"]
            return Link(f"{leg_name}_{dof}", (0, 0, tz), (0, 0, 0), axis, "revolute", b[f"{leg_name}_{dof}"])

        links = [
            OriginLink(),
            rev("ThC_roll", Z_AXIS),
            rev("ThC_yaw", X_AXIS),
            rev("ThC_pitch", Y_AXIS),
            rev("CTr_pitch", Y_AXIS, "Coxa"),
            rev("CTr_roll", Z_AXIS),
            rev("FTi_pitch", Y_AXIS, "Femur"),
            rev("TiTa_pitch", Y_AXIS, "Tibia"),
            Link(f"{leg_name}_Claw", (0, 0, -size[f"{leg_name}_Tarsus"]), (0, 0, 0), (0, 0, 0), "revolute",
                 (-np.pi, np.pi)),
        ]
        return Chain("chain", links, dict(kind="generic", leg=leg_name, factory=self))
