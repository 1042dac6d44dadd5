"""Opt-in alias: after ``import seqikpy_amd.as_seqikpy`` the import lines of a reference caller run unchanged --

    import seqikpy_amd.as_seqikpy            # once, before the lines below
    from seqikpy.kinematic_chain import KinematicChainSeq
    from seqikpy.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy.data import BOUNDS, INITIAL_ANGLES

``seqikpy`` and its path modules (``leg_inverse_kinematics``, ``kinematic_chain``, ``alignment``,
``head_inverse_kinematics``, ``data``, ``utils``) are registered in ``sys.modules`` as THIS package's modules (the
reference's package layout: ``/root/reference/setup.py``, ``seqikpy/__init__.py``; what its examples import:
``examples/example_entire_pipeline.py:7-20``).  ``seqikpy.visualization`` is not part of the path and is not provided:
importing it raises ``ImportError`` (install the reference next to this package for plotting and do NOT use the alias
then).  Nothing is registered when a real ``seqikpy`` has already been imported, unless ``install(force=True)``.
"""
import importlib
import sys

ALIASED = ("leg_inverse_kinematics", "kinematic_chain", "alignment", "head_inverse_kinematics", "data", "utils")


def install(force: bool = False) -> bool:
    """Registers the alias.  -> True when ``seqikpy`` now names this package, False when a real ``seqikpy`` was there
    first and was left alone."""
    import seqikpy_amd
    present = sys.modules.get("seqikpy")
    if present is not None and present is not seqikpy_amd and not force:
        return False
    sys.modules["seqikpy"] = seqikpy_amd
    for name in ALIASED:
        sys.modules[f"seqikpy.{name}"] = importlib.import_module(f"seqikpy_amd.{name}")
    return True


def uninstall() -> None:
    """Removes the alias (only the entries that point at this package)."""
    import seqikpy_amd
    for key in [k for k, v in sys.modules.items() if k == "seqikpy" or k.startswith("seqikpy.")]:
        mod = sys.modules[key]
        if mod is seqikpy_amd or getattr(mod, "__name__", "").startswith("seqikpy_amd"):
            del sys.modules[key]


installed = install()
