"""ORACLE / TEST INFRASTRUCTURE ONLY (build container only).

Makes the *reference's own* ``seqikpy`` package importable from
``/root/reference`` without copying it: injects the build-owned ``ikpy`` shim
(``oracle/shim``) and an empty ``cv2`` stub (``seqikpy/utils.py:7`` imports cv2
at module top; only video helpers use it).  ``/root/reference`` does not exist
on the GPU box, so nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may
call this; it is used by ``oracle/gen_golden.py`` and by container-only tests
that skip when the reference is absent.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SEQIK_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "seqikpy"))


def import_reference():
    """Returns the reference's ``seqikpy`` package (leg IK + chain modules loaded)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import seqikpy  # noqa: F401  (the reference's package)
    import seqikpy.kinematic_chain  # noqa: F401
    import seqikpy.leg_inverse_kinematics  # noqa: F401
    return sys.modules["seqikpy"]
