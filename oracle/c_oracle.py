"""ORACLE / TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/seqik_oracle.c.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``seqikpy_amd``) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libseqik_oracle.so")

DOFS = ["ThC_yaw", "ThC_pitch", "ThC_roll", "CTr_pitch", "CTr_roll", "FTi_pitch", "TiTa_pitch"]
SEGMENTS = ["Coxa", "Femur", "Tibia", "Tarsus"]

ERRORS = {
    -2: "Initial guess is outside of provided bounds",
    -3: "Each lower bound must be strictly less than each upper bound.",
    -4: "bad argument",
}

_lib = None


def build(force: bool = False) -> str:
    """Compiles the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "seqik_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        lp = ctypes.POINTER(ctypes.c_int64)
        L.oracle_seq_leg.restype = ctypes.c_int
        L.oracle_seq_leg.argtypes = [dp, ctypes.c_int64, dp, dp, dp, ctypes.c_int, ctypes.c_int,
                                     dp, dp, ip, ip, lp, ip, dp]
        L.oracle_generic_leg.restype = ctypes.c_int
        L.oracle_generic_leg.argtypes = [dp, ctypes.c_int64, dp, dp, dp, dp, dp, ip, ip, lp]
        L.oracle_stage_solve.restype = ctypes.c_int
        L.oracle_stage_solve.argtypes = [ctypes.c_int, dp, dp, dp, dp, dp, dp, ip, ip]
        L.oracle_stage_fk.restype = ctypes.c_int
        L.oracle_stage_fk.argtypes = [ctypes.c_int, dp, dp, dp, dp, dp]
        L.oracle_seq_batch.restype = ctypes.c_int
        L.oracle_seq_batch.argtypes = [dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, dp, dp, dp, dp, dp]
        L.oracle_sincos.restype = None
        L.oracle_sincos.argtypes = [ctypes.c_double, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if a is not None else None


def leg_params(leg, bounds_dof, body_size, initial_angles):
    """(seg[4], bounds[7,2], seeds[27]) from the reference's dict-shaped inputs."""
    seg = np.array([body_size[f"{leg}_{s}"] for s in SEGMENTS], dtype=np.float64)
    bounds = np.array([bounds_dof[f"{leg}_{d}"] for d in DOFS], dtype=np.float64)
    seeds = np.concatenate([np.asarray(initial_angles[leg][f"stage_{k}"], dtype=np.float64)
                            for k in (1, 2, 3, 4)])
    assert seeds.shape == (27,)
    return seg, bounds, seeds


def seq_leg(pose, seg, bounds, seeds, first_stage=1, last_stage=4, prior_angles=None, want_fk=True, init=None):
    """Runs one leg through the C oracle.  Returns dict(angles[N,7], fk[N,9,3], status[N,4], nfev[N,4])."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    n = pose.shape[0]
    assert pose.shape == (n, 5, 3)
    angles = np.zeros((n, 7)) if prior_angles is None else np.ascontiguousarray(prior_angles, dtype=np.float64).copy()
    fk = np.zeros((n, 9, 3)) if want_fk else None
    status = np.full((n, 4), -1, dtype=np.int32)
    nfev = np.zeros((n, 4), dtype=np.int32)
    ef = ctypes.c_int64(-1)
    es = ctypes.c_int32(-1)
    seg = np.ascontiguousarray(seg, dtype=np.float64)
    bounds = np.ascontiguousarray(bounds, dtype=np.float64)
    seeds = np.ascontiguousarray(seeds, dtype=np.float64)
    rc = lib().oracle_seq_leg(_dp(pose), n, _dp(seg), _dp(bounds), _dp(seeds), first_stage, last_stage,
                              _dp(angles), _dp(fk), _ip(status), _ip(nfev), ctypes.byref(ef), ctypes.byref(es),
                              _dp(np.ascontiguousarray(init, dtype=np.float64)) if init is not None else None)
    if rc != 0:
        raise ValueError(f"{ERRORS.get(rc, rc)} (frame {ef.value}, stage {es.value})")
    return dict(angles=angles, fk=fk, status=status, nfev=nfev)


def seq_batch(pose, segs, bounds, seeds, want_fk=True):
    """[S, L, N, 5, 3] key points -> dict(angles [S, L, N, 7], fk [S, L, N, 9, 3]); one C call (the GIL
    is released for its whole duration, so callers can run one batch per thread)."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    S, L, N = pose.shape[:3]
    segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(L, 4)
    bounds = np.ascontiguousarray(bounds, dtype=np.float64).reshape(L, 7, 2)
    seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(L, 27)
    angles = np.zeros((S, L, N, 7))
    fk = np.zeros((S, L, N, 9, 3)) if want_fk else None
    rc = lib().oracle_seq_batch(_dp(pose), S, L, N, _dp(segs), _dp(bounds), _dp(seeds), _dp(angles), _dp(fk))
    if rc != 0:
        raise ValueError(ERRORS.get(rc, str(rc)))
    return dict(angles=angles, fk=fk)


def generic_leg(pose, seg, bounds, seed9, want_fk=True):
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    n = pose.shape[0]
    angles = np.zeros((n, 7))
    fk = np.zeros((n, 9, 3)) if want_fk else None
    status = np.full((n,), -1, dtype=np.int32)
    nfev = np.zeros((n,), dtype=np.int32)
    ef = ctypes.c_int64(-1)
    seg = np.ascontiguousarray(seg, dtype=np.float64)
    bounds = np.ascontiguousarray(bounds, dtype=np.float64)
    seed9 = np.ascontiguousarray(seed9, dtype=np.float64)
    rc = lib().oracle_generic_leg(_dp(pose), n, _dp(seg), _dp(bounds), _dp(seed9), _dp(angles), _dp(fk),
                                  _ip(status), _ip(nfev), ctypes.byref(ef))
    if rc != 0:
        raise ValueError(f"{ERRORS.get(rc, rc)} (frame {ef.value})")
    return dict(angles=angles, fk=fk, status=status, nfev=nfev)


def stage_solve(stage, seg, bounds, prior_angles, target, x0):
    """One ``calculate_ik`` call on the stage-``stage`` chain: returns (x, status, nfev)."""
    n = (0, 4, 6, 8, 9)[stage]
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    assert x0.shape == (n,)
    out = np.zeros(n)
    st = ctypes.c_int32(0)
    nf = ctypes.c_int32(0)
    pa = np.zeros(7) if prior_angles is None else np.ascontiguousarray(prior_angles, dtype=np.float64)
    seg = np.ascontiguousarray(seg, dtype=np.float64)
    bounds = np.ascontiguousarray(bounds, dtype=np.float64)
    target = np.ascontiguousarray(target, dtype=np.float64)
    rc = lib().oracle_stage_solve(stage, _dp(seg), _dp(bounds), _dp(pa), _dp(target), _dp(x0), _dp(out),
                                  ctypes.byref(st), ctypes.byref(nf))
    if rc != 0:
        raise ValueError(ERRORS.get(rc, str(rc)))
    return out, st.value, nf.value


def stage_fk(stage, seg, bounds, prior_angles, q):
    n = (0, 4, 6, 8, 9)[stage]
    q = np.ascontiguousarray(q, dtype=np.float64)
    pos = np.zeros((n, 3))
    pa = np.zeros(7) if prior_angles is None else np.ascontiguousarray(prior_angles, dtype=np.float64)
    seg = np.ascontiguousarray(seg, dtype=np.float64)
    bounds = np.ascontiguousarray(bounds, dtype=np.float64)
    lib().oracle_stage_fk(stage, _dp(seg), _dp(bounds), _dp(pa), _dp(q), _dp(pos))
    return pos


def sincos(x):
    s = ctypes.c_double()
    c = ctypes.c_double()
    lib().oracle_sincos(float(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


# --- test hooks: algorithm variants kept in the C file for comparison (defaults in brackets) -----------------------
def set_variant(tr2_shortcut=None, closed_form_2x2=None, generic_svd=None, analytic_jacobian=None, generic_rtl=None,
                woodbury_form=None):
    """tr2_shortcut [True]: shortcut of scipy's ten-iteration root search when the Gauss-Newton step is inside the
    trust region (False = the loop verbatim); closed_form_2x2 [True]: closed-form trust-region step for two
    unknowns (False = one-sided Jacobi SVD); generic_svd [False]: SVD-based step for the generic chain (default:
    3 x 3 push-through form); analytic_jacobian [False]: geometric Jacobian instead of scipy's 2-point differences
    (rejected, see the C file); generic_rtl [True]: the generic chain's claw position as a vector pushed through the
    links right to left (False = last column of the left-to-right matrix product, as the sequential stages);
    woodbury_form [1]: 0 = the round-3 form of the 3 x 3 step (two general solves per evaluation)."""
    L = lib()
    if tr2_shortcut is not None:
        L.oracle_set_tr2_shortcut(1 if tr2_shortcut else 0)
    if closed_form_2x2 is not None:
        L.oracle_set_closed_form(1 if closed_form_2x2 else 0)
    if generic_svd is not None:
        L.oracle_set_generic_mode(0 if generic_svd else 2)
    if analytic_jacobian is not None:
        L.oracle_set_analytic_jacobian(1 if analytic_jacobian else 0)
    if generic_rtl is not None:
        L.oracle_set_generic_rtl(1 if generic_rtl else 0)
    if woodbury_form is not None:
        L.oracle_set_woodbury_form(int(woodbury_form))


def reset_variants():
    set_variant(tr2_shortcut=True, closed_form_2x2=True, generic_svd=False, analytic_jacobian=False, generic_rtl=True,
                woodbury_form=1)
