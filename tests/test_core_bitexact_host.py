"""CPU tier: the kernel's device functions (csrc/seqik_core.hpp), compiled for the host by
tests/harness, must reproduce the generic C oracle BIT FOR BIT -- angles, FK, scipy status and
nfev.  The same comparison runs on the GPU in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import leg_arrays, load_golden


def _cmp(h, o):
    assert np.array_equal(h["angles"], o["angles"])
    assert np.array_equal(h["fk"], o["fk"])
    assert np.array_equal(h["status"], o["status"])
    assert np.array_equal(h["nfev"], o["nfev"])


def test_sincos_identical(oracle, host_harness):
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.uniform(-3.3, 3.3, 5000), [0.0, -0.0, np.pi, -np.pi, 1e-9]]):
        assert oracle.sincos(x) == host_harness.sincos(x)


@pytest.mark.parametrize("name,frames", [("anipose_shipped", 1500), ("df3d_1000", 1000), ("df3d_100", 100)])
def test_core_equals_oracle_on_recordings(oracle, host_harness, name, frames):
    z = load_golden(name)
    for leg in z["legs"]:
        pose, seg, b, seeds = leg_arrays(z, str(leg))
        pose = pose[:frames]
        _cmp(host_harness.run(pose, seg, b, seeds), oracle.seq_leg(pose, seg, b, seeds))


def test_core_equals_oracle_through_degenerate_episode(oracle, host_harness):
    z = load_golden("anipose_shipped")
    pose, seg, b, seeds = leg_arrays(z, "LF")
    _cmp(host_harness.run(pose[200:400], seg, b, seeds), oracle.seq_leg(pose[200:400], seg, b, seeds))


def test_core_without_diagnostics_gives_same_angles(host_harness):
    z = load_golden("df3d_100")
    pose, seg, b, seeds = leg_arrays(z, "LH")
    a = host_harness.run(pose, seg, b, seeds, diag=True)
    c = host_harness.run(pose, seg, b, seeds, diag=False)
    assert np.array_equal(a["angles"], c["angles"]) and np.array_equal(a["fk"], c["fk"])


@pytest.mark.parametrize("first,last", [(1, 1), (1, 2), (1, 3), (2, 2), (2, 4), (3, 4), (4, 4)])
def test_core_stage_subsets(oracle, host_harness, first, last):
    z = load_golden("df3d_100")
    pose, seg, b, seeds = leg_arrays(z, "RH")
    full = oracle.seq_leg(pose, seg, b, seeds)
    prior = np.zeros_like(full["angles"])
    prior[:, : 2 * (first - 1)] = full["angles"][:, : 2 * (first - 1)]
    o = oracle.seq_leg(pose, seg, b, seeds, first, last, prior_angles=prior)
    h = host_harness.run(pose, seg, b, seeds, first, last, prior=prior)
    ncol = min(2 * last, 7)
    assert np.array_equal(h["angles"][:, :ncol], o["angles"][:, :ncol])
    assert np.array_equal(h["angles"][:, :ncol], full["angles"][:, :ncol])
    assert np.array_equal(h["status"][:, first - 1:last], o["status"][:, first - 1:last])
    if last == 4:
        assert np.array_equal(h["fk"], full["fk"])


def test_core_equals_oracle_on_synthetic(oracle, host_harness):
    """i.i.d. and smooth synthetic key points (bench workload), incl. targets
    that jump across the workspace from frame to frame."""
    from seqikpy_amd import data, synthetic, utils
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    for variant in ("iid", "smooth"):
        pose = synthetic.synthetic_pose(2, 40, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                        variant=variant)
        for li, leg in enumerate(legs):
            seg, b, seeds = oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
            for s in range(2):
                _cmp(host_harness.run(pose[s, li], seg, b, seeds), oracle.seq_leg(pose[s, li], seg, b, seeds))


def test_core_rejects_bad_parameters(host_harness):
    z = load_golden("df3d_100")
    pose, seg, b, seeds = leg_arrays(z, "RF")
    bad = seeds.copy()
    bad[4 + 3] = 10.0  # stage-2 roll seed outside [-pi, pi]
    with pytest.raises(ValueError):
        host_harness.run(pose[:2], seg, b, bad)
    bb = b.copy()
    bb[6] = (0.0, 0.0)
    with pytest.raises(ValueError):
        host_harness.run(pose[:2], seg, bb, seeds)


def test_continuation_with_init_angles(oracle, host_harness):
    """A recording solved in two pieces, the second started from the last angles of the first, equals the
    one-piece run bit for bit (oracle and device core)."""
    z = load_golden("df3d_1000")
    pose, seg, b, seeds = leg_arrays(z, "LF")
    full = oracle.seq_leg(pose[:200], seg, b, seeds)
    cut = 83
    for run in (lambda p, init: oracle.seq_leg(p, seg, b, seeds, init=init),
                lambda p, init: host_harness.run(p, seg, b, seeds, init=init)):
        a = run(pose[:cut], None)
        c = run(pose[cut:200], a["angles"][-1])
        assert np.array_equal(np.concatenate([a["angles"], c["angles"]]), full["angles"])
        assert np.array_equal(np.concatenate([a["fk"], c["fk"]]), full["fk"])


def test_core_equals_oracle_on_random_legs_and_nasty_targets(oracle, host_harness):
    """Made-up legs (lengths, limits, seeds on bounds) with unreachable / degenerate / repeated targets."""
    from conftest import random_leg_case
    rng = np.random.default_rng(20241022)
    for _ in range(96):
        pose, seg, b, seeds = random_leg_case(rng, 32)
        o = oracle.seq_leg(pose, seg, b, seeds)
        assert np.isfinite(o["angles"]).all() and np.isfinite(o["fk"]).all()
        assert (o["angles"] >= b[:, 0]).all() and (o["angles"] <= b[:, 1]).all()
        _cmp(host_harness.run(pose, seg, b, seeds), o)


def _unbound_some_joints(rng, b):
    """IKPy links without limits have bounds (-inf, inf) (`ikpy.link.URDFLink` default) and scipy treats one-sided
    limits too: open some sides of some joints of a made-up leg."""
    b = b.copy()
    for j in range(7):
        kind = rng.integers(0, 4)
        if kind == 1:
            b[j, 0] = -np.inf
        elif kind == 2:
            b[j, 1] = np.inf
        elif kind == 3:
            b[j] = (-np.inf, np.inf)
    return b


def test_core_equals_oracle_with_open_and_one_sided_limits(oracle, host_harness):
    """Joints without limits / with a limit on one side only: the finiteness tests of CL_scaling_vector,
    make_strictly_feasible and the finite-difference step (folded into per-leg constants in the kernel) against the
    oracle's isfinite() calls."""
    from conftest import random_leg_case
    rng = np.random.default_rng(4711)
    opened = 0
    for _ in range(48):
        pose, seg, b, seeds = random_leg_case(rng, 24)
        b = _unbound_some_joints(rng, b)
        opened += int(np.isinf(b).sum())
        o = oracle.seq_leg(pose, seg, b, seeds)
        assert np.isfinite(o["angles"]).all() and np.isfinite(o["fk"]).all()
        _cmp(host_harness.run(pose, seg, b, seeds), o)
    assert opened > 100


def test_latency_form_of_the_reflective_step_gives_the_same_bits(host_harness):
    """run_stage<..., LAT> (the 256-register build of the stage pipeline) uses select_step_reflective_ilp: every quotient
    formed unconditionally and selected, stride-independent parts first.  Same operations on the same operands wherever a
    value is used -> the same bits as the compact form, on made-up inputs that hit every branch: steps that leave through
    one bound / both, zero directions, points on a bound, tiny and huge radii, one and two unknowns."""
    import ctypes
    dp = ctypes.POINTER(ctypes.c_double)
    fn = host_harness.lib.harness_select_step
    fn.restype = None
    fn.argtypes = [ctypes.c_int32, ctypes.c_int32, dp, dp, dp, dp, dp, dp, dp, ctypes.c_double, dp, dp, ctypes.c_double, dp]
    rng = np.random.default_rng(17)
    for case in range(20000):
        na = 2 if rng.random() < 0.8 else 1
        lb = rng.uniform(-3.0, 0.0, 2)
        ub = lb + rng.choice([0.05, 0.5, 3.0], 2) * rng.uniform(0.5, 1.0, 2)
        if rng.random() < 0.2:
            ub[rng.integers(0, 2)] = 0.0
            lb = np.minimum(lb, ub - 0.1)
        u = rng.random(2)
        u[rng.random(2) < 0.15] = rng.choice([1e-17, 1.0 - 1e-16])          # next to a bound
        x = lb + u * (ub - lb)
        d = np.sqrt(np.maximum(np.minimum(x - lb, ub - x), 1e-300))
        Jh = rng.standard_normal((3, 2)) * d
        g_h = rng.standard_normal(2) * d
        diag_h = np.abs(rng.standard_normal(2)) * rng.choice([0.0, 1.0], 2)
        Delta = float(rng.choice([1e-6, 1e-2, 1.0, 50.0]) * rng.uniform(0.5, 2.0))
        p_h = rng.standard_normal(2)
        p_h *= Delta / np.linalg.norm(p_h[:na])
        if rng.random() < 0.1:
            p_h[rng.integers(0, 2)] = 0.0                                       # a zero direction
        if na == 1:
            p_h[1] = 0.0
            Jh[:, 1] = 0.0
            g_h[1] = 0.0
            diag_h[1] = 0.0
        p = d * p_h
        # make sure the step leaves the bounds (the caller only calls select_step_reflective then)
        if np.all((x + p >= lb) & (x + p <= ub)):
            p_h *= 10.0 * (ub - lb).max() / max(np.abs(p).max(), 1e-300)
            p = d * p_h
        theta = max(0.995, 1.0 - float(rng.random()) * 1e-2)
        outs = []
        for ilp in (0, 1):
            out = np.zeros(5)
            args = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, Jh, diag_h, g_h, p, p_h, d)]
            fn(na, ilp, *[a.ctypes.data_as(dp) for a in args], Delta, lb.ctypes.data_as(dp), ub.ctypes.data_as(dp), theta,
               out.ctypes.data_as(dp))
            outs.append(out)
        assert np.array_equal(outs[0], outs[1], equal_nan=True), (case, outs)
