"""CPU tier: the C-ABI shared library loads and exports every symbol include/seqik.h declares;
argument validation works without a GPU; a missing library fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG_PARENT, ROOT, leg_arrays, load_golden


def declared_functions():
    text = open(os.path.join(ROOT, "include", "seqik.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(seqik_[a-z_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    assert declared_functions() == sorted(["seqik_abi_version", "seqik_device_count", "seqik_last_error",
                                           "seqik_device_attributes", "seqik_release_workspaces",
                                           "seqik_validate_legs", "seqik_peer_alloc", "seqik_peer_free", "seqik_peer_export",
                                           "seqik_peer_open", "seqik_peer_close", "seqik_peer_copy", "seqik_solve_seq", "seqik_solve_seq_device",
                                           "seqik_head_angles", "seqik_head_angles_device",
                                           "seqik_head_angles_ex", "seqik_head_angles_ex_device", "seqik_signed_angles",
                                           "seqik_validate_legs_generic", "seqik_solve_generic",
                                           "seqik_solve_generic_device",
                                           "seqik_host_alloc", "seqik_host_free", "seqik_host_register",
                                           "seqik_host_unregister", "seqik_frame_chunk_plan", "seqik_selftest_div_sqrt", "seqik_selftest_sqrt_pos",
                                           "seqik_check_faults", "seqik_check_faults_stream", "seqik_stream_open", "seqik_stream_submit",
                                           "seqik_stream_wait", "seqik_stream_reset_carry", "seqik_stream_set_carry", "seqik_stream_close",
                                           "seqik_align_stats_open", "seqik_align_stats_add", "seqik_align_stats_finish",
                                           "seqik_align_stats_reset", "seqik_align_stats_close"])


def test_library_exports_every_declared_symbol(hiplib):
    lib = hiplib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted(hiplib.EXPORTED_SYMBOLS) == declared_functions()
    assert lib.seqik_abi_version() == 7 == hiplib.ABI_VERSION


def test_struct_layout_matches_header(hiplib):
    assert ctypes.sizeof(hiplib.SeqikLegParams) == 8 * (4 + 14 + 27)
    assert ctypes.sizeof(hiplib.SeqikOptions) == 88  # ABI 3: + chunk_flags, chunk_states, chunk_resume
    assert hiplib.SeqikOptions.frame_chunk.offset == 32 and hiplib.SeqikOptions.chunk_tol.offset == 40
    assert hiplib.SeqikOptions.frame_lead.offset == 52 and hiplib.SeqikOptions.chunk_stats.offset == 56
    assert hiplib.SeqikOptions.chunk_flags.offset == 64 and hiplib.SeqikOptions.chunk_states.offset == 72
    assert hiplib.SeqikOptions.chunk_resume.offset == 80
    assert ctypes.sizeof(hiplib.SeqikLayout) == 48 and ctypes.sizeof(hiplib.SeqikAffine) == 56


def test_validate_legs_error_codes(hiplib):
    z = load_golden("df3d_100")
    _, seg, b, seeds = leg_arrays(z, "RF")
    good = hiplib.leg_params_from_arrays(seg, b, seeds)
    hiplib.validate_legs([good])
    bad_seed = seeds.copy()
    bad_seed[1] = 4.0
    with pytest.raises(ValueError, match="Initial guess is outside of provided bounds"):
        hiplib.validate_legs([hiplib.leg_params_from_arrays(seg, b, bad_seed)])
    hiplib.validate_legs([hiplib.leg_params_from_arrays(seg, b, bad_seed)], 2, 4)  # stage 1 not run: not checked
    bb = b.copy()
    bb[2] = (1.0, -1.0)
    with pytest.raises(ValueError, match="strictly less"):
        hiplib.validate_legs([hiplib.leg_params_from_arrays(seg, bb, seeds)])
    with pytest.raises(ValueError, match="Maximum stage number is 4"):
        hiplib.validate_legs([good], 2, 5)
    # floating-point contract (csrc/seqik_core.hpp): a limit of exactly 0 is served (the shipped tables have three), a
    # non-zero limit below 2^-600 in magnitude is refused
    tiny = b.copy()
    tiny[3] = (tiny[3][0], 1e-250)
    with pytest.raises(ValueError, match="smaller than 2\\^-600"):
        hiplib.validate_legs([hiplib.leg_params_from_arrays(seg, tiny, seeds)])
    assert any(0.0 in (lo, hi) for lo, hi in b)


def test_solve_rejects_bad_arguments_before_touching_the_gpu(hiplib):
    z = load_golden("df3d_100")
    pose, seg, b, seeds = leg_arrays(z, "RF")
    lp = hiplib.leg_params_from_arrays(seg, b, seeds)
    with pytest.raises(ValueError):
        hiplib.solve_seq(pose[None, None, :4], [lp], 3, 2)
    with pytest.raises(ValueError):
        hiplib.solve_seq(pose[:4], [lp])  # wrong rank
    bad = seeds.copy()
    bad[26] = 9.0  # claw seed outside +-pi
    with pytest.raises(ValueError, match="outside of provided bounds"):
        hiplib.solve_seq(pose[None, None, :4], [hiplib.leg_params_from_arrays(seg, b, bad)])


def test_non_finite_key_points_raise_like_scipy(hiplib):
    z = load_golden("df3d_100")
    pose, seg, b, seeds = leg_arrays(z, "RF")
    bad = pose[None, None, :8].copy()
    bad[0, 0, 3, 2, 1] = np.nan
    with pytest.raises(ValueError, match="Residuals are not finite"):
        hiplib.solve_seq(bad, [hiplib.leg_params_from_arrays(seg, b, seeds)])
    with pytest.raises(ValueError, match="Residuals are not finite"):
        hiplib.solve_generic(bad, [hiplib.leg_params_from_arrays(seg, b, seeds)])


def test_missing_library_fails_loudly():
    code = ("import sys; sys.path.insert(0, %r); from seqikpy_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.SeqikLibraryError as e:\n    print('LOUD', e)\n" % PKG_PARENT)
    env = dict(os.environ, SEQIK_LIB="/nonexistent/libseqik_hip.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "LOUD" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr


def test_product_does_not_import_the_oracle():
    """The package must not reach into oracle/ (it is test infrastructure)."""
    pkg = os.path.join(PKG_PARENT, "seqikpy_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            for needle in ("import oracle", "from oracle", "c_oracle", "ref_import", "libseqik_oracle"):
                assert needle not in src, (fn, needle)
    csrc = os.path.join(PKG_PARENT, "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".hpp", ".h", ".cpp")):
            for line in open(os.path.join(csrc, fn)):
                if line.lstrip().startswith("#include"):
                    assert "oracle" not in line, (fn, line)


def test_committed_pmc_summary_belongs_to_these_kernel_sources(hiplib):
    """bench.py prices its roofline with the PMC counters of profiles/traffic_r06*.json only while they were measured on
    the kernel code of this tree (`_lib.csrc_sha256`: code only, comments and white space do not count).  If this fails the
    solver kernels changed: re-run `bash scripts/gpu_profile.sh r06` / `r06s --variant smooth` + `scripts/summarize_profile.py r06 r06` / `r06s r06`
    (and `bash scripts/gpu_latency_profile.sh r06` + `scripts/latency_floor.py r06 r06` for the latency-bound kernels)."""
    import json
    assert hiplib._code_only("a = b; // note\n/* block\n comment */  c  =\td;") == "a = b; c = d;"
    for name in ("traffic_r06.json", "traffic_r06_smooth.json"):
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert t["csrc_files"] == hiplib.KERNEL_SOURCES
        assert t["csrc_sha256"] == hiplib.csrc_sha256(), name
    lat = json.load(open(os.path.join(ROOT, "profiles", "r06_latency_floor.json")))
    assert lat["csrc_files"] == hiplib.LATENCY_SOURCES and lat["csrc_sha256"] == hiplib.csrc_sha256(hiplib.LATENCY_SOURCES)
