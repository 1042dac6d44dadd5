"""ctypes binding of ``libseqik_hip.so`` (C ABI: ``include/seqik.h``).

There is no CPU implementation behind this module: if the HIP library is missing, or no
GPU is visible when a solve is requested, the call raises.  ``build()`` compiles the
library in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

from .data import DOFS, SEGMENTS

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_PKG), "csrc")
# installed (pyproject.toml / setup.py): the built library inside the package wins whenever it exists; the source tree is
# recognised by a file of THIS project next to the package -- a stray top-level csrc/ in site-packages (several native packages
# install one) must not switch an installed package into tree mode (round-5 advice)
_NATIVE = os.path.join(_PKG, "_native")
_IN_TREE = not os.path.isfile(os.path.join(_NATIVE, "libseqik_hip.so")) and os.path.isfile(os.path.join(CSRC, "seqik_hip.hip"))
_LIB_DIR = CSRC if _IN_TREE else _NATIVE
LIB_PATH = os.environ.get("SEQIK_LIB", os.path.join(_LIB_DIR, "libseqik_hip.so"))  # SEQIK_LIB: A/B builds
SOURCES = ["seqik_hip.hip", "seqik_head.hip", "seqik_stream.hip", "seqik_align.hip", "seqik_peer.hip", "seqik_core.hpp",
           "seqik_consts.hpp", "seqik_head.hpp", "seqik_generic.hpp", "seqik_device_scope.hpp", "seqik_hostctx.hpp"]
COMPILE_UNITS = ["seqik_hip.hip", "seqik_head.hip", "seqik_stream.hip", "seqik_align.hip", "seqik_peer.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]

SEQIK_OK = 0
ERR_HIP, ERR_X0, ERR_BOUNDS, ERR_ARG, ERR_STAGE = -1, -2, -3, -4, -5

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


class SeqikLegParams(ctypes.Structure):
    """Mirror of ``struct SeqikLegParams`` (include/seqik.h)."""
    _fields_ = [("seg", ctypes.c_double * 4),
                ("bounds", (ctypes.c_double * 2) * 7),
                ("seeds", ctypes.c_double * 27)]


class SeqikOptions(ctypes.Structure):
    """Mirror of ``struct SeqikOptions`` (include/seqik.h; unchanged since ABI 3)."""
    _fields_ = [("device", ctypes.c_int32), ("block_size", ctypes.c_int32),
                ("stage_events", ctypes.POINTER(ctypes.c_void_p)), ("reserved", ctypes.c_int32 * 4),
                ("frame_chunk", ctypes.c_int32), ("frame_halo", ctypes.c_int32), ("chunk_tol", ctypes.c_double),
                ("chunk_rounds", ctypes.c_int32), ("frame_lead", ctypes.c_int32),
                ("chunk_stats", ctypes.POINTER(ctypes.c_int32)),
                ("chunk_flags", ctypes.POINTER(ctypes.c_uint8)), ("chunk_states", ctypes.POINTER(ctypes.c_double)),
                ("chunk_resume", ctypes.c_int32), ("pad2_", ctypes.c_int32)]


ABI_VERSION = 7
N_CHUNK_STATS = 16
CHUNK_STATS_FIELDS = ("chunks", "frames_per_chunk", "run_in_frames", "repaired_round_1", "repaired_round_2",
                      "repaired_later_rounds", "repaired_by_sweep", "inconsistent_at_first_check",
                      "chains_walked_serially", "chunks_of_those_chains")
CHUNK_FLAG_FAILED_FIRST, CHUNK_FLAG_REPAIRED, CHUNK_FLAG_SWEPT, CHUNK_FLAG_SERIAL = 1, 2, 4, 8
CHUNK_FLAG_LEFT_BLOCKED = 0x80   # input of a lockstep round (chunk_resume = 4), see include/seqik.h


def chunk_stats_dict(stats):
    """int32[16] of ``SeqikOptions.chunk_stats`` -> dict (all zero: the call ran serially; ``chains_walked_serially``:
    chains the automatic mode's per-chain guard handed to the serial walk because more than one of their chunks in eight
    failed the first verification)."""
    return {k: int(v) for k, v in zip(CHUNK_STATS_FIELDS, stats)}


def selftest_div_sqrt(a, b):
    """(a / b, sqrt(a)) as the kernels compute them on the device (``seqik_selftest_div_sqrt``)."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    if a.shape != b.shape:
        raise ValueError("a and b must have the same number of elements")
    q, r = np.empty_like(a), np.empty_like(a)
    rc = load().seqik_selftest_div_sqrt(a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), q.ctypes.data_as(_dp),
                                        r.ctypes.data_as(_dp), a.size)
    if rc != SEQIK_OK:
        _raise(rc)
    return q, r


def selftest_sqrt_pos(a):
    """sqrt(a) as the kernels compute it for the Coleman-Li distances (``sqrt_pos_``: no zero / infinity selects)."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    r = np.empty_like(a)
    rc = load().seqik_selftest_sqrt_pos(a.ctypes.data_as(_dp), r.ctypes.data_as(_dp), a.size)
    if rc != SEQIK_OK:
        _raise(rc)
    return r


def check_faults(stream=None):
    """``seqik_check_faults`` / ``seqik_check_faults_stream``: raises ``SeqikLibraryError`` when a kernel launched through
    the asynchronous device entry points reported a fault (the stage pipeline's watchdog) since the last check.  Call it
    after synchronising.  ``stream=None``: every stream of the process (single-threaded callers); ``stream=<hipStream_t as
    int>`` (0 = the default stream): the launches made on that stream of the current device only -- what a host thread
    that shares the process with other threads' GPUs / streams uses (ABI 6)."""
    rc = load().seqik_check_faults() if stream is None else load().seqik_check_faults_stream(ctypes.c_void_p(int(stream)))
    if rc != SEQIK_OK:
        _raise(rc)


def frame_chunk_plan(n_frames, frame_chunk=-1, frame_halo=0, frame_lead=0):
    """``seqik_frame_chunk_plan``: (frames per chunk, run-in frames, chunks per chain) a call over recordings of
    ``n_frames`` frames would use -- (0, 0, 0) when it would be walked serially.  No GPU needed."""
    opt = SeqikOptions()
    opt.frame_chunk, opt.frame_halo, opt.frame_lead = int(frame_chunk), int(frame_halo), int(frame_lead)
    c, h, k = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0)
    rc = load().seqik_frame_chunk_plan(int(n_frames), ctypes.byref(opt), ctypes.byref(c), ctypes.byref(h), ctypes.byref(k))
    if rc != SEQIK_OK:
        _raise(rc)
    return int(c.value), int(h.value), int(k.value)


class SeqikLayout(ctypes.Structure):
    """Mirror of ``struct SeqikLayout``: element strides of the device buffers."""
    _fields_ = [("pose_chain", ctypes.c_int64), ("pose_row", ctypes.c_int64), ("pose_frame", ctypes.c_int64),
                ("ang_chain", ctypes.c_int64), ("ang_dof", ctypes.c_int64), ("ang_frame", ctypes.c_int64)]


def planar_layout(n_frames: int) -> SeqikLayout:
    """pose [chain][5][frame][3], angles [chain][7][frame]: every key-point row / joint a time series."""
    return SeqikLayout(15 * n_frames, 3 * n_frames, 3, 7 * n_frames, n_frames, 1)


class SeqikAffine(ctypes.Structure):
    """Mirror of ``struct SeqikAffine``: fused AlignPose.align_leg of one leg."""
    _fields_ = [("fixed_coxa", ctypes.c_double * 3), ("scale", ctypes.c_double),
                ("template_coxa", ctypes.c_double * 3)]


def make_affine(fixed_coxa, scale, template_coxa) -> SeqikAffine:
    a = SeqikAffine()
    for i in range(3):
        a.fixed_coxa[i] = float(fixed_coxa[i])
        a.template_coxa[i] = float(template_coxa[i])
    a.scale = float(scale)
    return a


class SeqikLibraryError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


KERNEL_SOURCES = ["seqik_core.hpp", "seqik_consts.hpp", "seqik_hip.hip"]   # what the solver kernels are compiled from
LATENCY_SOURCES = KERNEL_SOURCES + ["seqik_generic.hpp"]   # ... and the generic-chain kernel (profiles/r04_latency_floor.json)


def _code_only(text: str) -> str:
    """C++ source without comments and with white space collapsed (what the compiler sees, roughly)."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return " ".join(text.split())


def csrc_sha256(files=None, read=None) -> str:
    """sha256 over the CODE of the solver kernels' sources (``KERNEL_SOURCES`` order; comments and white space do not
    count): ties a committed PMC summary (profiles/traffic_rNN.json, written by scripts/summarize_profile.py) to the build
    it was measured on.  ``read(name) -> str``: another source of the files (e.g. an older commit)."""
    import hashlib
    h = hashlib.sha256()
    for name in (files or KERNEL_SOURCES):
        if read is not None:
            text = read(name)
        else:
            with open(os.path.join(CSRC, name), "r") as f:
                text = f.read()
        h.update(_code_only(text).encode())
    return h.hexdigest()


def is_stale() -> bool:
    if not _IN_TREE:
        return False          # installed package: the library was compiled when the distribution was built
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force: bool = False) -> str:
    """Compiles ``csrc/seqik_hip.hip`` -> ``csrc/libseqik_hip.so`` (gfx950)."""
    if force or is_stale():
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, u) for u in COMPILE_UNITS]
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


WATCHDOG_LIB_PATH = os.path.join(CSRC, "libseqik_hip_watchdog.so")


def build_watchdog_variant(force: bool = False) -> str:
    """DIAGNOSTIC build for tests/test_gpu_parity.py::test_pipeline_watchdog_is_reported: the same sources with the stage
    pipeline's watchdog limit set to ONE pass (``-DSEQIK_PIPE_SPIN_LIMIT=1``), so that the fault path -- NaN in the
    chain, fault word, ``SEQIK_ERR_HIP`` from the entry points -- can be exercised.  Never loaded by the package itself
    (only through ``SEQIK_LIB`` in a child process of that test)."""
    stale = (not os.path.exists(WATCHDOG_LIB_PATH) or
             any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(WATCHDOG_LIB_PATH) for s in SOURCES))
    if force or stale:
        cmd = ([_hipcc()] + HIPCC_FLAGS + ["-DSEQIK_PIPE_SPIN_LIMIT=1", "-o", WATCHDOG_LIB_PATH] +
               [os.path.join(CSRC, u) for u in COMPILE_UNITS])
        subprocess.check_call(cmd, cwd=CSRC)
    return WATCHDOG_LIB_PATH


def load():
    """Loads the HIP library; raises ``SeqikLibraryError`` if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SeqikLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "seqikpy_amd has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one
        # process cannot both own the GPU, so when torch is installed let it load first: the
        # dynamic linker then binds this library to the already-loaded runtime (same SONAME)
        # and torch tensors / streams / RCCL can be shared with it.
        if os.environ.get("SEQIK_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        L.seqik_abi_version.restype = ctypes.c_int
        if L.seqik_abi_version() != ABI_VERSION:
            raise SeqikLibraryError(f"{LIB_PATH} has ABI {L.seqik_abi_version()}, this package needs {ABI_VERSION}: "
                                    "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
        L.seqik_device_count.restype = ctypes.c_int
        L.seqik_last_error.restype = ctypes.c_char_p
        L.seqik_release_workspaces.restype = ctypes.c_int
        L.seqik_release_workspaces.argtypes = []
        _vpp = ctypes.POINTER(ctypes.c_void_p)
        for name, args in (("seqik_peer_alloc", [_vpp, ctypes.c_size_t]), ("seqik_peer_free", [ctypes.c_void_p]),
                           ("seqik_peer_export", [ctypes.c_void_p, ctypes.c_char_p]),
                           ("seqik_peer_open", [ctypes.c_char_p, _vpp]), ("seqik_peer_close", [ctypes.c_void_p]),
                           ("seqik_peer_copy", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = args
        L.seqik_device_attributes.restype = ctypes.c_int
        L.seqik_device_attributes.argtypes = [ctypes.c_int32, _ip, _ip, ctypes.POINTER(ctypes.c_int64)]
        L.seqik_validate_legs.restype = ctypes.c_int
        L.seqik_validate_legs.argtypes = [ctypes.POINTER(SeqikLegParams), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.seqik_solve_seq.restype = ctypes.c_int
        L.seqik_solve_seq.argtypes = [_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                      ctypes.POINTER(SeqikLegParams), ctypes.c_int32, ctypes.c_int32,
                                      _dp, _dp, _ip, _ip, _dp, ctypes.POINTER(SeqikAffine),
                                      ctypes.POINTER(SeqikOptions)]
        L.seqik_solve_seq_device.restype = ctypes.c_int
        L.seqik_solve_seq_device.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                             ctypes.POINTER(SeqikLegParams), ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.POINTER(SeqikLayout), ctypes.POINTER(SeqikAffine),
                                             ctypes.POINTER(SeqikOptions), ctypes.c_void_p]
        L.seqik_selftest_div_sqrt.restype = ctypes.c_int
        L.seqik_selftest_div_sqrt.argtypes = [_dp, _dp, _dp, _dp, ctypes.c_int64]
        L.seqik_selftest_sqrt_pos.restype = ctypes.c_int
        L.seqik_selftest_sqrt_pos.argtypes = [_dp, _dp, ctypes.c_int64]
        L.seqik_check_faults.restype = ctypes.c_int
        L.seqik_check_faults.argtypes = []
        L.seqik_check_faults_stream.restype = ctypes.c_int
        L.seqik_check_faults_stream.argtypes = [ctypes.c_void_p]
        L.seqik_frame_chunk_plan.restype = ctypes.c_int
        L.seqik_frame_chunk_plan.argtypes = [ctypes.c_int64, ctypes.POINTER(SeqikOptions), _ip, _ip,
                                             ctypes.POINTER(ctypes.c_int64)]
        L.seqik_validate_legs_generic.restype = ctypes.c_int
        L.seqik_validate_legs_generic.argtypes = [ctypes.POINTER(SeqikLegParams), ctypes.c_int32]
        L.seqik_solve_generic.restype = ctypes.c_int
        L.seqik_solve_generic.argtypes = [_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.POINTER(SeqikLegParams), _dp, _dp, _ip, _ip, _dp,
                                          ctypes.POINTER(SeqikAffine), ctypes.POINTER(SeqikOptions)]
        L.seqik_solve_generic_device.restype = ctypes.c_int
        L.seqik_solve_generic_device.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                                 ctypes.POINTER(SeqikLegParams), ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.POINTER(SeqikLayout), ctypes.POINTER(SeqikAffine),
                                                 ctypes.POINTER(SeqikOptions), ctypes.c_void_p]
        L.seqik_head_angles.restype = ctypes.c_int
        L.seqik_head_angles.argtypes = [_dp, _dp, ctypes.c_int64, _dp, ctypes.c_int64, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_int32, _dp, ctypes.POINTER(SeqikOptions)]
        L.seqik_head_angles_device.restype = ctypes.c_int
        L.seqik_head_angles_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                               ctypes.c_void_p, ctypes.c_void_p]
        L.seqik_head_angles_ex.restype = ctypes.c_int
        L.seqik_head_angles_ex.argtypes = [_dp, _dp, ctypes.c_int64, ctypes.c_int32, _dp, ctypes.c_int64, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_int32, _dp, _dp, ctypes.POINTER(SeqikOptions)]
        L.seqik_head_angles_ex_device.restype = ctypes.c_int
        L.seqik_head_angles_ex_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.seqik_signed_angles.restype = ctypes.c_int
        L.seqik_signed_angles.argtypes = [_dp, ctypes.c_int64, _dp, ctypes.c_int64, _dp, ctypes.c_int64, _dp,
                                          ctypes.POINTER(SeqikOptions)]
        L.seqik_host_alloc.restype = ctypes.c_void_p
        L.seqik_host_alloc.argtypes = [ctypes.c_size_t]
        L.seqik_host_free.restype = None
        L.seqik_host_free.argtypes = [ctypes.c_void_p]
        L.seqik_host_register.restype = ctypes.c_int
        L.seqik_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.seqik_host_unregister.restype = ctypes.c_int
        L.seqik_host_unregister.argtypes = [ctypes.c_void_p]
        L.seqik_stream_open.restype = ctypes.c_int
        L.seqik_stream_open.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.POINTER(SeqikLegParams),
                                        ctypes.POINTER(SeqikAffine), ctypes.c_int64, ctypes.c_int64,
                                        ctypes.POINTER(SeqikLayout), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.POINTER(SeqikOptions)]
        L.seqik_stream_submit.restype = ctypes.c_int
        L.seqik_stream_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p]
        L.seqik_stream_set_carry.restype = ctypes.c_int
        L.seqik_stream_set_carry.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
        for name in ("seqik_stream_wait", "seqik_stream_reset_carry", "seqik_stream_close"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.seqik_align_stats_open.restype = ctypes.c_int
        L.seqik_align_stats_open.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int64,
                                             ctypes.POINTER(SeqikOptions)]
        L.seqik_align_stats_add.restype = ctypes.c_int
        L.seqik_align_stats_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.POINTER(SeqikLayout), ctypes.c_void_p]
        L.seqik_align_stats_finish.restype = ctypes.c_int
        L.seqik_align_stats_finish.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, _dp,
                                               ctypes.c_void_p]
        for name in ("seqik_align_stats_reset", "seqik_align_stats_close"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [ctypes.c_void_p]
        _lib = L
        return _lib


EXPORTED_SYMBOLS = ["seqik_abi_version", "seqik_device_count", "seqik_last_error", "seqik_device_attributes", "seqik_release_workspaces",
                    "seqik_validate_legs", "seqik_frame_chunk_plan", "seqik_selftest_div_sqrt", "seqik_selftest_sqrt_pos", "seqik_check_faults", "seqik_check_faults_stream",
                    "seqik_peer_alloc", "seqik_peer_free", "seqik_peer_export", "seqik_peer_open", "seqik_peer_close",
                    "seqik_peer_copy",
                    "seqik_solve_seq", "seqik_solve_seq_device", "seqik_head_angles", "seqik_head_angles_device",
                    "seqik_head_angles_ex", "seqik_head_angles_ex_device", "seqik_signed_angles",
                    "seqik_validate_legs_generic", "seqik_solve_generic", "seqik_solve_generic_device",
                    "seqik_host_alloc", "seqik_host_free", "seqik_host_register", "seqik_host_unregister",
                    "seqik_stream_open", "seqik_stream_submit", "seqik_stream_wait", "seqik_stream_reset_carry", "seqik_stream_set_carry",
                    "seqik_stream_close", "seqik_align_stats_open", "seqik_align_stats_add",
                    "seqik_align_stats_finish", "seqik_align_stats_reset", "seqik_align_stats_close"]


class AlignStats:
    """``seqik_align_stats_*``: order statistics of the seven per-leg series AlignPose reduces (coxa x, y, z and
    the four segment lengths), computed on the GPU.  ``add`` takes host arrays ``(S, L, N, 5, 3)`` (or a raw
    device pointer with ``on_device=True``); ``finish(ranks)`` returns ``(L, 7, len(ranks))``."""

    def __init__(self, n_legs, capacity_frames, device=-1):
        self.n_legs = int(n_legs)
        self._h = ctypes.c_void_p()
        opt = SeqikOptions()
        opt.device = device
        rc = load().seqik_align_stats_open(ctypes.byref(self._h), self.n_legs, int(capacity_frames), ctypes.byref(opt))
        if rc != SEQIK_OK:
            self._h = ctypes.c_void_p()
            _raise(rc)

    def add(self, pose, n_seq=None, n_frames=None, layout=None, on_device=False, stream=0):
        if on_device:
            ptr = ctypes.c_void_p(int(pose))
        else:
            pose = np.ascontiguousarray(pose, dtype=np.float64)
            if layout is None:
                if pose.ndim != 5 or pose.shape[1] != self.n_legs or pose.shape[3:] != (5, 3):
                    raise ValueError(f"pose must have shape (S, {self.n_legs}, N, 5, 3), got {pose.shape}")
                n_seq, n_frames = pose.shape[0], pose.shape[2]
            ptr = ctypes.c_void_p(pose.ctypes.data)
        rc = load().seqik_align_stats_add(self._h, ptr, 1 if on_device else 0, int(n_seq), int(n_frames),
                                          ctypes.byref(layout) if layout is not None else None,
                                          ctypes.c_void_p(stream or None))
        if rc != SEQIK_OK:
            _raise(rc)

    def finish(self, ranks, stream=0):
        ranks = np.ascontiguousarray(ranks, dtype=np.int64)
        out = np.zeros((self.n_legs, 7, len(ranks)))
        rc = load().seqik_align_stats_finish(self._h, ranks.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(ranks),
                                             out.ctypes.data_as(_dp), ctypes.c_void_p(stream or None))
        if rc != SEQIK_OK:
            _raise(rc)
        return out

    def reset(self):
        load().seqik_align_stats_reset(self._h)

    def close(self):
        if self._h:
            load().seqik_align_stats_close(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PEER_HANDLE_BYTES = 64


class PeerBuffer:
    """Device memory that other processes can map (``seqik_peer_alloc`` / ``_export``) or the mapping of another
    process's buffer (``PeerBuffer.open(handle, nbytes)``).  ``tensor(shape)`` views it as a float64 torch tensor."""

    def __init__(self, nbytes, _mapped_ptr=None):
        self.nbytes = int(nbytes)
        self.mapped = _mapped_ptr is not None
        if self.mapped:
            self.ptr = _mapped_ptr
        else:
            p = ctypes.c_void_p()
            rc = load().seqik_peer_alloc(ctypes.byref(p), self.nbytes)
            if rc != SEQIK_OK:
                _raise(rc)
            self.ptr = p.value

    @classmethod
    def open(cls, handle: bytes, nbytes):
        assert len(handle) == PEER_HANDLE_BYTES
        p = ctypes.c_void_p()
        rc = load().seqik_peer_open(handle, ctypes.byref(p))
        if rc != SEQIK_OK:
            _raise(rc)
        return cls(nbytes, _mapped_ptr=p.value)

    def handle(self) -> bytes:
        buf = ctypes.create_string_buffer(PEER_HANDLE_BYTES)
        rc = load().seqik_peer_export(self.ptr, buf)
        if rc != SEQIK_OK:
            _raise(rc)
        return buf.raw

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes // 8,), "typestr": "<f8", "data": (self.ptr, False), "version": 2}

    def tensor(self, shape):
        import torch
        return torch.as_tensor(self, device="cuda").view(shape)

    def close(self):
        if self.ptr:
            rc = load().seqik_peer_close(self.ptr) if self.mapped else load().seqik_peer_free(self.ptr)
            self.ptr = None
            if rc != SEQIK_OK:
                _raise(rc)


def peer_copy(dst_ptr, src_ptr, nbytes, stream=0):
    """``seqik_peer_copy``: asynchronous device-to-device copy on ``stream`` (raw pointers)."""
    rc = load().seqik_peer_copy(dst_ptr, src_ptr, nbytes, ctypes.c_void_p(stream))
    if rc != SEQIK_OK:
        _raise(rc)


def release_workspaces():
    """Frees the per-stream stage hand-off workspaces the library keeps between calls (drains the device)."""
    rc = load().seqik_release_workspaces()
    if rc != SEQIK_OK:
        _raise(rc)


def device_attributes(device=0):
    """(compute units, peak clock in kHz, HBM bytes) of a GPU."""
    cu, khz, mem = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0)
    rc = load().seqik_device_attributes(device, ctypes.byref(cu), ctypes.byref(khz), ctypes.byref(mem))
    if rc != SEQIK_OK:
        _raise(rc)
    return cu.value, khz.value, mem.value


def solve_generic(pose, legs, want_fk=True, want_diag=False, device=-1, block_size=0, affine=None, init_angles=None,
                  lanes_per_wave=0, lane_groups=True, chain_queue=0):
    """``seqik_solve_generic`` on host arrays: pose (S, L, N, 5, 3) -> dict(angles (S, L, N, 7),
    fk (S, L, N, 9, 3) or None, status / nfev (S, L, N) or None).  ``chain_queue`` (``SeqikOptions.reserved[1]``): batches
    of full wavefronts on persistent wavefronts whose lanes pull chains from a per-leg counter -- 0 = automatic (at least
    four chains per lane of the GPU), 1 = never, 2 = whenever full wavefronts are used; same bits either way."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    if pose.ndim != 5 or pose.shape[3:] != (5, 3):
        raise ValueError(f"pose must have shape (S, L, N, 5, 3), got {pose.shape}")
    S, L, N = pose.shape[:3]
    if len(legs) != L:
        raise ValueError("one SeqikLegParams per leg expected")
    _check_finite(pose)
    angles = np.zeros((S, L, N, 7))
    fk = np.full((S, L, N, 9, 3), np.nan) if want_fk else None
    status = np.full((S, L, N), -1, dtype=np.int32) if want_diag else None
    nfev = np.zeros((S, L, N), dtype=np.int32) if want_diag else None
    if init_angles is not None:
        init_angles = np.ascontiguousarray(init_angles, dtype=np.float64)
        if init_angles.shape != (S, L, 7):
            raise ValueError(f"init_angles must have shape {(S, L, 7)}")
    opt = SeqikOptions()
    opt.device = device
    opt.block_size = block_size
    opt.reserved[0] = lanes_per_wave
    opt.reserved[1] = chain_queue
    opt.reserved[3] = 0 if lane_groups else 3  # measurements: thin waves without the split over groups of 8 lanes
    rc = load().seqik_solve_generic(pose.ctypes.data_as(_dp), S, L, N, (SeqikLegParams * L)(*legs),
                                    angles.ctypes.data_as(_dp), fk.ctypes.data_as(_dp) if fk is not None else None,
                                    status.ctypes.data_as(_ip) if status is not None else None,
                                    nfev.ctypes.data_as(_ip) if nfev is not None else None,
                                    init_angles.ctypes.data_as(_dp) if init_angles is not None else None,
                                    _affine_array(affine, L), ctypes.byref(opt))
    if rc != SEQIK_OK:
        _raise(rc)
    return dict(angles=angles, fk=fk, status=status, nfev=nfev)


def head_angles(r_head, l_head, neck, rest_head_pitch, rest_antenna_pitch, compute_ant=True, device=-1, head_roll=None):
    """``seqik_head_angles_ex`` on host arrays: (N, K, 3), (N, K, 3), neck (3,) or (N, 3) -> (7 or 3, N).

    K = key points per side: point 0 gives the head angles, point 1 (the antenna tip) the antenna angles; K = 1 is
    allowed without them.  ``head_roll`` (N,): derotate the antenna vectors by this roll instead of the frame's own."""
    r_head = np.ascontiguousarray(r_head, dtype=np.float64)
    l_head = np.ascontiguousarray(l_head, dtype=np.float64)
    n = r_head.shape[0]
    if r_head.ndim != 3 or r_head.shape[2] != 3 or r_head.shape[1] < 1 or l_head.shape != r_head.shape:
        raise ValueError("R_head / L_head must have the same shape (N, key points, 3)")
    k = r_head.shape[1]
    if compute_ant and k < 2:
        # what the reference's get_ant_vector runs into (head_inverse_kinematics.py:159-161)
        raise IndexError(f"index 1 is out of bounds for axis 1 with size {k}: the antenna angles need the antenna base "
                         "and tip; call compute_head_angles(compute_ant_angles=False)")
    neck = np.ascontiguousarray(neck, dtype=np.float64).reshape(-1, 3)
    if neck.shape[0] not in (1, n):
        raise ValueError("Neck must hold one point or one point per frame")
    stride = 3 if (neck.shape[0] == n and n > 1) else 0
    roll_p = None
    if head_roll is not None and compute_ant:
        head_roll = np.ascontiguousarray(np.broadcast_to(np.asarray(head_roll, dtype=np.float64).reshape(-1), (n,)))
        roll_p = head_roll.ctypes.data_as(_dp)
    out = np.zeros((7 if compute_ant else 3, n))
    opt = SeqikOptions()
    opt.device = device
    rc = load().seqik_head_angles_ex(r_head.ctypes.data_as(_dp), l_head.ctypes.data_as(_dp), n, k,
                                     neck.ctypes.data_as(_dp), stride, float(rest_head_pitch),
                                     float(rest_antenna_pitch), 1 if compute_ant else 0, roll_p,
                                     out.ctypes.data_as(_dp), ctypes.byref(opt))
    if rc != SEQIK_OK:
        _raise(rc)
    return out


def signed_angles(v1, v2, rot_axis, device=-1):
    """``seqik_signed_angles``: the reference's ``angle_between_segments`` for (N, 3) arrays (either side may be one
    vector, used for every row) -> (N,)."""
    v1 = np.ascontiguousarray(np.asarray(v1, dtype=np.float64).reshape(-1, 3))
    v2 = np.ascontiguousarray(np.asarray(v2, dtype=np.float64).reshape(-1, 3))
    axis = np.ascontiguousarray(np.asarray(rot_axis, dtype=np.float64).reshape(3))
    n = max(v1.shape[0], v2.shape[0])
    if v1.shape[0] not in (1, n) or v2.shape[0] not in (1, n):
        raise ValueError(f"operands could not be broadcast together with shapes {v1.shape} {v2.shape}")
    out = np.zeros(n)
    opt = SeqikOptions()
    opt.device = device
    rc = load().seqik_signed_angles(v1.ctypes.data_as(_dp), 3 if v1.shape[0] == n and n > 1 else 0, v2.ctypes.data_as(_dp),
                                    3 if v2.shape[0] == n and n > 1 else 0, axis.ctypes.data_as(_dp), n,
                                    out.ctypes.data_as(_dp), ctypes.byref(opt))
    if rc != SEQIK_OK:
        _raise(rc)
    return out


def make_leg_params(leg, bounds_dof, body_size, initial_angles) -> SeqikLegParams:
    """Packs the reference's dict-shaped chain description of one leg into the ABI struct."""
    lp = SeqikLegParams()
    for i, seg in enumerate(SEGMENTS):
        lp.seg[i] = float(body_size[f"{leg}_{seg}"])
    for i, dof in enumerate(DOFS):
        lb, ub = bounds_dof[f"{leg}_{dof}"]
        lp.bounds[i][0] = float(lb)
        lp.bounds[i][1] = float(ub)
    seeds = np.concatenate([np.asarray(initial_angles[leg][f"stage_{k}"], dtype=np.float64).ravel()
                            for k in (1, 2, 3, 4)])
    if seeds.shape != (27,):
        raise ValueError(f"initial_angles[{leg!r}] must hold 4, 6, 8 and 9 values for stage_1..stage_4")
    for i in range(27):
        lp.seeds[i] = float(seeds[i])
    return lp


def leg_params_from_arrays(seg, bounds, seeds) -> SeqikLegParams:
    lp = SeqikLegParams()
    for i in range(4):
        lp.seg[i] = float(seg[i])
    for i in range(7):
        lp.bounds[i][0] = float(bounds[i][0])
        lp.bounds[i][1] = float(bounds[i][1])
    for i in range(27):
        lp.seeds[i] = float(seeds[i])
    return lp


def _raise(rc: int):
    msg = load().seqik_last_error().decode("utf-8", "replace")
    if rc in (ERR_X0, ERR_BOUNDS, ERR_STAGE):
        raise ValueError(msg)
    if rc == ERR_ARG:
        raise ValueError(f"seqik: bad argument: {msg}")
    raise SeqikLibraryError(f"seqik: HIP error: {msg}")


def validate_legs(legs, first_stage=1, last_stage=4):
    arr = (SeqikLegParams * len(legs))(*legs)
    rc = load().seqik_validate_legs(arr, len(legs), first_stage, last_stage)
    if rc != SEQIK_OK:
        _raise(rc)


def _check_finite(pose):
    """scipy refuses non-finite residuals at the start point (``ValueError: Residuals are not finite in
    the initial point.``); a NaN key point would do exactly that in the reference's frame loop."""
    if not np.isfinite(pose).all():
        raise ValueError("Residuals are not finite in the initial point.")


def _affine_array(affine, n_legs):
    if affine is None:
        return None
    if len(affine) != n_legs:
        raise ValueError("one SeqikAffine per leg expected")
    return (SeqikAffine * n_legs)(*affine)


def solve_seq(pose, legs, first_stage=1, last_stage=4, angles=None, want_fk=True, want_diag=False,
              device=-1, block_size=0, affine=None, init_angles=None, lanes_per_wave=0, staged=0, interleave_legs=0,
              frame_chunk=0, frame_halo=0, chunk_tol=0.0, chunk_rounds=0, pipeline=0, want_chunk_flags=False):
    """``seqik_solve_seq`` on host arrays.

    pose: (S, L, N, 5, 3) float64; legs: list of L ``SeqikLegParams``; angles: optional
    (S, L, N, 7) with earlier-stage columns filled when ``first_stage > 1``.
    ``affine``: optional list of L ``SeqikAffine`` -- ``pose`` then holds RAW key points and the
    alignment is fused into the kernels.  ``lanes_per_wave``: chains per wavefront (0 = automatic; 128, 192, ... 4096 = the chain
    queue of the single-launch kernel, see ``SeqikOptions.reserved[0]``); ``staged=1``: one launch per stage instead of the single fused launch.
    ``frame_chunk`` (0 = serial walk, bit-exact; -1 = automatic; > 0 = frames per chunk), ``frame_halo``,
    ``chunk_tol``, ``chunk_rounds``: frame chunks, see ``SeqikOptions`` in include/seqik.h -- one long recording
    solved in concurrently running pieces, equal to the serial walk to about ``chunk_tol`` (default 1e-6 rad).
    ``pipeline``: stage pipeline (``SeqikOptions.reserved[3]``): 0 = automatic (few chains), 1 = never, 2 = always.
    ``device``: HIP device ordinal, -1 = the calling thread's current device.
    ``want_chunk_flags``: also return the per-chunk report ``chunk_flags`` (S, L, K) uint8 (``CHUNK_FLAG_*`` bits: failed
    the first verification / repaired / swept / chain walked serially); None when the call was not chunked.
    Returns dict(angles, fk or None, status or None, nfev or None, chunk_stats, chunk_flags).
    """
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    if pose.ndim != 5 or pose.shape[3:] != (5, 3):
        raise ValueError(f"pose must have shape (S, L, N, 5, 3), got {pose.shape}")
    S, L, N = pose.shape[:3]
    if len(legs) != L:
        raise ValueError("one SeqikLegParams per leg expected")
    _check_finite(pose)
    if angles is None:
        angles = np.zeros((S, L, N, 7), dtype=np.float64)
    else:
        angles = np.array(angles, dtype=np.float64, order="C", copy=True)
        if angles.shape != (S, L, N, 7):
            raise ValueError(f"angles must have shape {(S, L, N, 7)}")
    fk = np.full((S, L, N, 9, 3), np.nan) if (want_fk and last_stage == 4) else None
    status = np.full((S, L, N, 4), -1, dtype=np.int32) if want_diag else None
    nfev = np.zeros((S, L, N, 4), dtype=np.int32) if want_diag else None
    arr = (SeqikLegParams * L)(*legs)
    opt = SeqikOptions()
    opt.device = device
    opt.block_size = block_size
    opt.reserved[0] = lanes_per_wave
    opt.reserved[1] = staged
    opt.reserved[2] = interleave_legs
    opt.reserved[3] = pipeline
    opt.frame_chunk, opt.frame_halo, opt.chunk_tol, opt.chunk_rounds = frame_chunk, frame_halo, chunk_tol, chunk_rounds
    stats = np.zeros(N_CHUNK_STATS, dtype=np.int32)
    opt.chunk_stats = stats.ctypes.data_as(_ip)
    if init_angles is not None:
        init_angles = np.ascontiguousarray(init_angles, dtype=np.float64)
        if init_angles.shape != (S, L, 7):
            raise ValueError(f"init_angles must have shape {(S, L, 7)}")
    lib = load()
    flags = None
    if want_chunk_flags and frame_chunk != 0 and first_stage == 1 and last_stage == 4 and not want_diag:
        k = frame_chunk_plan(N, frame_chunk, frame_halo)[2]
        if k > 0:
            flags = np.zeros((S, L, k), dtype=np.uint8)
            opt.chunk_flags = flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    rc = lib.seqik_solve_seq(pose.ctypes.data_as(_dp), S, L, N, arr, first_stage, last_stage,
                             angles.ctypes.data_as(_dp),
                             fk.ctypes.data_as(_dp) if fk is not None else None,
                             status.ctypes.data_as(_ip) if status is not None else None,
                             nfev.ctypes.data_as(_ip) if nfev is not None else None,
                             init_angles.ctypes.data_as(_dp) if init_angles is not None else None,
                             _affine_array(affine, L), ctypes.byref(opt))
    if rc != SEQIK_OK:
        _raise(rc)
    return dict(angles=angles, fk=fk, status=status, nfev=nfev, chunk_stats=chunk_stats_dict(stats), chunk_flags=flags)


def solve_seq_device(d_pose, n_seq, n_legs, n_frames, legs, d_angles, d_fk=0, d_status=0, d_nfev=0,
                     first_stage=1, last_stage=4, stream=0, block_size=0, layout=None, affine=None, d_init=0,
                     stage_events=None, lanes_per_wave=0, staged=0, interleave_legs=0,
                     frame_chunk=0, frame_halo=0, chunk_tol=0.0, chunk_rounds=0, d_chunk_stats=0, pipeline=0,
                     frame_lead=0, d_chunk_flags=0, d_chunk_states=0, chunk_resume=0):
    """``seqik_solve_seq_device``: raw device pointers (ints), asynchronous on ``stream``.
    ``layout``: a ``SeqikLayout`` (``planar_layout(n_frames)``) or None for the dense layout.
    ``stage_events``: optional 5 raw hipEvent_t handles (e.g. ``torch.cuda.Event(...).cuda_event`` after a
    first ``record()``), recorded in front of each stage kernel and behind the last one."""
    arr = (SeqikLegParams * n_legs)(*legs)
    opt = SeqikOptions()
    opt.block_size = block_size
    opt.reserved[0] = lanes_per_wave
    opt.reserved[1] = staged
    opt.reserved[2] = interleave_legs
    opt.reserved[3] = pipeline
    opt.frame_chunk, opt.frame_halo, opt.chunk_tol, opt.chunk_rounds = frame_chunk, frame_halo, chunk_tol, chunk_rounds
    opt.frame_lead, opt.chunk_resume = int(frame_lead), int(chunk_resume)
    if d_chunk_stats:  # device int32[16]
        opt.chunk_stats = ctypes.cast(ctypes.c_void_p(int(d_chunk_stats)), _ip)
    if d_chunk_flags:  # device uint8 [n_seq][n_legs][K]
        opt.chunk_flags = ctypes.cast(ctypes.c_void_p(int(d_chunk_flags)), ctypes.POINTER(ctypes.c_uint8))
    if d_chunk_states:  # device float64 [n_seq][n_legs][K][7]
        opt.chunk_states = ctypes.cast(ctypes.c_void_p(int(d_chunk_states)), _dp)
    if stage_events is not None:
        ev = (ctypes.c_void_p * 5)(*[ctypes.c_void_p(int(e)) for e in stage_events])
        opt.stage_events = ctypes.cast(ev, ctypes.POINTER(ctypes.c_void_p))
    rc = load().seqik_solve_seq_device(ctypes.c_void_p(d_pose), n_seq, n_legs, n_frames, arr, first_stage,
                                       last_stage, ctypes.c_void_p(d_angles), ctypes.c_void_p(d_fk or None),
                                       ctypes.c_void_p(d_status or None), ctypes.c_void_p(d_nfev or None),
                                       ctypes.c_void_p(d_init or None),
                                       ctypes.byref(layout) if layout is not None else None,
                                       _affine_array(affine, n_legs),
                                       ctypes.byref(opt), ctypes.c_void_p(stream or None))
    if rc != SEQIK_OK:
        _raise(rc)
