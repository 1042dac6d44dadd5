"""Slab streaming over host buffers (BASELINE config 5; C ABI ``seqik_stream_*`` in include/seqik.h).

The reference processes a recording with one ``AlignPose.align_pose()`` +
``LegInvKinSeq.run_ik_and_fk()`` call (``seqikpy/alignment.py:345``,
``seqikpy/leg_inverse_kinematics.py:324``); everything has to be in memory at once.  Here recordings
are pushed through the MI355X in *slabs* of ``(n_seq, n_legs, n_frames)`` key points living in pinned
host memory: upload, the four stage kernels and download of consecutive slabs overlap on three HIP
streams.  With ``carry=True`` consecutive slabs are consecutive pieces in time of the same
recordings (frame 0 of a slab is warm-started from the last frame of the slab before, on the
device), so the result equals one call over the concatenated recording bit for bit.

All work is done by ``libseqik_hip.so``; this module only owns buffers and the handle.
"""
import ctypes
from typing import List, Optional

import numpy as np

from . import _lib


def pinned_array(shape) -> np.ndarray:
    """A float64 numpy array over pinned (page-locked) host memory from ``seqik_host_alloc``.  The memory belongs to
    the array: it is returned to the driver when the array AND every view of it have been garbage collected (a
    finalizer on the buffer object all of them keep alive), so an expression like ``pinned_array(shape)[:n]`` or an
    array handed to ``SeqikStream.submit`` can never outlive its memory."""
    import weakref
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape))
    lib = _lib.load()
    ptr = lib.seqik_host_alloc(max(n, 1) * 8)
    if not ptr:
        _lib._raise(_lib.ERR_HIP)
    buf = (ctypes.c_double * max(n, 1)).from_address(ptr)
    weakref.finalize(buf, lib.seqik_host_free, ctypes.c_void_p(ptr))
    return np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)


class PinnedArray:
    """Holder of a ``pinned_array`` (kept for callers that want an explicit handle).  ``.array`` is safe to keep after
    the holder is gone; ``free()`` drops the holder's reference only -- the memory goes when the last view does."""

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)
        self.array = pinned_array(self.shape)

    def free(self):
        self.array = None


class SeqikStream:
    """``with SeqikStream(legs, slab_seq, n_frames, ...) as st: st.submit(pose, angles, fk); ...; st.wait()``.

    legs: list of ``SeqikLegParams``; affine: optional list of ``SeqikAffine`` (pose slabs are then RAW
    key points, ``AlignPose.align_leg`` runs in the kernel prologue); layout: optional ``SeqikLayout`` of
    the pose / angle slabs (``_lib.planar_layout(n_frames)``), default dense reference-shaped arrays.
    """

    def __init__(self, legs: List[_lib.SeqikLegParams], slab_seq: int, n_frames: int, affine=None, layout=None,
                 want_fk: bool = True, n_slots: int = 3, carry: bool = False, generic: bool = False,
                 device: int = -1, block_size: int = 0, frame_chunk: int = 0, frame_halo: int = 0,
                 chunk_tol: float = 0.0):
        self._lib = _lib.load()
        self.n_legs = len(legs)
        self.slab_seq, self.n_frames = int(slab_seq), int(n_frames)
        self.want_fk, self.layout = bool(want_fk), layout
        self._handle = ctypes.c_void_p()
        opt = _lib.SeqikOptions()
        opt.device = device
        opt.block_size = block_size
        # frame chunks inside every slab (SeqikOptions.frame_chunk): with carry=True and ONE long recording per slab
        # (slab_seq = 1) this is BASELINE config 5 read literally -- a 10 M-frame recording streamed in time slabs, each
        # slab cut into chunks on the device, its first chunk warm-started from the carried last frame of the slab before
        opt.frame_chunk, opt.frame_halo, opt.chunk_tol = int(frame_chunk), int(frame_halo), float(chunk_tol)
        rc = self._lib.seqik_stream_open(ctypes.byref(self._handle), self.n_legs,
                                         (_lib.SeqikLegParams * self.n_legs)(*legs),
                                         _lib._affine_array(affine, self.n_legs), self.slab_seq, self.n_frames,
                                         ctypes.byref(layout) if layout is not None else None,
                                         1 if want_fk else 0, int(n_slots), 1 if carry else 0, 1 if generic else 0,
                                         ctypes.byref(opt))
        if rc != _lib.SEQIK_OK:
            self._handle = ctypes.c_void_p()
            _lib._raise(rc)
        self._keep = []  # host arrays of slabs in flight (kept alive until wait())

    # shapes of one slab's host arrays (dense layout)
    def pose_shape(self, n_seq=None):
        return (n_seq or self.slab_seq, self.n_legs, self.n_frames, 5, 3)

    def angles_shape(self, n_seq=None):
        return (n_seq or self.slab_seq, self.n_legs, self.n_frames, 7)

    def fk_shape(self, n_seq=None):
        return (n_seq or self.slab_seq, self.n_legs, self.n_frames, 9, 3)

    def submit(self, pose: np.ndarray, angles: np.ndarray, fk: Optional[np.ndarray] = None):
        """Queues one slab.  Arrays must be C-contiguous float64 and stay untouched until ``wait()``."""
        for name, a in (("pose", pose), ("angles", angles), ("fk", fk)):
            if a is not None and (a.dtype != np.float64 or not a.flags.c_contiguous):
                raise ValueError(f"{name} must be a C-contiguous float64 array")
        per_seq = self.n_legs * self.n_frames
        if self.layout is not None:
            # custom layout: a slab is the contiguous block of n_seq * n_legs chain strides (include/seqik.h)
            pc, ac = int(self.layout.pose_chain) * self.n_legs, int(self.layout.ang_chain) * self.n_legs
            n_seq = pose.size // pc
            if n_seq * pc != pose.size or angles.size != n_seq * ac:
                raise ValueError("pose / angles do not hold n_seq x n_legs chain strides of the stream's layout")
        else:
            n_seq = pose.shape[0]
            if pose.size != n_seq * per_seq * 15 or angles.size != n_seq * per_seq * 7:
                raise ValueError("pose / angles do not hold n_seq x n_legs x n_frames leg-frames")
        if self.want_fk and (fk is None or fk.size != n_seq * per_seq * 27):
            raise ValueError("fk must hold n_seq x n_legs x n_frames x 9 x 3 values")
        rc = self._lib.seqik_stream_submit(self._handle, ctypes.c_void_p(pose.ctypes.data), n_seq,
                                           ctypes.c_void_p(angles.ctypes.data),
                                           ctypes.c_void_p(fk.ctypes.data) if (self.want_fk and fk is not None) else None)
        if rc != _lib.SEQIK_OK:
            _lib._raise(rc)
        self._keep.append((pose, angles, fk))

    def wait(self):
        rc = self._lib.seqik_stream_wait(self._handle)
        self._keep.clear()
        if rc != _lib.SEQIK_OK:
            _lib._raise(rc)

    def reset_carry(self):
        self._lib.seqik_stream_reset_carry(self._handle)

    def set_carry(self, state, on_device: bool = False):
        """Carried streams: the next slab continues from ``state`` (n_seq, n_legs, 7) -- a numpy array, or (with
        ``on_device``) a raw device pointer / an object with ``data_ptr()`` on the stream's GPU together with
        ``n_seq`` as ``state = (ptr, n_seq)``."""
        if on_device:
            ptr, n_seq = state
            ptr = int(ptr.data_ptr()) if hasattr(ptr, "data_ptr") else int(ptr)
            rc = self._lib.seqik_stream_set_carry(self._handle, ctypes.c_void_p(ptr), int(n_seq), 1)
        else:
            arr = np.ascontiguousarray(state, dtype=np.float64)
            if arr.ndim != 3 or arr.shape[1:] != (self.n_legs, 7):
                raise ValueError(f"state must have shape (n_seq, {self.n_legs}, 7)")
            rc = self._lib.seqik_stream_set_carry(self._handle, ctypes.c_void_p(arr.ctypes.data), arr.shape[0], 0)
        if rc != _lib.SEQIK_OK:
            _lib._raise(rc)

    def close(self):
        if self._handle:
            self._lib.seqik_stream_close(self._handle)
            self._handle = ctypes.c_void_p()
            self._keep.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_streamed(pose: np.ndarray, legs, slab_seq: int, affine=None, want_fk: bool = True, n_slots: int = 3,
                   device: int = -1):
    """Convenience: ``pose (S, L, N, 5, 3)`` pushed through a stream in slabs of ``slab_seq`` sequences.
    Returns ``dict(angles (S, L, N, 7), fk (S, L, N, 9, 3) or None)`` -- equal to ``_lib.solve_seq`` bit for bit."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    _lib._check_finite(pose)
    S, L, N = pose.shape[:3]
    angles = np.zeros((S, L, N, 7))
    fk = np.full((S, L, N, 9, 3), np.nan) if want_fk else None
    with SeqikStream(legs, min(slab_seq, max(S, 1)), N, affine=affine, want_fk=want_fk, n_slots=n_slots,
                     device=device) as st:
        for s0 in range(0, S, slab_seq):
            s1 = min(S, s0 + slab_seq)
            st.submit(pose[s0:s1], angles[s0:s1], fk[s0:s1] if want_fk else None)
        st.wait()
    return dict(angles=angles, fk=fk)


def solve_streamed_in_time(pose: np.ndarray, legs, slab_frames: int, affine=None, want_fk: bool = True,
                           n_slots: int = 3, device: int = -1, frame_chunk: int = 0):
    """``pose (S, L, N, 5, 3)`` pushed through a carried stream in slabs of ``slab_frames`` frames (the S
    recordings advance in lock step).  Equal to ``_lib.solve_seq(pose, ...)`` bit for bit with ``frame_chunk = 0``;
    with frame chunks (-1 = automatic) every slab is cut into concurrently solved chunks on the device (a few long
    recordings then fill the GPU) and the result equals the serial walk to about the chunk tolerance."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    _lib._check_finite(pose)
    S, L, N = pose.shape[:3]
    angles = np.zeros((S, L, N, 7))
    fk = np.full((S, L, N, 9, 3), np.nan) if want_fk else None
    T = min(slab_frames, N)
    n_full = N // T if T else 0
    if n_full:
        with SeqikStream(legs, S, T, affine=affine, want_fk=want_fk, n_slots=n_slots, carry=True, device=device,
                         frame_chunk=frame_chunk) as st:
            bufs = []
            for k in range(n_full):
                sl = slice(k * T, (k + 1) * T)
                a = np.empty((S, L, T, 7))
                f = np.empty((S, L, T, 9, 3)) if want_fk else None
                st.submit(np.ascontiguousarray(pose[:, :, sl]), a, f)
                bufs.append((sl, a, f))
            st.wait()
        for sl, a, f in bufs:
            angles[:, :, sl] = a
            if want_fk:
                fk[:, :, sl] = f
    t0 = n_full * T
    if t0 < N:  # a shorter last piece: one direct call, continued from the last streamed frame
        rest = _lib.solve_seq(pose[:, :, t0:], legs, want_fk=want_fk, affine=affine, device=device,
                              init_angles=np.ascontiguousarray(angles[:, :, t0 - 1]) if t0 else None,
                              frame_chunk=frame_chunk)
        angles[:, :, t0:] = rest["angles"]
        if want_fk:
            fk[:, :, t0:] = rest["fk"]
    return dict(angles=angles, fk=fk)
