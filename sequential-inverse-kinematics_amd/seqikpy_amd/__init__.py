"""seqikpy_amd -- MI355X-native drop-in for the leg-IK path of ``seqikpy``.

Same module / class names as the reference for this path:

    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES

plus, beyond the reference: ``batch.run_ik_and_fk_many`` (many recordings in one launch),
``streaming.SeqikStream`` (slabs from pinned host memory), verified frame chunks that let ONE recording fill the GPU (the
default of ``run_ik_and_fk`` since 0.5: ``frame_parallel="auto"``; ``frame_parallel=False`` is the reference's frame-by-frame
walk, bit-identical to the C restatement), ``frame_sharding`` (one recording over the GPUs of a node).

All arithmetic runs in ``csrc/libseqik_hip.so`` (hand-written HIP for gfx950) behind the C ABI
of ``include/seqik.h``; there is no CPU fallback.
"""
__version__ = "0.5.0"

import os as _os


def recommended_env(steps_in_flight=3, process_group=True):
    """Environment a process should START with to get what `bench.py` measures -- returned, never applied here: a library
    that edits ``os.environ`` when it is imported changes the behaviour of everything else in the process, and only
    works when it happens to be imported before the HIP runtime starts.

    ``GPU_MAX_HW_QUEUES``: streams of a process share that many hardware queues (HIP default: 4) and streams on one queue
    run one after the other.  `steps_in_flight` solver calls on streams of their own, the streaming pipeline's copy
    streams and -- `process_group` -- RCCL's stream each want one: 8 covers three calls in flight next to a communicator
    (measured 4.1e8 against 3.0e8 leg-frames/s with the default 4); the per-GPU shares of a strong-scaling job keep up to
    20 calls in flight and want 22, and no more (a process that HOLDS 24 queues pays ~10 % on its long kernels;
    DESIGN.md 5).  ``HSA_ENABLE_IPC_MODE_LEGACY=0``: dmabuf IPC, which `hipIpc*` handles (``peer_gather``) and RCCL need
    across processes on this driver.

    Use it before anything touches the GPU::

        os.environ.update(seqikpy_amd.recommended_env())        # or export the same in the job script

    or set ``SEQIK_SET_ENV=1`` to have the import do exactly that (values already set win)."""
    queues = min(22, max(8, int(steps_in_flight) + 2 + (1 if process_group else 0)))
    return {"GPU_MAX_HW_QUEUES": str(queues), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def runtime_env():
    """The variables of `recommended_env` (and any other GPU_* / HSA_* / HIP_* override) as this process has them: what a
    measurement should record next to its numbers (`bench.py` puts it into ``config.env``)."""
    keys = sorted(k for k in _os.environ if k.startswith(("GPU_", "HSA_", "HIP_", "ROCR_", "SEQIK_", "NCCL_", "RCCL_")))
    return {k: _os.environ[k] for k in keys}


if _os.environ.get("SEQIK_SET_ENV") == "1":       # opt-in: the import applies recommended_env(), explicit settings win
    for _k, _v in recommended_env().items():
        _os.environ.setdefault(_k, _v)

from . import data, utils  # noqa: F401
from .kinematic_chain import KinematicChainGeneric, KinematicChainSeq  # noqa: F401
from .leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq  # noqa: F401
