"""seqikpy_amd -- MI355X-native drop-in for the leg-IK path of ``seqikpy``.

Same module / class names as the reference for this path:

    from seqikpy_amd.kinematic_chain import KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinSeq
    from seqikpy_amd.data import BOUNDS, INITIAL_ANGLES

plus, beyond the reference: ``batch.run_ik_and_fk_many`` (many recordings in one launch),
``streaming.SeqikStream`` (slabs from pinned host memory), ``run_ik_and_fk(frame_parallel="auto")`` (one long recording
in verified frame chunks), ``frame_sharding`` (one recording over the GPUs of a node).

All arithmetic runs in ``csrc/libseqik_hip.so`` (hand-written HIP for gfx950) behind the C ABI
of ``include/seqik.h``; there is no CPU fallback.
"""
__version__ = "0.4.0"

import os as _os

# Streams of a process share GPU_MAX_HW_QUEUES hardware queues (HIP default: 4) and streams on one queue serialise;
# the streaming pipeline (upload, 2 x compute, download) next to the caller's own streams or RCCL needs more
# (measured: three solver streams + an RCCL communicator 3.0e8 solves/s with 4 queues, 4.1e8 with 8).  Only
# effective when this package is imported before the HIP runtime starts; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import data, utils  # noqa: F401
from .kinematic_chain import KinematicChainGeneric, KinematicChainSeq  # noqa: F401
from .leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq  # noqa: F401
