"""The final joint-angle gather as peer WRITES over xGMI, same interface as ``sharding.GatherPipeline``.

Rank ``dst`` owns, per peer, one exported device allocation holding ``n_buffers`` angle blocks
(``seqik_peer_alloc`` / ``seqik_peer_export``); every other rank maps its allocation once (``seqik_peer_open``) and,
per step, pushes its block with ONE ``hipMemcpyAsync`` on a copy stream (``seqik_peer_copy``): a copy-engine
transfer over the direct xGMI link -- no kernel, no compute unit on either GPU.  That matters on the root: its
solver launches keep every SIMD slot busy, and RCCL receive kernels moving 7 x 336 MB per step through the same
compute units cost it 15-30 % of its step (``scripts/contention/gather_contention.py``, one-GPU rehearsal).

Completion and slot reuse ride on one 8-byte all-reduce per step (``torch.distributed``; RCCL on GPUs), enqueued
behind the copy on every rank and chained: copy i waits for all-reduce i - 1.  Hence
  * all-reduce i complete  =>  every rank's block of step i has landed in the root's buffers (``wait_buffer``);
  * a peer overwrites slot b (step i + n_buffers) only after the root has enqueued all-reduce i + n_buffers - 1, so
    the root may read ``recv[b]`` from ``wait_buffer(b)`` until its submit BEFORE the one that reuses b.

``make_gather`` picks this path when the process group runs on RCCL, verifies it end to end (pattern written by
every peer, read back on the root, bandwidth measured) and falls back to ``GatherPipeline`` (grouped RCCL
point-to-point) on every rank alike if any rank could not map the root's memory, the data did not arrive, or the
link is slower than the step needs.

torch is used for streams, events and the process group only.
"""
import os
import time
from typing import List, Optional

from . import _lib
from .sharding import GatherPipeline


class PeerWriteGather:
    name = "peer writes: one copy-engine transfer per rank and step into rank 0's exported buffers (xGMI), " \
           "8-byte all-reduce as completion flag"

    def __init__(self, dist, world: int, rank: int, like, dst: int = 0, n_buffers: int = 2):
        import torch
        self.dist, self.world, self.rank, self.dst, self.n_buffers = dist, world, rank, dst, n_buffers
        self.on_rccl = dist.get_backend() == "nccl"
        self.block_bytes = like.numel() * like.element_size()
        self.shape = tuple(like.shape)
        self.copy_stream = torch.cuda.Stream()
        self.work: List[Optional[object]] = [None] * n_buffers
        self.prev_work = None
        self.flags = [torch.zeros(1, dtype=torch.float64, device="cuda" if self.on_rccl else "cpu")
                      for _ in range(n_buffers)]
        self.owned, self.remote, self.recv = {}, None, None
        self.ok = True
        handles = None
        if rank == dst:
            try:
                for r in range(world):
                    if r != dst:
                        self.owned[r] = _lib.PeerBuffer(n_buffers * self.block_bytes)
                handles = {r: buf.handle() for r, buf in self.owned.items()}
            except Exception as exc:  # noqa: BLE001  (reported to every rank below)
                self.error, handles, self.ok = repr(exc), None, False
        box = [handles]
        dist.broadcast_object_list(box, src=dst)   # collective: reached by every rank whatever happened above
        handles = box[0]
        if handles is None:
            self.ok = False
        elif rank != dst:
            try:
                self.remote = _lib.PeerBuffer.open(handles[rank], n_buffers * self.block_bytes)
            except Exception as exc:  # noqa: BLE001
                self.error, self.ok = repr(exc), False
        else:
            views = {r: buf.tensor((n_buffers,) + self.shape) for r, buf in self.owned.items()}
            self.recv = [[None if r == dst else views[r][b] for r in range(world)] for b in range(n_buffers)]
        self.ok = self._all_agree(self.ok)

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _all_agree(self, ok: bool, value: float = 0.0):
        """MIN over ranks of (ok, value); returns ok when value is not asked for."""
        import torch
        t = torch.tensor([1.0 if ok else 0.0, value], dtype=torch.float64, device="cuda" if self.on_rccl else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t[0].item() > 0.5)

    def _flag_all_reduce(self, b: int):
        """The completion flag of buffer b: ordered behind everything queued on the copy stream."""
        import torch
        if self.on_rccl:
            with torch.cuda.stream(self.copy_stream):
                return self.dist.all_reduce(self.flags[b], async_op=True)
        self.copy_stream.synchronize()   # gloo (tests / dry runs): host-side ordering
        return self.dist.all_reduce(self.flags[b], async_op=True)

    # ---- GatherPipeline interface ----------------------------------------------------------------------------
    def wait_buffer(self, b: int):
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None

    def submit(self, b: int, tensor):
        import torch
        self.wait_buffer(b)
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() * tensor.element_size() == self.block_bytes
        self.copy_stream.wait_stream(torch.cuda.current_stream())   # the solver launch that produced `tensor`
        if self.prev_work is not None:                              # chain: copy i after all-reduce i - 1
            if self.on_rccl:
                with torch.cuda.stream(self.copy_stream):
                    self.prev_work.wait()
            else:
                self.prev_work.wait()
        if self.rank != self.dst:
            _lib.peer_copy(self.remote.ptr + b * self.block_bytes, tensor.data_ptr(), self.block_bytes,
                           stream=self.copy_stream.cuda_stream)
        else:
            self.recv[b][self.dst] = tensor
        self.work[b] = self.prev_work = self._flag_all_reduce(b)

    def drain(self):
        for b in range(self.n_buffers):
            self.wait_buffer(b)
        self.copy_stream.synchronize()

    def close(self):
        import torch
        torch.cuda.synchronize()
        if self.remote is not None:
            self.remote.close()
            self.remote = None
        self.dist.barrier()            # every mapping is gone before the owner frees
        self.recv = None
        for buf in self.owned.values():
            buf.close()
        self.owned = {}

    # ---- end-to-end check ------------------------------------------------------------------------------------
    def probe(self, repeats: int = 3):
        """Every peer writes a pattern into slot 0 `repeats` times (all peers at once); the root checks what
        arrived.  Returns (ok on every rank, slowest peer's GB/s)."""
        import torch
        src = torch.full(self.shape, float(self.rank + 1), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        self.dist.barrier()
        gbps = float("inf")
        ok = True
        try:
            if self.rank != self.dst:
                t0 = time.perf_counter()
                for _ in range(repeats):
                    _lib.peer_copy(self.remote.ptr, src.data_ptr(), self.block_bytes, stream=self.copy_stream.cuda_stream)
                self.copy_stream.synchronize()
                gbps = repeats * self.block_bytes / (time.perf_counter() - t0) / 1e9
        except Exception as exc:  # noqa: BLE001
            self.error, ok = repr(exc), False
        self.dist.barrier()
        if self.rank == self.dst and ok:
            torch.cuda.synchronize()
            for r in range(self.world):
                if r != self.dst and not bool((self.recv[0][r] == float(r + 1)).all().item()):
                    ok = False
        import torch as _t
        t = _t.tensor([1.0 if ok else 0.0, gbps if gbps != float("inf") else 1e9], dtype=_t.float64,
                      device="cuda" if self.on_rccl else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t[0].item() > 0.5), float(t[1].item())


def make_gather(dist, world: int, rank: int, like, n_buffers: int, dst: int = 0, min_gbps: float = 0.0,
                prefer: Optional[str] = None):
    """The gather pipeline for this process group: ``PeerWriteGather`` when it works on every rank (and each link
    delivers at least ``min_gbps``), else ``GatherPipeline``.  ``prefer``: "peer" / "rccl" / None = the environment
    variable SEQIK_GATHER or "auto" (peer writes for device buffers, point-to-point of the process group otherwise).
    Returns (pipeline, description).

    COLLECTIVE: every rank of the group must enter it (the shape check below and the set-up of the peer path are
    collectives).  A caller whose ranks may fail on their own before this point -- allocating `like`, say -- reaches consensus on
    that first (an all-reduce(MIN) of "my set-up worked": `bench_support.Ranks.all_ok`) and only then calls make_gather on all
    ranks or on none; a rank that skips it while the others wait inside would hang the job."""
    prefer = prefer or os.environ.get("SEQIK_GATHER", "auto")
    # both pipelines move EQUAL blocks (one receive buffer shape / one slot size for every peer): a rank with another block
    # shape would make the point-to-point sizes disagree -- a hang on RCCL -- so that is refused here, on every rank alike
    if world > 1:
        shapes = [None] * world
        dist.all_gather_object(shapes, (tuple(like.shape), str(like.dtype)))
        if any(sh != shapes[0] for sh in shapes):
            raise ValueError(f"make_gather: the ranks' angle blocks differ in shape ({shapes}); pad them to the largest share")
    # (gloo process groups over device buffers are the one-GPU rehearsal: the peer path works there too -- the flags
    # then travel over gloo -- and RCCL's stand-in would stage every block through the host)
    want_peer = prefer == "peer" or (prefer == "auto" and like.is_cuda)
    if want_peer and world > 1:
        pg = PeerWriteGather(dist, world, rank, like, dst=dst, n_buffers=n_buffers)
        if pg.ok:
            ok, gbps = pg.probe()
            if ok and gbps >= min_gbps:
                return pg, f"{PeerWriteGather.name}; probe: slowest link {gbps:.1f} GB/s with all peers writing"
            why = f"probe failed or too slow ({gbps:.1f} GB/s < {min_gbps:.1f})"
        else:
            why = "a rank could not export / map the root's buffers"
        pg.close()
        return (GatherPipeline(dist, world, rank, like, dst=dst, n_buffers=n_buffers),
                f"grouped RCCL point-to-point (peer writes unavailable: {why})")
    return GatherPipeline(dist, world, rank, like, dst=dst, n_buffers=n_buffers), "grouped RCCL point-to-point"
