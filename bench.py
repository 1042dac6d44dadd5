#!/usr/bin/env python3
"""Benchmark of the sequential leg-IK hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1: either started by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one rank per GPU),
or plainly as `python bench.py --gpus N ...`, in which case this process only LAUNCHES that command as a child, relays
rank 0's JSON line and exits with the child's code (launch_ranks_if_needed below; it never touches the GPU).

Metric (BASELINE.json): leg-IK solves/s, one solve = one (frame, leg) = 4 stage sub-solves ->
7 joint angles (+ the stage-4 forward kinematics); and max |d theta| vs the reference (`parity`).

Workload: BASELINE config 3, "synthetic 1M frames x 6 legs, random in-workspace target key points, 1 -> 8 MI355X
frame-sharded": ONE fixed problem.
  --scaling strong (default) config 3 literally: 1,000,000 frames IN TOTAL, cut into 15,625 independent sequences of 64
                   frames (frame t of a sequence is warm-started from frame t-1, frame 0 from the seeds -- the reference's
                   semantics applied to many recordings), 6 legs each = 93,750 chains; rank r solves sequences
                   [r S/N, (r+1) S/N).  At N = 1 strong and weak are the same run.
  --scaling weak   1,000,000 frames PER GPU (a named leg of the N > 1 line either way: `multi_gpu.weak`)
A step is one pass of the hot path (one launch in which every wave takes its chains through stages 1-4;
`--staged`: the 4 stage kernels) over the rank's batch with inputs resident in HBM; consecutive steps overlap on
`--streams` HIP streams -- by default (`--streams 0`) as many as a calibration over the very region that is about to be
measured finds best (`config.depth_calibration`; depth_candidates(): up to 16 steps in flight on the lane-per-chain
kernels -- a 1/8 share of the problem only fills the GPU that way --, the last partial round on the library's own kernel
choice).  For N > 1 every step also sends the rank's joint angles to rank 0 (copy-engine peer writes
over xGMI into rank 0's IPC-exported buffers, an 8-byte RCCL all-reduce as completion flag; grouped RCCL
point-to-point if the peer path is unavailable, or when SEQIK_GATHER=rccl; `config.gather` says which ran).

Prints ONE JSON line on rank 0: `value` (pipeline throughput of the timed region), `roofline` (dominant kernel, live
HIP-event timing; the bound that matters here is FP64 VALU issue, the HBM figures are kept beside it) and, at N = 1,
  single_job         the same batch with ONE launch in flight at a time (no overlap between steps)
  variants           the other synthetic variant (smooth <-> iid), same pipeline
  single_recording   ONE recording of 1M frames x 6 legs (real locomotion poses repeated), walked as the reference
                     walks a recording, by frame chunks (SeqikOptions.frame_chunk)
  strong_projection  the per-rank share of the fixed 1M-frame problem at N = 2, 4, 8, timed on this GPU over the same region
                     (same steps / warm-up) at every candidate depth, + one job at a time, + the issue floor of the 1/8 share
  parity             HIP vs the committed reference fixtures (shipped anipose outputs, df3d reference-source run):
                     max |d theta|, leg-frames over 1e-4 rad and where, for the serial walk and for frame chunks
  cpu_baseline       the C oracle on the host cores, bounded sample of the same workload (+ Python/scipy pool)
  value_single_job   = single_job.value, first class: what ONE 1M-frame x 6-leg job gets (no second batch to overlap with)
  configs            every BASELINE.json config in this one driver-timed line: 1 / 2 / 4 through the reference-shaped
                     Python API (default serial walk AND frame_parallel="auto": ms, leg-frames/s, max |d theta| vs the
                     fixture, chunk statistics, latency_floor_frac), 4 with the head / antenna angles in the same
                     submission, 3 = the headline, 5 streamed from pinned host slabs with the alignment fused
                     (PCIe-inclusive, checked), the generic chain on the shipped 6000-frame recording and in batches
                     (static launch against the chain queue)
  <scalars>          a handful of the figures above again as top-level scalars (config1_default_ms, ...,
                     strong_projected_speedup_n8, head_kernel_frac_of_box_copy): they survive in the driver's own record
and, at N > 1, `multi_gpu` (every leg with its per-rank ms, the gather it used and `efficiency_vs_n1`):
  ranks_seen, rank_ms_per_step   who took part (rank, host, device from the process group) and how even the ranks were
  n1_reference       the WHOLE fixed problem on rank 0's GPU alone, same run, same pipeline (the other ranks wait): what
                     `efficiency_vs_n1` = value / (N x that) of the headline and of the weak leg is measured against
  gather_compare     the same batch with the angle gather as peer writes, as grouped RCCL point-to-point, and without
  weak | strong      the other scaling mode, a short run beside the headline
  one_recording      config 3 as ONE recording of 1M frames, truly frame-sharded (contiguous frame slabs, boundary repair),
                     with its own one-GPU reference (`n1_reference_ms`)
  config5            BASELINE config 5 on the N GPUs: 10M frames x 6 legs streamed from pinned host memory with the alignment
                     fused (PCIe-inclusive): independent sequences split over the ranks, and ONE recording in contiguous slabs
                     per rank with the warm start carried across slabs and ranks
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share a queue run one after the other.  Three solver streams fit -- until RCCL adds its own: measured on one MI355X,
# 4.13e8 -> 3.0e8 solves/s as soon as a process group exists (same as GPU_MAX_HW_QUEUES=2 without one); with 8
# queues both cases run at 4.13e8.  Must be set before the runtime initialises.
# Round 5: the per-rank shares of the fixed problem (strong scaling) only fill the GPU with MANY steps in flight -- 16-20 streams,
# each on a hardware queue of its own: 1/8 share 3.4 -> 2.3 (16 in flight) -> 2.1 ms per step (20 in flight)
# (profiles/r05_share_streams.jsonl, r05_stream_cliff.jsonl).  How many queues: torch hands out streams from a pool of 32, HIP
# maps them onto at most GPU_MAX_HW_QUEUES hardware queues, and a process that HOLDS 24 of them pays ~10 % on its other
# long-running kernels (one-chain generic call 1.67 -> 1.86 s, config-5 stream 1.84e8 -> 1.65e8; up to 22: nothing --
# profiles/r05_queue_count_probe_touched.jsonl, r05_queues_ab_bench_legs.jsonl, r05_queues_22_depth20_bench_legs.jsonl).  So:
# 22 queues, at most 20 steps in flight.
MAX_DEPTH = int(os.environ.get("SEQIK_BENCH_MAX_DEPTH") or 20)   # steps in flight at most (24 streams of a 1/8 share lose again)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and hipIpc handles across processes need it

ROOT = os.path.dirname(os.path.abspath(__file__))


def launch_ranks_if_needed(argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process becomes the LAUNCHER.  It starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child (one rank per GPU; the
    shape of the reference's own parallel script, which builds its pool and merges the results itself:
    examples/example_leg_inv_kinematics_parallel.py:163-198), relays rank 0's single JSON line and exits with the
    child's code -- non-zero when any rank failed, 124 on time-out, 3 when no JSON line came back.  It runs BEFORE torch
    is imported and never touches the GPU or the HIP library (a process that has initialised the GPU must not be
    replaced or forked into ranks)."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ or "LOCAL_RANK" in os.environ:
        return  # already a rank of a torch.distributed.run job
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import signal
    import socket
    import subprocess
    with socket.socket() as sock:  # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    limit = float(os.environ.get("SEQIK_BENCH_TIMEOUT", "1500"))
    sys.stderr.write("bench.py launcher: " + " ".join(cmd) + "\n")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=limit)
        rc = child.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)  # the child's own session: torchrun and every rank
        except ProcessLookupError:
            pass
        out, _ = child.communicate()
        sys.stderr.write(f"bench.py launcher: no result after {limit:.0f} s, ranks killed\n")
        rc = 124
    lines = [l for l in (out or "").splitlines() if l.startswith('{"metric"')]
    for l in (out or "").splitlines():
        if not l.startswith('{"metric"'):
            sys.stderr.write(l + "\n")
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 3
    sys.exit(rc)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])

for p in (os.path.join(ROOT, "sequential-inverse-kinematics_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (loads the HIP runtime that libseqik_hip.so binds to)

from seqikpy_amd import _lib, data, peer_gather, sharding, synthetic, utils  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TF = 78.6  # 256 CUs x 4 SIMDs x 16 f64 lanes x 2 flop x 2.4 GHz
# Algorithmic HBM bytes per leg-frame (DESIGN.md "Kernels"):
BYTES_PATH = 120 + 56 + 216   # key points in, 7 angles out, 9x3 FK out
BYTES_STAGE = {1: 48 + 16 + 96, 2: 48 + 96 + 16 + 96 + 48, 3: 48 + 96 + 16 + 96 + 24, 4: 48 + 96 + 8 + 144}
# stage k reads the origin + its key point (48 B) and the 96-byte prefix frame the previous stage left in
# the workspace, writes its own angles, the next prefix frame (96 B) and its rows of the 9 x 3 FK record
TRAFFIC_ROUNDS = ("r05", "r04", "r03", "r02", "r01")  # newest first; a summary is used only if it matches the workload AND the build
LATENCY_ROUND = "r05"     # profiles/<round>_latency_floor.json (scripts/latency_floor.py)
LF_WINDOW = (284, 302)   # tests/conftest.py::LF_DEGENERATE: the anipose LF kinematic-singularity episode


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 100 steps = 1.4 s: the first and last launches of a run overlap with fewer neighbours, and with 20 steps that
    # edge still costs 5 % (20 steps 4.11e8, 400 steps 4.30e8, 1500 steps 4.31e8 solves/s)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): --frames in total, split over the ranks (BASELINE config 3 literally: ONE fixed problem); "
                         "weak: --frames per GPU.  The same run at N = 1; the other mode is a named leg of the N > 1 line")
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames (x 6 legs) per GPU (weak) or in total (strong)")
    ap.add_argument("--frames-per-seq", type=int, default=64)
    ap.add_argument("--variant", default="iid", choices=["iid", "smooth"])
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--lanes-per-wave", type=int, default=0, help="chains per wavefront (0 = automatic)")
    ap.add_argument("--interleave-legs", type=int, default=0,
                    help="1 = consecutive chains per wave (legs interleaved) instead of leg-pure, longest-leg-first waves")
    ap.add_argument("--stage-pipeline", type=int, default=-1,
                    help="SeqikOptions.reserved[3]: 0 = the library's choice (stage pipeline up to 40 000 chains: right for ONE call, "
                         "whose latency it halves), 1 = never (lane per chain kernels: right when many steps are in flight), 2 = always; "
                         "-1 (default) = 0 with an explicit --streams, calibrated together with the depth otherwise")
    ap.add_argument("--staged", action="store_true",
                    help="one launch per stage (SeqikOptions.reserved[1] = 1) instead of the default single launch in "
                         "which every wave takes its chains through the four stages in turn")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps are issued on round-robin (consecutive steps overlap).  0 (default) = calibrated in "
                         "the run: (3 streams, library's kernel choice), (8 / 12 / 16 streams, lane-per-chain kernels) are each timed "
                         "for a few steps on this rank's batch with the gather running, the fastest on the slowest rank is taken")
    ap.add_argument("--one-recording", action="store_true",
                    help="config 3 literally: ONE recording of --frames frames x 6 legs (real locomotion poses repeated), "
                         "frame-sharded over the ranks on the library's frame chunks, end states exchanged, angles all-gathered")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-python-baseline", action="store_true",
                    help="skip the Python + scipy process-pool leg of the CPU baseline (about 20 s)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` object (BASELINE configs 1 / 2 / 4 / 5 and the generic chain; about a minute)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip single_job / variants / single_recording / strong_projection / parity (profiling runs)")
    ap.add_argument("--cpu-sample-seqs", type=int, default=8192,
                    help="sequences of the batch the CPU baseline solves (8192 x 6 x 64 = 3.1 M leg-frames: 10-20 s on 16 cores)")
    return ap.parse_args()


def make_workload(n_seq, n_frames, variant, seed):
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    pose = synthetic.synthetic_pose(n_seq, n_frames, legs, data.BOUNDS_LOCOMOTION, body,
                                    data.TEMPLATE_NMF_LOCOMOTION, variant=variant, seed=seed)
    params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    return legs, body, pose, params


def usable_cores():
    """Hardware threads this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box reports 256 logical CPUs but grants a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(pose, legs, body, n_seq_sample, python_pool=True):
    """The C oracle (oracle/seqik_oracle.c) on the host cores: one task per (sequence, leg), the
    shape of the reference's parallel example (examples/example_leg_inv_kinematics_parallel.py:186)."""
    from oracle import c_oracle
    c_oracle.lib()
    cores = usable_cores()
    n_seq_sample = min(n_seq_sample, pose.shape[0])
    par = [c_oracle.leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    segs, bnds, seeds = (np.stack([p[i] for p in par]) for i in range(3))
    workers = min(cores, n_seq_sample)
    spans = [sharding.partition(n_seq_sample, workers, w) for w in range(workers)]

    def run(span):  # one C call per worker: ctypes releases the GIL for its whole duration
        c_oracle.seq_batch(pose[span[0]:span[1]], segs, bnds, seeds, want_fk=True)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(run, spans))
    dt = time.perf_counter() - t0
    cores = workers
    units = n_seq_sample * len(legs) * pose.shape[2]
    out = {"value": units / dt, "unit": "leg-frame solves/s", "cores": cores, "kind": "port",
           "sample": f"{n_seq_sample} of the {pose.shape[0]} sequences x 6 legs x {pose.shape[2]} frames "
                     f"({units} leg-frames, {dt:.1f} s wall); reference's own published rates for real IKPy: "
                     "5.6/s serial, 17.2/s on 4 cores (example_leg_inv_kinematics_parallel.py:4-6)"}
    if python_pool:
        # The reference's CPU path in its own shape: Python frame loop + real scipy.optimize.least_squares per
        # (frame, stage) over an IKPy stand-in, multiprocessing.Pool with one task per (sequence, leg)
        # (examples/example_leg_inv_kinematics_parallel.py:186-187).  Runs in a fresh interpreter.
        from oracle import scipy_oracle
        n_py = min(cores, pose.shape[0])
        _, secs = scipy_oracle.pool_run_subprocess(pose[:n_py], legs, data.BOUNDS_LOCOMOTION, body,
                                                   data.INITIAL_ANGLES_LOCOMOTION, cores)
        py_units = n_py * len(legs) * pose.shape[2]
        out["python_scipy_pool"] = {"value": py_units / secs, "unit": "leg-frame solves/s", "cores": cores,
                                    "engine": "oracle/scipy_oracle.py: real scipy TRF per (frame, stage), numpy link "
                                              "matrices instead of IKPy's sympy-built ones (faster than real IKPy)",
                                    "sample": f"{n_py} sequences x 6 legs x {pose.shape[2]} frames = {py_units} "
                                              f"leg-frames, {secs:.1f} s in the pool"}
    return out


_STREAMS = []


def stream_pool(n):
    """The first n streams of ONE pool per process (the current stream first): every batch of a run launches on the same
    streams, so the process never holds more streams than the deepest pipeline asks for -- the library keeps a hand-off
    workspace per (device, stream), at most 16 of them, and the hardware queues are as few."""
    if not _STREAMS:
        _STREAMS.append(torch.cuda.current_stream())
    while len(_STREAMS) < n:
        _STREAMS.append(torch.cuda.Stream())
    return list(_STREAMS[:n])


class Batch:
    """One rank's batch resident in HBM (planar layout) + the launch of one step on a given stream."""

    def __init__(self, pose, params, args, n_streams, pipeline=None, like=None, s_pad=None):
        """`like`: another Batch of the SAME key points (its device copy and streams are shared, only FK buffers are added).
        `s_pad`: sequences the ANGLE buffers are allocated for (>= this rank's own): the shares of the fixed problem differ by
        one sequence between ranks (15 625 = 8 x 1 953 + 1), and the gather moves equal blocks from every rank."""
        self.params, self.args = params, args
        self.s_pad = s_pad if s_pad is not None else (like.s_pad if like is not None else None)
        self.pipeline = max(0, getattr(args, "stage_pipeline", 0)) if pipeline is None else pipeline
        self.streams = stream_pool(n_streams)
        self.main = self.streams[0]
        self.lat_range = None     # [lo, hi) steps of a timed region launched with the library's own kernel choice (depth_candidates)
        if like is not None:
            self.S, self.L, self.T, self.layout, self.d_pose = like.S, like.L, like.T, like.layout, like.d_pose
            self.d_fks = list(like.d_fks[:n_streams])
        else:
            self.S, self.L, self.T = pose.shape[:3]
            self.layout = _lib.planar_layout(self.T)
            # planar device layout (include/seqik.h, SeqikLayout): pose [S][L][5][T][3], angles [S][L][7][T]
            self.d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
            self.d_fks = []
        while len(self.d_fks) < len(self.streams):
            self.d_fks.append(torch.zeros((self.S, self.L, self.T, 9, 3), dtype=torch.float64, device="cuda"))
        self.units = self.S * self.L * self.T

    def angle_buffer(self):
        return torch.zeros((max(self.S, self.s_pad or 0), self.L, 7, self.T), dtype=torch.float64, device="cuda")

    def launch(self, i, buf, events=None, n_streams=None, tail=False):
        k = i % (n_streams or len(self.streams))
        stream = self.streams[k]
        a = self.args
        # ONE C-ABI call = the whole hot path; the library records the given HIP events around its kernels
        # (`tail`: a step of `lat_range` -- the partial round of a deep pipeline -- is launched with the library's own kernel
        # choice: for a share that is the stage pipeline, whose launch is over in half the time)
        pipe = 0 if tail else self.pipeline
        _lib.solve_seq_device(self.d_pose.data_ptr(), self.S, self.L, self.T, self.params, buf.data_ptr(),
                              self.d_fks[k].data_ptr(), stream=stream.cuda_stream, block_size=a.block, layout=self.layout,
                              lanes_per_wave=a.lanes_per_wave, staged=int(a.staged), interleave_legs=a.interleave_legs,
                              pipeline=pipe,
                              stage_events=[e.cuda_event for e in events] if events else None)
        return stream


def depth_candidates(steps, chains=None):
    """(steps in flight, SeqikOptions.reserved[3], latency-kernel steps [lo, hi) or None) the run calibrates among (parse():
    --streams 0).  Beside the fixed depths: the BALANCED depth -- `steps` cut into the fewest rounds of at most MAX_DEPTH, all
    of the same size (20 steps: all at once; 100 steps: 5 x 20) -- and depths 16 / 20 with the partial round (at most 8 steps)
    launched with the library's own kernel choice instead of the lane-per-chain kernels: it is the LAST round, which runs on
    a draining GPU, and the stage pipeline's launch is over in half the time (1/8 share, 20 steps at depth 16: 2.9 -> 2.3 ms
    per step; the same launches put FIRST, to make room early, lose: 3.1; profiles/r05_depth_calibration_k20_k100.jsonl).
    `chains`: chains per step -- depths that would put more than twice the GPU's wavefront slots in flight are left out (the
    whole problem: 3 and 4; a 1/8 share: everything): deeper buys nothing there, and every stream is a hardware queue, of
    which a process should not hold more than it needs (see the strong_projection leg)."""
    cap = depth_cap(chains)
    return [c for c in _depth_candidates(steps) if c[0] <= cap]


def depth_cap(chains):
    """Steps in flight beyond which a batch of `chains` chains has more than twice the GPU's 3 072 wavefront slots in flight."""
    return MAX_DEPTH if not chains else max(3, min(MAX_DEPTH, -(-2 * 3072 // max(1, -(-chains // 64)))))


def _depth_candidates(steps):
    cands = [c for c in ((3, 0, None), (8, 1, None), (12, 1, None), (16, 1, None), (20, 1, None)) if c[0] <= MAX_DEPTH]
    rounds = -(-steps // MAX_DEPTH)
    balanced = -(-steps // rounds)
    if balanced > 3 and balanced not in (8, 12, 16, 20):
        cands.append((balanced, 1, None))
    for depth in (16, 20):
        rest = steps % depth
        if depth <= MAX_DEPTH and steps > depth and 0 < rest <= 8:
            cands.append((depth, 1, (steps - rest, steps)))
    return cands


def in_lat_range(batch, i):
    return batch.lat_range is not None and batch.lat_range[0] <= i < batch.lat_range[1]


DEPTH_CANDIDATES = _depth_candidates(10 ** 6)   # the fixed depths (a long run has no partial round worth a special case)


def setup_streams(batch, bufs, n_streams):
    """SETUP, not warm-up: the library allocates a stream's stage hand-off workspace (96 B per leg-frame) at the first launch
    it sees on that stream, with a device-wide synchronisation; one launch per stream that has not carried this batch size
    yet keeps those allocations out of every timed region, however few warm-up steps the caller asks for."""
    done = getattr(batch, "_streams_set_up", 0)
    if done >= n_streams:
        return
    for k in range(done, n_streams):
        with torch.cuda.stream(batch.streams[k]):
            batch.launch(k, bufs[k % len(bufs)], n_streams=n_streams)
    torch.cuda.synchronize()
    batch._streams_set_up = n_streams


def timed_steps(batch, bufs, steps, n_streams, warmup=2):
    """`steps` launches round-robin over `n_streams` streams; returns seconds (host clock around a full drain)."""
    setup_streams(batch, bufs, n_streams)
    for i in range(warmup):
        with torch.cuda.stream(batch.streams[i % n_streams]):
            batch.launch(i, bufs[i % len(bufs)], n_streams=n_streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(batch.streams[i % n_streams]):
            batch.launch(i, bufs[i % len(bufs)], n_streams=n_streams, tail=in_lat_range(batch, i))
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def single_recording(n_frames=1_000_000, steps=4):
    """Config 3's size on ONE recording: 1M frames x 6 legs walked as the reference would walk them (frame t
    warm-started from frame t-1 over the whole recording), solved by frame chunks with automatic parameters
    (SeqikOptions.frame_chunk = -1).  The key points are the df3d locomotion recording of the fixtures (1000 frames
    x 6 legs, tests/golden/df3d_1000.npz) repeated end to end: real, temporally continuous fly poses -- on the
    synthetic random poses of the sequence benchmark the warm start selects among several equivalent leg
    configurations, the run-in of a chunk often lands in another one than the serial walk, and most chunks have to be
    repaired (DESIGN.md "Frame chunks", measured)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    L = len(legs)
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    base = np.stack([z[f"{l}_pose"] for l in legs])                  # (L, 1000, 5, 3)
    reps = -(-n_frames // base.shape[1])
    N = n_frames
    rec = np.ascontiguousarray(np.tile(base, (1, reps, 1, 1))[:, :N].transpose(0, 2, 1, 3))  # [L][5][N][3] planar
    d_pose = torch.from_numpy(rec).cuda()
    d_ang = torch.zeros((1, L, 7, N), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((1, L, N, 9, 3), dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(_lib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
    layout = _lib.planar_layout(N)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.solve_seq_device(d_pose.data_ptr(), 1, L, N, params, d_ang.data_ptr(), d_fk.data_ptr(), stream=stream,
                              layout=layout, frame_chunk=-1, d_chunk_stats=d_stats.data_ptr())
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    stats = _lib.chunk_stats_dict(d_stats.cpu().numpy())
    # the first 2000 frames walked serially (bit-exact restatement): chunk 0 must reproduce that walk bit for bit,
    # the other chunks to the noise floor of the verification tolerance; and against the fixture's reference angles
    n_head = min(N, 2000)
    h_pose = torch.from_numpy(np.ascontiguousarray(rec[:, :, :n_head])).cuda()
    h_ang = torch.zeros((1, L, 7, n_head), dtype=torch.float64, device="cuda")
    _lib.solve_seq_device(h_pose.data_ptr(), 1, L, n_head, params, h_ang.data_ptr(), 0, stream=stream,
                          layout=_lib.planar_layout(n_head))
    torch.cuda.synchronize()
    c = max(stats["frames_per_chunk"], 1)
    diff = (d_ang[0, :, :, :n_head] - h_ang[0]).abs()
    ref = torch.from_numpy(np.stack([z[f"{l}_angles"] for l in legs]).transpose(0, 2, 1)).cuda()  # (L, 7, 1000)
    n_ref = min(N, 1000)
    out = {"value": L * N / dt, "unit": "leg-frame solves/s", "ms_per_step": dt * 1e3, "frames": N, "legs": L,
           "data": "df3d locomotion recording (fixture, 1000 frames x 6 legs) repeated end to end",
           "mode": "frame chunks, automatic parameters (SeqikOptions.frame_chunk = -1), 7 angles + FK",
           "chunk_stats": stats,
           "check": {"frames_walked_serially": n_head,
                     "first_chunk_equals_serial_bit_for_bit": bool((diff[:, :, :c] == 0).all().item()),
                     "max_abs_vs_serial": float(diff.max().item()),
                     "leg_frames_over_1e-4_vs_serial": int((diff.amax(1) > 1e-4).sum().item()),
                     "max_abs_vs_reference_first_1000_frames": float((d_ang[0, :, :, :n_ref] - ref[:, :, :n_ref]).abs().max().item())}}
    del d_pose, d_ang, d_fk, h_pose, h_ang
    return out


def one_recording_leg(dist, world, rank, n_frames, steps, warmup, coll_dev="cpu"):
    """Config 3 read literally: ONE recording of n_frames x 6 legs (the df3d locomotion recording of the fixtures repeated
    end to end), contiguous frame slabs over the ranks (seqikpy_amd.frame_sharding: every rank's slab goes through the
    library's frame chunks with a run-in, the ranks all-gather their 56-byte end states, settle their first chunk in a
    resume call, and all-gather the joint angles; FK stays sharded).  A step = one such solve with the key points
    resident in HBM.  At N = 1 this is `single_recording`."""
    from seqikpy_amd import frame_sharding
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    L = len(legs)
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    base = np.stack([z[f"{l}_pose"] for l in legs])                                          # (L, 1000, 5, 3)
    pose = np.tile(base, (1, -(-n_frames // base.shape[1]), 1, 1))[None, :, :n_frames]       # (1, L, N, 5, 3)
    rec = frame_sharding.FrameShardedRecording(pose, params, want_fk=True)

    def sync():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(max(1, warmup)):
        out = rec.solve(gather_fk=False)
    sync()
    rec.spec_events = []          # solve() records a pair of HIP events around the speculative pass of every step
    t0 = time.perf_counter()
    for _ in range(steps):
        out = rec.solve(gather_fk=False)
    sync()
    mine = time.perf_counter() - t0
    rec.check_faults()            # outside the timed region: a kernel fault of any step raises here
    tmax = mine
    if dist:
        t = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
    # check on rank 0: the first 2000 frames against the serial walk of those frames (chunk 0 bit for bit, the rest to the
    # verification tolerance's noise floor) and against the fixture's reference angles
    chk = None
    if rank == 0:
        n_head = min(n_frames, 2000)
        ser = _lib.solve_seq(np.ascontiguousarray(pose[:, :, :n_head]), params, want_fk=False)["angles"]
        got = out["angles"][:, :, :n_head].cpu().numpy()
        ref = np.stack([z[f"{l}_angles"] for l in legs])[None]
        n_ref = min(n_frames, 1000)
        chk = {"frames_walked_serially": n_head, "max_abs_vs_serial": float(np.abs(got - ser).max()),
               "first_chunk_equals_serial_bit_for_bit": bool(np.array_equal(got[:, :, :rec.C], ser[:, :, :rec.C])),
               "max_abs_vs_reference_first_1000_frames": float(np.abs(got[:, :, :n_ref] - ref[:, :, :n_ref]).max())}
    spec_ms = [a.elapsed_time(b) for a, b in rec.spec_events] if rec.spec_events else []
    return {"value": L * n_frames * steps / tmax, "unit": "leg-frame solves/s", "ms_per_step": tmax / steps * 1e3, "steps": steps,
            "speculative_pass_ms_this_rank": float(np.mean(spec_ms)) if spec_ms else None,
            "frames": n_frames, "legs": L, "frames_per_rank": [b - a for a, b in rec.slabs],
            "frames_per_chunk": rec.C, "run_in_frames": rec.h, "boundary_rounds": rec.stats.get("boundary_rounds"),
            "resume_calls_per_step_this_rank": rec.stats.get("resume_calls"),
            "data": "df3d locomotion recording (fixture, 1000 frames x 6 legs) repeated end to end",
            "exchange": "all-gather of 56 B end states per leg and rank + one padded all-gather of the joint angles; FK stays sharded",
            "check": chk}


def measured_cost_floor(mix, n_all, simds, clock_hz, ms_per_step):
    """The same issue floor priced with MEASURED issue costs per instruction class (scripts/microbench/valu_issue.hip ->
    profiles/r03_valu_issue_costs.json, three wavefronts per SIMD): f64 add / mul / fma ~4.25 cycles per wavefront
    instruction, v_rcp_f64 / v_rsq_f64 ~16.2, and of the remaining vector instructions the share that profiles/
    r03_fused_isa.json finds to be f64-class / scalar-mask / 64-bit instructions (~4.25 cycles too) against plain 32-bit
    ones (~2.7).  None when the files are absent."""
    try:
        costs = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_issue_costs.json")))["classes"]
        isa_path = next(p for p in (os.path.join(ROOT, "profiles", f"{r}_fused_isa.json") for r in ("r05", "r04", "r03")) if os.path.exists(p))
        isa = json.load(open(isa_path))["kernels"]["fused_kernel<fk=1>"]
        share4 = isa["valu_not_f64_arith_issue_split"]["share_about_4.2_cycles"]
    except (OSError, KeyError, ValueError, StopIteration):
        return None
    c = lambda name: costs[name]["waves_per_simd_3"]["cycles_per_inst"]  # noqa: E731
    c_f64 = (c("v_fma_f64") + c("v_mul_f64") + c("v_add_f64")) / 3.0
    c_trans = (c("v_rcp_f64") + c("v_rsq_f64")) / 2.0
    c_other4 = (c("v_cmp_lt_f64") + c("v_max_f64") + c("v_mov_b64") + c("the same with the mask in an SGPR pair (VOP3)")) / 4.0
    c_other2 = (c("v_mov_b32") + c("v_add_u32") + c("v_xor_b32")) / 3.0
    n_arith = sum(mix["add"]) + sum(mix["mul"]) + sum(mix["fma"])
    n_trans = sum(mix["trans"])
    n_other = n_all - n_arith - n_trans
    cycles = n_arith * c_f64 + n_trans * c_trans + n_other * (share4 * c_other4 + (1.0 - share4) * c_other2)
    floor_ms = cycles / (simds * clock_hz) * 1e3
    return {"cycles_per_inst": {"f64_add_mul_fma": c_f64, "f64_rcp_rsq": c_trans, "other_4_cycle_class": c_other4,
                                "other_32_bit": c_other2, "share_of_other_in_4_cycle_class": share4},
            "issue_floor_ms_per_step": floor_ms, "frac_of_valu_issue_floor": floor_ms / ms_per_step,
            "source": "profiles/r03_valu_issue_costs.json (microbenchmark, 3 waves per SIMD) x PMC counts; "
                      "profiles/r03_fused_isa.json for the split of the instructions the PMC classes do not cover"}


def pmc_roofline(variant, staged, key, units_per_step, ms_per_step, device_index):
    """The VALU-side roofline figures from the newest committed PMC summary (profiles/traffic_rNN*.json, written by
    scripts/summarize_profile.py) that matches this workload -- used only if it was measured on THIS build: the summary
    carries the sha256 of the solver kernels' sources, which must equal the sources the loaded library was built from.
    -> (traffic, valu, fp64, matches_build, file)"""
    suffix = ("_staged" if staged else "") + ("" if variant == "iid" else "_" + variant)
    for rnd in TRAFFIC_ROUNDS:
        tpath = os.path.join(ROOT, "profiles", f"traffic_{rnd}{suffix}.json")
        if not os.path.exists(tpath):
            continue
        tj = json.load(open(tpath))
        if tj.get("variant") != variant or not tj.get("units_per_launch"):
            continue
        # A rank of an N > 1 run (or a --frames run) solves a SHARE of the same synthetic distribution with the same kernel:
        # the per-launch counters are scaled by the number of leg-frames (the instruction mix per leg-frame is a property of
        # the data distribution and the kernel; `pmc_scaled_from_units` says when that was done)
        scale = units_per_step / float(tj["units_per_launch"])
        if scale != 1.0:
            tj = {k: (v * scale if isinstance(v, (int, float)) and k.endswith("_per_launch") and k != "units_per_launch" else v)
                  for k, v in tj.items()}
            tj["scaled_from_units"] = tj["units_per_launch"]
        matches = tj.get("csrc_sha256") == _lib.csrc_sha256()
        traffic = tj.get(f"{key}_hbm_bytes_per_launch")
        if not matches:   # counters of another build say nothing about this one
            return None, None, None, False, os.path.basename(tpath)
        valu, fp64 = None, None
        names = [f"stage{k}" for k in (1, 2, 3, 4)] if staged else ["fused"]
        insts = [tj.get(f"{n}_valu_insts_per_launch") for n in names]
        mix = {c: [tj.get(f"{n}_f64_{c}_insts_per_launch") for n in names] for c in ("add", "mul", "fma", "trans")}
        utils_ = [tj.get(f"{n}_valu_lane_utilisation") for n in names]
        if all(v is not None for v in insts) and all(v is not None for vs in mix.values() for v in vs):
            # What actually bounds the path: VALU issue.  A wave64 FP64 instruction occupies its SIMD's 16 f64 lanes
            # for 4 cycles; every other VALU instruction (selects, compares, moves, 64-bit address arithmetic) takes
            # 2 cycles on the SIMD-32 when several waves share a SIMD (MI355X_MICROARCH.md, "Execution model" and the
            # cycle-constants row `v_fma_f32` wave64).  The quarter-rate rcp / rsq / sqrt seeds are priced like the
            # other f64 instructions, so this is a FLOOR: the step cannot be shorter than
            #     (f64 instructions x 4 + other VALU instructions x 2) / (SIMDs x clock).
            n_cu, clock_khz, _ = _lib.device_attributes(device_index)
            simds, clock_hz = n_cu * 4, clock_khz * 1e3
            n_f64 = sum(sum(vs) for vs in mix.values())
            n_all = sum(insts)
            floor_ms = (n_f64 * 4.0 + (n_all - n_f64) * 2.0) / (simds * clock_hz) * 1e3
            valu = {"valu_insts_per_step": n_all, "f64_insts_per_step": n_f64, "simds": simds, "clock_MHz": clock_khz / 1e3,
                    "cycles_per_inst": {"f64": 4, "other": 2},
                    "issue_floor_ms_per_step": floor_ms, "measured_ms_per_step": ms_per_step,
                    "frac_of_valu_issue_floor": floor_ms / ms_per_step,
                    "measured_costs": measured_cost_floor(mix, n_all, simds, clock_hz, ms_per_step),
                    "lane_utilisation": utils_,
                    "source": "SQ_INSTS_VALU / SQ_INSTS_VALU_*_F64 / SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU per launch "
                              f"from profiles/{os.path.basename(tpath)} (rocprofv3 --pmc, own passes), timing live"}
            if all(u is not None for u in utils_):
                # FP64 operations actually performed: wave-level instruction counts by class x 64 lanes x the
                # share of active lanes (FMA = 2 flops), against the 78.6 TFLOP/s FP64 vector peak
                flops = sum((mix["add"][i] + mix["mul"][i] + mix["trans"][i] + 2.0 * mix["fma"][i]) * 64.0 * utils_[i]
                            for i in range(len(names)))
                fp64 = {"flops_per_step": flops, "f64_insts_per_step": {c: sum(vs) for c, vs in mix.items()},
                        "note": "lane share taken from all VALU instructions (SQ_THREAD_CYCLES_VALU)"}
        if fp64 is not None and tj.get("scaled_from_units"):
            fp64["pmc_scaled_from_units"] = tj["scaled_from_units"]
        return traffic, valu, fp64, True, os.path.basename(tpath)
    return None, None, None, None, None


def parity_tail(err, ok, legs):
    """The tail of |d theta| of one fixture, so that a drift toward the 1e-4 bar is visible before it crosses: p99 / p99.9
    over all (leg, frame, joint) values outside the excluded window, how many of them lie above half the bar, and where
    the maximum sits.  `err` (L, N, 7), `ok` (L, N) bool."""
    vals = err[ok]                                        # (leg-frames kept, 7)
    masked = np.where(ok[:, :, None], err, -1.0)
    li, t, j = np.unravel_index(int(np.argmax(masked)), masked.shape)
    return {"p99_abs_dtheta": float(np.quantile(vals, 0.99)), "p99.9_abs_dtheta": float(np.quantile(vals, 0.999)),
            "values_over_5e-5": int((vals > 5e-5).sum()), "values_compared": int(vals.size),
            "max_at": {"leg": legs[li], "joint": data.DOFS[j], "frame": int(t)},
            "frac_of_1e-4_budget": float(vals.max() / 1e-4)}


def parity_report():
    """HIP vs the committed reference fixtures, on the GPU, fixtures only (no oracle involved): the shipped anipose
    outputs (reference's leg_joint_angles.pkl, RF + LF x 6000 frames) and the df3d recording solved by the
    reference's unmodified source over real scipy in the build container (6 legs x 1000 frames)."""
    rep = {"tolerance_rad": 1e-4,
           "lf_window": "anipose LF frames %d-%d: kinematic-singularity episode, the reference itself is not "
                        "reproducible there (tests/conftest.py::LF_DEGENERATE, profiles/r02_perturbation_report.json)" % (LF_WINDOW[0], LF_WINDOW[1] - 1)}
    for name in ("anipose_shipped", "df3d_1000"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        legs = [str(l) for l in z["legs"]]
        params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
        pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
        ref = np.stack([z[f"{l}_angles"] for l in legs])
        ok = np.ones(ref.shape[:2], bool)
        if name == "anipose_shipped":
            ok[legs.index("LF"), LF_WINDOW[0]:LF_WINDOW[1]] = False
        entry = {"legs": legs, "frames": int(pose.shape[2])}
        for mode, kw in (("serial_walk", {}), ("frame_chunks", dict(frame_chunk=-1))):
            out = _lib.solve_seq(pose, params, want_fk=False, **kw)
            err = np.abs(out["angles"][0] - ref)              # (L, N, 7)
            bad = np.argwhere(err.max(-1) > 1e-4)
            entry[mode] = {"max_abs_dtheta": float(err[ok].max()),
                           "max_abs_dtheta_incl_lf_window": float(err.max()),
                           "leg_frames_over_1e-4": int(len(bad)),
                           "leg_frames_over_1e-4_outside_lf_window": int(sum(ok[i, t] for i, t in bad)),
                           "where": [[legs[i], int(t)] for i, t in bad[:32]],
                           "median_abs_dtheta": float(np.median(err)),
                           **parity_tail(err, ok, legs)}
            if kw:
                entry[mode]["chunk_stats"] = {k: v for k, v in out["chunk_stats"].items() if v}
        rep[name] = entry
    return rep


def best_ms(fn, reps=5):
    """fn once untimed, then the fastest of `reps` runs, in ms (host clock: the whole call, transfers included)."""
    fn()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def latency_floor(kernel_key):
    """Committed PMC-derived issue floor of a latency-bound kernel (profiles/r05_latency_floor.json, written by
    scripts/latency_floor.py from rocprofv3 --pmc / --kernel-trace runs): the VALU instructions ONE wavefront issues per
    frame on the critical path, priced at the lone-wavefront issue cost per class.  None when absent / another build."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", f"{LATENCY_ROUND}_latency_floor.json")))
    except (OSError, ValueError):
        return None
    if j.get("csrc_sha256") != _lib.csrc_sha256(_lib.LATENCY_SOURCES):
        return None
    return j.get(kernel_key)


def share_floor(n):
    """Committed issue floor of the lone 1/n share of the fixed problem (profiles/r05_latency_floor.json `strong_share_<n>`,
    scripts/latency_floor.py: the instruction stream of the critical stage's wavefront at the lone-wavefront issue rate).
    None when absent or measured on other kernel sources."""
    fl = latency_floor(f"strong_share_{n}")
    if not fl:
        return None
    return {"issue_floor_ms": fl["issue_floor_ms"], "source": "profiles/%s_latency_floor.json (%s)" % (LATENCY_ROUND, fl.get("kernel", ""))}


def generic_batches(za, frames=32, sizes=(32768, 262144)):
    """Batches of generic chains (`LegInvKinGeneric` over many recordings: seqikpy/leg_inverse_kinematics.py:545-613 once
    per recording in the reference): windows of `frames` frames of the shipped 6000-frame recording, legs RF + LF, one lane
    per chain, device-resident, 7 angles + FK.  For every size the static launch (a wavefront owns 64 chains and lives as
    long as its slowest lane) against the chain queue (persistent wavefronts, one per SIMD; a lane that has finished its
    chain takes the next one of its leg) -- same bits, checked -- with the pass counts that explain the difference (from a
    diagnostics run: nfev per frame).  The library's automatic choice is the queue from four chains per GPU lane on."""
    import ctypes
    legs = ["RF", "LF"]
    params = [_lib.leg_params_from_arrays(za[f"{l}_seg"], za[f"{l}_bounds"], za[f"{l}_seeds"]) for l in legs]
    arr = (_lib.SeqikLegParams * 2)(*params)
    d_rec = torch.from_numpy(np.stack([za[f"{l}_pose"] for l in legs])).cuda()                     # (2, 6000, 5, 3)
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    n_cu = _lib.device_attributes(torch.cuda.current_device())[0]
    res = {"workload": f"windows of {frames} frames of the shipped recording (offsets 11 s mod {6000 - frames}), legs RF + LF, "
                       "generic chain, one lane per chain; leg_frames_per_s of the faster launch at the largest size is the figure",
           "gpu_lanes_for_this_kernel": n_cu * 4 * 64, "sizes": {}}
    for S in sizes:
        offs = (torch.arange(S, device="cuda") * 11) % (6000 - frames)
        idx = offs[:, None] + torch.arange(frames, device="cuda")[None, :]
        d_pose = d_rec[:, idx].permute(1, 0, 2, 3, 4).contiguous()                                  # (S, 2, T, 5, 3)
        d_ang = torch.zeros((S, 2, frames, 7), dtype=torch.float64, device="cuda")
        d_fk = torch.zeros((S, 2, frames, 9, 3), dtype=torch.float64, device="cuda")
        d_st = torch.zeros((S, 2, frames), dtype=torch.int32, device="cuda")
        d_nf = torch.zeros((S, 2, frames), dtype=torch.int32, device="cuda")

        def run(queue, diag=False):
            opt = _lib.SeqikOptions()
            opt.reserved[1] = queue
            rc = lib.seqik_solve_generic_device(d_pose.data_ptr(), S, 2, frames, arr, d_ang.data_ptr(), d_fk.data_ptr(),
                                                d_st.data_ptr() if diag else None, d_nf.data_ptr() if diag else None,
                                                None, None, None, ctypes.byref(opt), stream)
            if rc != 0:
                raise RuntimeError("seqik_solve_generic_device failed")

        row = {"sequences": S, "chains": 2 * S, "frames": frames, "leg_frames": 2 * S * frames,
               "chains_per_gpu_lane": 2 * S / (n_cu * 4 * 64.0)}
        keep, best = {}, {}
        variants = (("static", 1), ("queue", 2), ("automatic", 0))
        for name, q in variants:           # warm-up + the results of every variant
            run(q)
            torch.cuda.synchronize()
            keep[name] = (d_ang.clone(), d_fk.clone())
            best[name] = float("inf")
        for _ in range(3):                 # variants interleaved: the first launches after a pause run slower
            for name, q in variants:
                t0 = time.perf_counter()
                run(q)
                torch.cuda.synchronize()
                best[name] = min(best[name], time.perf_counter() - t0)
        for name, _ in variants:
            row[name] = {"ms": best[name] * 1e3, "leg_frames_per_s": 2 * S * frames / best[name]}
        row["queue_equals_static_bit_for_bit"] = bool(torch.equal(keep["static"][0], keep["queue"][0]) and
                                                      torch.equal(keep["static"][1], keep["queue"][1]) and
                                                      torch.equal(keep["static"][0], keep["automatic"][0]))
        row["queue_speedup_over_static"] = row["static"]["ms"] / row["queue"]["ms"]
        run(1, diag=True)
        torch.cuda.synchronize()
        passes = (d_nf - 1 + (d_st == 1).int()).sum(2)                                               # (S, 2) passes per chain
        pad = (-S) % 64
        wave = torch.stack([torch.nn.functional.pad(passes[:, l], (0, pad)).reshape(-1, 64).max(1).values for l in range(2)])
        row["passes"] = {"mean_lane": float(passes.float().mean().item()), "mean_wavefront_static": float(wave.float().mean().item()),
                         "slowest_wavefront_static": int(wave.max().item()), "slowest_chain": int(passes.max().item())}
        res["sizes"][str(2 * S)] = row
        del d_pose, d_ang, d_fk, d_st, d_nf, keep
        torch.cuda.empty_cache()
    big = res["sizes"][str(2 * sizes[-1])]
    res["leg_frames_per_s"] = max(big["queue"]["leg_frames_per_s"], big["static"]["leg_frames_per_s"])
    res["queue_speedup_over_static_largest"] = big["queue_speedup_over_static"]
    res["bound"] = ("profiles/r05_generic_queue_bound.json (oracle pass counts, list scheduling at a constant pass time): 1.00 / 1.21 / "
                    "1.38 / 1.57 at 1 / 2 / 4 / 8 chains per lane; the static launch beats that model because its passes get faster as the GPU drains")
    _lib.check_faults()
    return res


def reference_configs(time_box_s=240.0):
    """BASELINE.json configs 1, 2, 4, 5 and the generic chain, as a user of the reference would run them, timed in this
    process (`configs` of the JSON line).  Reference shapes: examples/example_leg_inv_kinematics.py:23-62 (config 1 and
    the generic chain), examples/example_leg_inv_kinematics_parallel.py:143-198 (config 2), examples/
    example_entire_pipeline.py:48-106 (config 4).  Every entry carries its parity figure next to its time."""
    import importlib.util
    from seqikpy_amd.head_inverse_kinematics import ANGLE_NAMES
    from seqikpy_amd.kinematic_chain import KinematicChainGeneric, KinematicChainSeq
    from seqikpy_amd.leg_inverse_kinematics import LegInvKinGeneric, LegInvKinSeq
    from seqikpy_amd.pipeline import run_body_ik
    t_start = time.perf_counter()
    DOFS = data.DOFS
    za = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    zd = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    zh = np.load(os.path.join(ROOT, "tests", "golden", "anipose_head.npz"))
    out = {"note": "ms = fastest of 5 whole calls on host arrays (upload, kernels, download, dict building); "
                   "leg_frames_per_s = legs x frames / that; default = the reference's serial walk (bit-identical to the C "
                   "restatement), frame_parallel_auto = verified frame chunks (opt-in); latency_floor_frac = issue floor of "
                   "the critical wavefront (committed PMC instruction counts x lone-wavefront issue costs) / measured kernel "
                   "time, for the kernels that are bound by the latency of one dependent chain, not by throughput"}

    def leg_entry(z, legs, n, bounds, init, template, workload, mask_lf):
        aligned = {f"{l}_leg": np.ascontiguousarray(z[f"{l}_pose"][:n]) for l in legs}
        body = utils.calculate_body_size(template, legs)
        chain = KinematicChainSeq(bounds_dof=bounds, legs_list=legs, body_size=body)
        ref = np.stack([z[f"{l}_angles"][:n] for l in legs])                       # (L, n, 7)
        ok = np.ones(ref.shape[:2], bool)
        if mask_lf and "LF" in legs:
            ok[legs.index("LF"), LF_WINDOW[0]:min(LF_WINDOW[1], n)] = False
        entry = {"workload": workload, "legs": legs, "frames": n, "leg_frames": len(legs) * n}
        got = {}
        for key, mode in (("default", False), ("frame_parallel_auto", "auto")):
            holder = {}

            def call():
                ik = LegInvKinSeq(aligned_pos=aligned, kinematic_chain_class=chain, initial_angles=init, log_level="ERROR")
                holder["ang"], holder["fk"] = ik.run_ik_and_fk(export_path=None, frame_parallel=mode)
                holder["ik"] = ik
            ms = best_ms(call)
            a = np.stack([np.stack([holder["ang"][f"Angle_{l}_{d}"] for d in DOFS], 1) for l in legs])   # (L, n, 7)
            got[key] = a
            err = np.abs(a - ref)
            e = {"ms": ms, "leg_frames_per_s": len(legs) * n / ms * 1e3,
                 "max_abs_dtheta_vs_fixture": float(err[ok].max()),
                 "leg_frames_over_1e-4": int((err.max(-1) > 1e-4)[ok].sum()), **parity_tail(err, ok, legs)}
            if mask_lf and "LF" in legs:
                e["max_abs_dtheta_incl_lf_window"] = float(err.max())
            if mode:
                st = holder["ik"].frame_chunk_stats
                e["chunk_stats"] = {k: v for k, v in st.items() if v}
                dd = np.abs(a - got["default"])
                e["max_abs_vs_default"] = float(dd[ok].max())
                if mask_lf and "LF" in legs:
                    e["max_abs_vs_default_incl_lf_window"] = float(dd.max())
            entry[key] = e
        return entry, aligned, chain

    legs6 = [str(l) for l in zd["legs"]]
    out["1"], _, _ = leg_entry(za, ["RF"], 100, data.BOUNDS, data.INITIAL_ANGLES, data.NMF_TEMPLATE,
                               "config 1: single right-front leg, 100 frames of anipose_220525_aJO_Fly001_001 "
                               "(LegInvKinSeq.run_ik_and_fk; fixture = the shipped leg_joint_angles.pkl)", False)
    out["2"], _, _ = leg_entry(zd, legs6, 1000, data.BOUNDS_LOCOMOTION, data.INITIAL_ANGLES_LOCOMOTION,
                               data.TEMPLATE_NMF_LOCOMOTION,
                               "config 2: all 6 legs, df3d locomotion recording, 1000 frames (fixture = the reference's source run "
                               "over real scipy, oracle/gen_golden.py)", False)
    # ---- the reference's semantics at the target rate: MANY recordings per call (a lab's flies / trials), every chain still
    # walked frame by frame (the default), each recording's result the bits it gets alone (tests/test_frame_chunks.py)
    from seqikpy_amd.batch import run_ik_and_fk_many
    recs = [{f"{l}_leg": np.ascontiguousarray(zd[f"{l}_pose"]) for l in legs6} for _ in range(64)]
    chain6 = KinematicChainSeq(bounds_dof=data.BOUNDS_LOCOMOTION, legs_list=legs6,
                               body_size=utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs6))
    holder = {}

    def many():
        holder["res"] = run_ik_and_fk_many(recs, chain6, data.INITIAL_ANGLES_LOCOMOTION)
    ms_many = best_ms(many, reps=3)
    a_many = np.stack([np.stack([holder["res"][-1][0][f"Angle_{l}_{d}"] for d in DOFS], 1) for l in legs6])
    ref6 = np.stack([zd[f"{l}_angles"] for l in legs6])
    out["2"]["default_64_recordings_one_call"] = {
        "what": "run_ik_and_fk_many: 64 recordings x 6 legs x 1000 frames in one call, DEFAULT semantics (serial walk per chain, "
                "bit-identical to the one-recording call)", "ms": ms_many, "leg_frames_per_s": 64 * 6 * 1000 / ms_many * 1e3,
        "max_abs_dtheta_vs_fixture_last_recording": float(np.abs(a_many - ref6).max())}
    # ---- config 4: legs + head / antenna angles of the shipped 6000-frame recording in ONE submission ----------------
    e4, aligned4, chain4 = leg_entry(za, ["RF", "LF"], 6000, data.BOUNDS, data.INITIAL_ANGLES, data.NMF_TEMPLATE,
                                     "config 4: anipose_220525_aJO_Fly001_001 (6000 frames; stands in for the absent "
                                     "anipose_220807_Fly002_002), legs RF + LF + the 7 head / antenna angles", True)
    body_in = dict(aligned4, R_head=zh["R_head"], L_head=zh["L_head"], Neck=zh["Neck"])
    e4["legs_and_head_one_submission"] = {}
    for key, mode in (("default", False), ("frame_parallel_auto", "auto")):
        holder = {}

        def call():
            holder["body"], holder["fk"] = run_body_ik(body_in, chain4, data.NMF_TEMPLATE, data.INITIAL_ANGLES, frame_parallel=mode)
        ms = best_ms(call)
        head = np.stack([holder["body"][k] for k in ANGLE_NAMES], 1)
        legs_a = np.stack([np.stack([holder["body"][f"Angle_{l}_{d}"] for d in DOFS], 1) for l in ("RF", "LF")])
        ref = np.stack([za[f"{l}_angles"] for l in ("RF", "LF")])
        ok = np.ones(ref.shape[:2], bool)
        ok[1, LF_WINDOW[0]:LF_WINDOW[1]] = False
        e4["legs_and_head_one_submission"][key] = {
            "ms": ms, "leg_frames_per_s": 2 * 6000 / ms * 1e3, "angles_per_frame": 21,
            "max_abs_dtheta_legs_vs_fixture": float(np.abs(legs_a - ref)[ok].max()),
            "max_abs_head_vs_shipped_head_joint_angles": float(np.abs(head - zh["shipped"]).max())}
    # the head / antenna kernel on its own at a size where it is bound by HBM (config 4's 6000 frames are a launch latency):
    # 16 M frames resident in HBM, HIP events on the stream the kernel is launched on
    try:
        reps = 16_000_000 // 6000
        d_r = torch.from_numpy(zh["R_head"]).cuda().repeat(reps, 1, 1)
        d_l = torch.from_numpy(zh["L_head"]).cuda().repeat(reps, 1, 1)
        d_neck = torch.from_numpy(zh["Neck"][0, 0].copy()).cuda()
        n_h = d_r.shape[0]
        d_out = torch.zeros((7, n_h), dtype=torch.float64, device="cuda")
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream

        def head_launch():
            rc = lib.seqik_head_angles_device(d_r.data_ptr(), d_l.data_ptr(), n_h, d_neck.data_ptr(), 0,
                                              float(zh["rest_head_pitch"][0]), float(zh["rest_antenna_pitch"][0]), 1,
                                              d_out.data_ptr(), stream)
            if rc != 0:
                raise RuntimeError("seqik_head_angles_device failed")
        for _ in range(15):
            head_launch()
        k_h = 30
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k_h + 1)]
        evs[0].record()
        for i in range(k_h):
            head_launch()
            evs[i + 1].record()
        torch.cuda.synchronize()
        each = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(k_h)])
        gbps = (96 + 56) * n_h / float(each.mean()) / 1e6
        small = _lib.head_angles(zh["R_head"], zh["L_head"], zh["Neck"][:, 0], float(zh["rest_head_pitch"][0]),
                                 float(zh["rest_antenna_pitch"][0]))
        same = bool(np.array_equal(d_out[:, -6000:].cpu().numpy(), small))
        traffic = None
        try:   # committed PMC summary of the same kernel and size (scripts/gpu_head_profile.sh): HBM bytes per frame
            hp = next(p for p in (os.path.join(ROOT, "profiles", f"{r}_head_profile.json") for r in ("r05", "r04")) if os.path.exists(p))
            with open(hp) as fh:
                traffic = json.load(fh)["traffic_bytes_per_frame"] * n_h
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        # what THIS box's memory system gives a plain copy of the same byte volume right now (torch's vectorised copy kernel,
        # 76 B per frame each way, same events): the boxes of the pool differ by 15 % in this figure, and the head kernel
        # cannot be faster than a copy of its bytes (scripts/microbench/head_split.hip has the same-mix calibration kernels)
        box = None
        try:
            n_cp = n_h * 76 // 8
            c_src = torch.zeros(n_cp, dtype=torch.float64, device="cuda")
            c_dst = torch.empty_like(c_src)
            for _ in range(5):
                c_dst.copy_(c_src)
            cev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            cev[0].record()
            for i in range(20):
                c_dst.copy_(c_src)
                cev[i + 1].record()
            torch.cuda.synchronize()
            c_each = np.array([cev[i].elapsed_time(cev[i + 1]) for i in range(20)])
            box = 152.0 * n_h / float(c_each.mean()) / 1e6
            del c_src, c_dst
        except Exception:  # noqa: BLE001
            pass
        e4["head_kernel"] = {"kernel": "seqik_head_kernel<true>", "frames": n_h, "launches": k_h, "ms": float(each.mean()),
                             "box_copy_same_bytes_GBps": round(box, 1) if box else None,
                             "frac_of_box_copy": round(gbps / box, 3) if box else None,
                             "ms_best": float(each.min()), "frames_per_s": n_h / float(each.mean()) * 1e3,
                             "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s",
                                          "frac": round(gbps / 8000.0, 3), "algorithmic_bytes_per_frame": 152,
                                          "traffic": traffic,
                                          "traffic_source": "profiles/r0N_head_profile.json, newest (FETCH_SIZE doubled for 16-byte-per-"
                                                            "lane streaming loads as the guide prescribes, + WRITE_SIZE)"},
                             "equals_the_6000_frame_call_tiled": same}
        del d_r, d_l, d_out
    except Exception as exc:  # noqa: BLE001
        e4["head_kernel"] = {"error": f"{type(exc).__name__}: {exc}"}
    out["4"] = e4
    out["3"] = {"workload": "config 3: synthetic 1M frames x 6 legs", "see": "top level: value (3 batches in flight), "
                "value_single_job, variants.smooth, single_recording (ONE recording), strong_projection"}
    # ---- generic chain: the reference's LegInvKinGeneric example on the shipped recording ----------------------------
    zg = np.load(os.path.join(ROOT, "tests", "golden", "generic_rf_100.npz"))
    gen_aligned = {"RF_leg": np.ascontiguousarray(za["RF_pose"])}
    gchain = KinematicChainGeneric(bounds_dof=data.BOUNDS, legs_list=["RF"],
                                   body_size=utils.calculate_body_size(data.NMF_TEMPLATE, ["RF"]))
    holder = {}

    def gcall():
        ik = LegInvKinGeneric(aligned_pos=gen_aligned, kinematic_chain_class=gchain, initial_angles=data.INITIAL_ANGLES, log_level="ERROR")
        holder["ang"], holder["fk"] = ik.run_ik_and_fk()
    g_ms = best_ms(gcall, reps=2)
    g_ang = np.stack([holder["ang"][f"Angle_RF_{d}"] for d in DOFS], 1)
    claw = holder["fk"]["RF_leg"][:, 8]
    lo, hi = za["RF_bounds"][:, 0], za["RF_bounds"][:, 1]
    d_ref = np.abs(np.diff(zg["RF_angles"], axis=0))
    d_got = np.abs(np.diff(g_ang[:100], axis=0))
    out["generic"] = {
        "workload": "LegInvKinGeneric, RF, the shipped 6000-frame recording (example_leg_inv_kinematics.py:49-62)",
        "frames": 6000, "ms": g_ms, "us_per_frame": g_ms * 1e3 / 6000, "frames_per_s": 6000 / g_ms * 1e3,
        "max_abs_claw_vs_target": float(np.abs(claw - za["RF_pose"][:, 4]).max()),
        "max_abs_claw_vs_reference_run_first_100": float(np.abs(claw[:100] - zg["RF_fk"][:, 8]).max()),
        "all_angles_within_limits": bool((g_ang >= lo).all() and (g_ang <= hi).all()),
        "max_abs_dtheta_vs_reference_run_first_100": float(np.abs(g_ang[:100] - zg["RF_angles"]).max()),
        "frame_to_frame_step_p99_first_100": {"this": float(np.quantile(d_got, 0.99)), "reference_run": float(np.quantile(d_ref, 0.99))},
        "parity_note": "7 unknowns, 3 equations: the reference's angles are not reproducible by the reference itself "
                       "(profiles/r04_perturbation_generic.json: real scipy vs real scipy + 1 ulp), so the claw, the limits and "
                       "the smoothness of the joint series are what can be pinned; HIP == C restatement bit for bit (tests)"}
    # ---- BATCHES of generic chains: the chain queue (persistent wavefronts, lanes pull chains) against the static launch
    try:
        out["generic"]["batch"] = generic_batches(za)
    except Exception as exc:  # noqa: BLE001
        out["generic"]["batch"] = {"error": f"{type(exc).__name__}: {exc}"}
    # ---- latency floors of the two latency-bound kernels (item: "latency-bound" as a number) -------------------------
    for entry, kernel_key, live_ms in ((out["4"], "config4_serial_walk", out["4"]["default"]["ms"]),
                                       (out["generic"], "generic_rf_6000", out["generic"]["ms"])):
        fl = latency_floor(kernel_key)
        if fl:
            keep = ("kernel", "issue_floor_ms", "latency_floor_frac", "critical_stage", "kernel_ms", "kernel_ms_lane_pairs_on",
                    "kernel_ms_lane_pairs_off", "valu_insts_per_frame")
            entry["latency_floor"] = {k: fl[k] for k in keep if k in fl}
            entry["latency_floor"]["source"] = f"profiles/{LATENCY_ROUND}_latency_floor.json (rocprofv3 PMC instruction counts of one wavefront x lone-wavefront issue costs)"
            # live: the committed floor against THIS run's whole call (upload + kernel + download, host clock)
            entry["latency_floor_frac"] = fl["issue_floor_ms"] / live_ms
    # ---- config 5: streamed from pinned host slabs, alignment fused, PCIe-inclusive ----------------------------------
    spec = importlib.util.spec_from_file_location("stream_config5", os.path.join(ROOT, "scripts", "stream_config5.py"))
    sc5 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc5)
    from types import SimpleNamespace
    c5 = {"workload": "config 5: 10M frames x 6 legs streamed from pinned host memory in slabs, RAW key points, "
                      "AlignPose.align_leg fused into the kernel prologue; PCIe-inclusive (H2D 120 B, D2H 56 + 216 B per leg-frame)"}
    left = time_box_s - (time.perf_counter() - t_start)
    try:
        a5 = SimpleNamespace(frames=10_000_000, slab_frames=1_000_000, slots=3, no_fk=False)
        c5["one_recording"] = sc5.one_recording(a5)
        left = time_box_s - (time.perf_counter() - t_start)
        # the synthetic iid sequences of the headline (64 frames each): data generation costs ~6 s per distinct 1M-frame
        # slab on the host, so ONE distinct slab is generated and cycled (the kernels cannot tell); sized to the time left
        if left > 60:
            # gpu_stats: also pass 1 of config 5 -- AlignPose's whole-recording order statistics (the constants of the fused
            # affine) extracted and sorted on the GPU from the RAW slabs
            a5s = SimpleNamespace(frames=10_000_000, slab_frames=500_000, frames_per_seq=64, unique=1, slots=3, no_fk=False,
                                  pageable=False, check=True, gpu_stats=True)
            c5["synthetic_sequences"] = sc5.synthetic_sequences(a5s)
        else:
            c5["synthetic_sequences"] = {"skipped": f"time box: {left:.0f} s left"}
    except Exception as exc:  # noqa: BLE001  (pinned-memory limits of a box must not take the headline down)
        c5["error"] = f"{type(exc).__name__}: {exc}"
    out["5"] = c5
    out["seconds"] = time.perf_counter() - t_start
    return out


class Lifeline:
    """What an N > 1 run prints if it gets stuck.  Everything such a run does is a collective over the ranks, and a rank that
    fails where the others do not leaves them waiting for ever: the measurement that IS already made must still come out.
    Every rank arms the same deadline at the same points of the program (behind a collective); when it passes, rank 0 prints
    the best line there is so far -- `line_fn()` -- and every rank leaves with exit code 0."""

    def __init__(self, rank, json_fd):
        import threading
        self.rank, self.json_fd, self.deadline, self.line_fn, self.what = rank, json_fd, None, None, ""
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def arm(self, seconds, line_fn, what):
        self.line_fn, self.what, self.deadline = line_fn, what, time.time() + seconds

    def disarm(self):
        self.deadline = None

    def _watch(self):
        while True:
            time.sleep(0.25)
            d = self.deadline
            if d is not None and time.time() > d:
                try:
                    if self.rank == 0 and self.line_fn is not None:
                        os.write(self.json_fd, (json.dumps(self.line_fn()) + "\n").encode())
                    sys.stderr.write(f"bench.py rank {self.rank}: {self.what} did not finish in time -- the line measured so far "
                                     "is printed, leaving\n")
                finally:
                    os._exit(0)


def main():
    args = parse()
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when its communicator
    # is created, so everything up to the final print goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (the launcher starts one rank per GPU)")
    n_dev = torch.cuda.device_count()
    device_index = local_rank % max(n_dev, 1)  # (more ranks than GPUs only happens in the one-GPU rehearsal)
    torch.cuda.set_device(device_index)
    dist = None
    # SEQIK_BENCH_FORCE_DIST=1: run the process-group + gather path with a single rank too (rehearsal of the RCCL
    # code path on a one-GPU box; the gather is then a device-to-device copy)
    use_dist = world > 1 or os.environ.get("SEQIK_BENCH_FORCE_DIST") == "1"
    backend = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" = RCCL over xGMI, one GPU per rank.  Ranks that have to share a GPU (rehearsal on a one-GPU box: RCCL
        # refuses two ranks on one device) talk over gloo; the solver, the peer-write gather and the timing are the same.
        backend = os.environ.get("SEQIK_BENCH_BACKEND") or ("nccl" if n_dev >= world else "gloo")
        if backend == "nccl":
            # the solver keeps every CU busy for the whole step: give RCCL's stream priority so that the gather's
            # few workgroups are dispatched as soon as a slot frees up instead of behind the queued solver waves
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index),
                                    pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    if args.one_recording:
        # config 3 read literally is the whole job of this run: one recording, frame-sharded over the ranks
        leg = one_recording_leg(dist, world, rank, args.frames, args.steps, args.warmup, coll_dev)
        if rank == 0:
            units_rank0 = 6 * leg["frames_per_rank"][0]
            spec = leg["speculative_pass_ms_this_rank"]
            ach = BYTES_PATH * units_rank0 / (spec * 1e-3) / 1e9 if spec else None
            out = {"metric": "leg-IK solves/s (frames x 6 legs); max |d theta| vs reference in `check`",
                   "value": leg["value"], "unit": "leg-frame solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "config 3 literally: ONE recording of %d frames x 6 legs, contiguous frame slabs over "
                                          "the ranks (library frame chunks, end-state exchange, angle all-gather)" % args.frames,
                              "parallelism": f"frame-sharded x{world}" if world > 1 else "1 GPU",
                              "backend": backend, **{k: leg[k] for k in ("frames_per_rank", "frames_per_chunk", "run_in_frames",
                                                                          "boundary_rounds", "data", "exchange")}},
                   "roofline": {"bound": "hbm", "kernel": "seqik_chunk_kernel<true, SPEC> (speculative pass of rank 0's slab)",
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None,
                                "traffic": None, "avg_launch_ms": spec, "bytes_per_unit": BYTES_PATH,
                                "note": "algorithmic bytes of rank 0's slab / duration of its speculative pass (HIP events on the "
                                        "launch stream); the path is FP64-issue-bound, see the default run's roofline"},
                   "check": leg["check"]}
            sys.stdout.flush()
            os.dup2(json_fd, 1)
            print(json.dumps(out), flush=True)
            os.dup2(2, 1)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    T = args.frames_per_seq
    S_total = args.frames // T

    whole = {}   # rank 0 of an N > 1 job keeps the whole fixed problem: the one-GPU reference of the same run

    def workload_for(scaling):
        """(pose of this rank, legs, body, params, leg-frames per step over all ranks)"""
        lo, hi, S_job = sharding.rank_share(S_total, world, rank, scaling)
        if scaling == "strong":
            # the fixed problem: S_total sequences, generated identically on every rank, rank r solves its slice
            legs_, body_, pose_all, params_ = make_workload(S_total, T, args.variant, synthetic.SEED_BASE)
            pose_ = pose_all[lo:hi]
            if rank == 0 and world > 1:
                whole["pose"] = pose_all
            del pose_all
        else:
            legs_, body_, pose_, params_ = make_workload(S_total, T, args.variant, synthetic.SEED_BASE + 1000 * rank)
            if rank == 0 and world > 1 and "pose" not in whole:
                whole["pose"] = pose_       # rank 0's weak batch IS the fixed problem (same seed)
        return pose_, legs_, body_, params_, S_job * len(legs_) * T

    pose, legs, body, params, units_all = workload_for(args.scaling)
    S = pose.shape[0]
    L = len(legs)
    explicit_depth = args.streams > 0
    first_depth = (args.streams, max(0, args.stage_pipeline)) if explicit_depth else DEPTH_CANDIDATES[0][:2]
    # every rank's angle blocks have the size of the LARGEST share (the gather moves equal blocks; the solver fills this
    # rank's own sequences, the padding -- at most one sequence -- stays zero)
    s_pad = max(sharding.rank_share(S_total, world, r, args.scaling)[1] - sharding.rank_share(S_total, world, r, args.scaling)[0]
                for r in range(world))
    batch = Batch(pose, params, args, first_depth[0], pipeline=first_depth[1], s_pad=s_pad)
    units_per_step = batch.units  # leg-frames per step on this rank
    main_stream = batch.main
    # final joint-angle gather: chosen below (choose_gather), once timed_region exists
    gather, gather_how, gather_calibration = None, None, None
    depth_calibration = None

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(bt, bufs, g, steps, warmup, events=None):
        """W untimed + K timed steps of batch `bt` (round-robin over its streams, gather `g` per step when there is
        one), bracketed by barrier + synchronize.  Returns (max over ranks, this rank's) seconds."""
        nb = len(bufs)

        def step(i, evs=None, tail=False):
            b = i % nb
            with torch.cuda.stream(bt.streams[i % len(bt.streams)]):
                if g:
                    g.wait_buffer(b)  # the gather that last read this buffer has completed
                bt.launch(i, bufs[b], evs, tail=tail)
                if g:
                    g.submit(b, bufs[b])

        setup_streams(bt, bufs, len(bt.streams))     # allocations of the library, once per stream: outside every timed region
        for i in range(warmup):
            step(i)
        if g:
            g.drain()
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, events[i] if events else None, tail=in_lat_range(bt, i))
        if g:
            g.drain()
        sync_all()
        mine = time.perf_counter() - t0
        tmax = mine
        if dist:
            t = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tmax = float(t.item())
        return tmax, mine

    def buffers_for(bt):
        # angle buffers: one per step in flight + two spare, so that a gather that is still draining (it only gets CU slots
        # as solver waves retire) does not hold back the launch that wants to reuse its buffer
        return [bt.angle_buffer() for _ in range(max(2, len(bt.streams) + (2 if use_dist else 0)))]

    lifeline = Lifeline(rank, json_fd) if (use_dist and world > 1) else None
    if lifeline is not None:
        # A PROVISIONAL headline first, on the plainest path there is (3 steps in flight, the library's kernel choice, grouped
        # RCCL point-to-point as the gather): should a calibration, a gather probe or a leg below ever get stuck, this is the
        # line that comes out (marked provisional) instead of nothing.
        bufs0 = buffers_for(batch)
        g0 = sharding.GatherPipeline(dist, world, rank, bufs0[0], dst=0, n_buffers=len(bufs0))
        tm0, _ = timed_region(batch, bufs0, g0, args.steps, args.warmup)
        prov = {"metric": "leg-IK solves/s (frames x 6 legs)", "value": units_all * args.steps / tm0, "unit": "leg-frame solves/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": tm0 / args.steps * 1e3,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"config 3, {args.scaling} scaling over {world} ranks ({S_total} sequences of {T} frames x 6 legs"
                                       f"{' per GPU' if args.scaling == 'weak' else ' in total'}), PROVISIONAL measurement: 3 steps in flight, "
                                       "library's kernel choice, grouped RCCL point-to-point gather -- the run got stuck behind it",
                           "provisional": True, "streams": len(batch.streams), "sequences_per_gpu": S, "legs": L, "frames_per_sequence": T},
                "roofline": {"bound": "hbm", "achieved": BYTES_PATH * units_per_step / (tm0 / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": BYTES_PATH * units_per_step / (tm0 / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "note": "algorithmic bytes of this rank's share / wall time per step (no kernel events in the provisional line)"}}
        del bufs0, g0
        lifeline.arm(float(os.environ.get("SEQIK_BENCH_STAGE_TIMEOUT", "300")), lambda: prov, "the calibrations / the headline")

    if not explicit_depth:
        # How many steps to keep in flight, and on which kernel family: measured here, on this rank's batch, with the gather
        # running (its streams take hardware queues too), a few steps per candidate; every rank takes the candidate that is
        # fastest on the SLOWEST rank (timed_region returns the max over ranks).  One job on one GPU gains 2 % from depth
        # 16; the 1/8 share of the fixed problem 1.8 x (3.4 -> 1.9 ms per step): the lane-per-chain kernels need ~4 400
        # wavefronts in flight to fill 1 024 SIMDs, and a share only brings 183 per step.
        depth_calibration = {"candidates": [], "steps": args.steps, "warmup": args.warmup,
                             "rule": "fastest ms per step on the slowest rank over the SAME region as the headline (warm-up + steps between "
                                     "two synchronisations, fill and drain included); (streams, stage_pipeline): stage_pipeline 0 = the "
                                     "library's choice for ONE call, 1 = lane-per-chain kernels"}
        best = None
        for n_st, pipe, lat in depth_candidates(args.steps, S * L):
            try:
                bt = Batch(None, params, args, n_st, pipeline=pipe, like=batch)
                bt.lat_range = lat
                bufs = buffers_for(bt)
                g = None
                if use_dist:
                    g, _ = peer_gather.make_gather(dist, world, rank, bufs[0], n_buffers=len(bufs), min_gbps=8.0,
                                                   prefer=os.environ.get("SEQIK_GATHER") or ("rccl" if backend == "nccl" else None))
                # exactly the region that will be measured: args.warmup untimed + args.steps timed steps between two full
                # synchronisations -- the pipeline's fill and drain are INSIDE it, so with few steps a deep pipeline loses what
                # it gains in steady state (20 steps of the whole problem: 3 in flight 11.8 ms per step, 12 in flight 15.5)
                k = args.steps
                tm, _ = timed_region(bt, bufs, g, k, args.warmup)
                if g is not None and hasattr(g, "close"):
                    g.close()
                ms = tm / k * 1e3
                ok_flag = 1.0
            except Exception as exc:  # noqa: BLE001  (e.g. no memory for 16 FK buffers: the candidate is skipped on all ranks)
                sys.stderr.write(f"bench.py rank {rank}: depth candidate {(n_st, pipe)} failed ({type(exc).__name__}: {exc})\n")
                bt, bufs, ms, ok_flag = None, None, float("inf"), 0.0
            if use_dist:
                okt = torch.tensor([ok_flag], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                ok_flag = float(okt.item())
            depth_calibration["candidates"].append({"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat,
                                                    "ms_per_step": ms if ok_flag > 0.5 and ms != float("inf") else None})
            if ok_flag > 0.5 and (best is None or ms < best[0]):
                best = (ms, n_st, pipe, bt, lat)
            del bufs
        if best is None:
            raise SystemExit("bench: no pipeline depth could be run")
        depth_calibration["chosen"] = {"streams": best[1], "stage_pipeline": best[2], "latency_kernel_steps": best[4]}
        batch = best[3]
        del batch.d_fks[len(batch.streams):]
        torch.cuda.empty_cache()
    streams = batch.streams
    n_streams = len(streams)
    n_buf = max(2, len(streams) + (2 if use_dist else 0))
    d_ang = [batch.angle_buffer() for _ in range(n_buf)]

    def choose_gather():
        """The final joint-angle gather of the headline.  BASELINE.json's north star names it: "RCCL over xGMI only for the
        final joint-angle gather" -- so on a real multi-GPU job (process group on RCCL) grouped RCCL point-to-point is the
        DEFAULT, and the copy-engine peer writes (seqikpy_amd/peer_gather.py: no compute unit busy on either GPU) replace
        it only when a short calibration IN THIS RUN -- the same batch, the same process group, 6 steps each -- measures
        them at least 5 % faster on the slowest rank (they never saw two GPUs before the driver's scaling run, so the
        choice is made from a measurement, not from the one-GPU rehearsal).  Both figures go into config.gather.
        SEQIK_GATHER=rccl|peer forces one.  Ranks that share a GPU (rehearsal, process group on gloo): peer writes, as
        gloo would stage every block through the host.  -> (pipeline, description, calibration or None)"""
        if not use_dist:
            return None, None, None
        rehearse = os.environ.get("SEQIK_BENCH_CALIBRATE_GATHER") == "1"   # run the calibration on a gloo rehearsal too (tests)
        if os.environ.get("SEQIK_GATHER") or world == 1 or (backend != "nccl" and not rehearse):
            g, how = peer_gather.make_gather(dist, world, rank, d_ang[0], n_buffers=n_buf, min_gbps=8.0)
            return g, how, None
        k_cal, cal, cands = 6, {}, {}
        for how in ("rccl", "peer"):
            g2, desc = peer_gather.make_gather(dist, world, rank, d_ang[0], n_buffers=n_buf, prefer=how, min_gbps=8.0)
            if how == "peer" and not isinstance(g2, peer_gather.PeerWriteGather):
                cal["peer"] = {"unavailable": desc}     # it fell back to the RCCL pipeline on every rank alike
                if hasattr(g2, "close"):
                    g2.close()
                continue
            tm, _ = timed_region(batch, d_ang, g2, k_cal, 2)        # max over ranks: every rank sees the same figure
            cal[how] = {"ms_per_step": tm / k_cal * 1e3, "ran_as": desc}
            cands[how] = (g2, desc, tm)
        pick = "peer" if ("peer" in cands and cands["peer"][2] < 0.95 * cands["rccl"][2]) else "rccl"
        for how, (g2, _, _) in cands.items():
            if how != pick and hasattr(g2, "close"):
                g2.close()
        cal["rule"] = "RCCL point-to-point (the north star's gather) unless peer writes are >= 5 % faster in this calibration"
        cal["chosen"] = pick
        return cands[pick][0], cands[pick][1], cal

    try:
        gather, gather_how, gather_calibration = choose_gather()
        chose = True
    except Exception as exc:  # noqa: BLE001  (e.g. no memory for a second set of receive buffers)
        sys.stderr.write(f"bench.py rank {rank}: gather calibration failed ({type(exc).__name__}: {exc})\n")
        chose = False
    if use_dist:
        # every rank must end up with the same kind of pipeline: if the calibration failed anywhere, all ranks take the plain
        # RCCL point-to-point pipeline (the north star's gather)
        ok = torch.tensor([1.0 if chose else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 0.5:
            if chose and hasattr(gather, "close"):
                gather.close()
            gather = sharding.GatherPipeline(dist, world, rank, d_ang[0], dst=0, n_buffers=n_buf)
            gather_how, gather_calibration = "grouped RCCL point-to-point (calibration failed on a rank)", None

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    for row in ev:          # torch creates the underlying hipEvent_t lazily, on the first record()
        for e in row:
            e.record(main_stream)

    elapsed, elapsed_mine = timed_region(batch, d_ang, gather, args.steps, args.warmup, ev)

    # Outside the timed region: every buffer the overlapped launches wrote must hold, bit for bit, what one launch
    # made alone writes (all steps solve the same batch) -- a measurement of launches that disturbed each other
    # would be worthless.
    chk_ang, chk_fk = torch.zeros_like(d_ang[0]), torch.zeros_like(batch.d_fks[0])
    _lib.solve_seq_device(batch.d_pose.data_ptr(), S, L, T, params, chk_ang.data_ptr(), chk_fk.data_ptr(),
                          stream=main_stream.cuda_stream, block_size=args.block, layout=batch.layout,
                          lanes_per_wave=args.lanes_per_wave, staged=int(args.staged), interleave_legs=args.interleave_legs)
    torch.cuda.synchronize()
    n_used = max(args.steps, args.warmup)  # warm-up and timed steps both count from 0
    if not all(torch.equal(d_ang[b], chk_ang) for b in range(min(n_buf, n_used))) or \
            not all(torch.equal(f, chk_fk) for f in batch.d_fks[:min(len(batch.d_fks), n_used)]):
        raise SystemExit("bench: overlapped launches did not reproduce a launch made alone -- result invalid")
    del chk_ang, chk_fk

    # per-kernel durations from the HIP events recorded on the launch stream inside the timed region
    # (the steps of `lat_range` run another kernel: the dominant kernel's figures come from the other steps)
    main_steps = [i for i in range(args.steps) if not in_lat_range(batch, i)]
    stage_ms = np.array([[ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(4)] for i in main_steps])
    mean_stage_ms = stage_ms.mean(0)
    ms_per_step = elapsed / args.steps * 1e3
    if args.staged:
        dom = int(np.argmax(mean_stage_ms)) + 1
        kname, key, bytes_unit, dom_ms = f"seqik_stage_kernel<{dom}, ...>", f"stage{dom}", BYTES_STAGE[dom], float(mean_stage_ms[dom - 1])
    else:  # one kernel per step: event [0] is recorded in front of it, [1] behind it
        kname, key, bytes_unit, dom_ms = "seqik_fused_kernel<true>", "fused", BYTES_PATH, float(mean_stage_ms[0])
    ach_gbs = bytes_unit * units_per_step / (dom_ms * 1e-3) / 1e9
    traffic, valu, fp64, pmc_matches_build, pmc_file = pmc_roofline(args.variant, args.staged, key, units_per_step, ms_per_step, device_index)
    hbm = {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
           "bytes_per_unit": bytes_unit,
           "note": "algorithmic bytes x units per launch / the kernel's average launch duration (launches of "
                   f"{len(streams)} steps overlap, so a launch lasts ~{len(streams)}x a step); HBM is not what binds: "
                   "~1e4 f64 instructions per 392 B"}
    if fp64:
        # (three significant digits: the lane share behind `flops_per_step` is that of ALL vector instructions -- the FP64
        # ones sit mostly in the dense body of a pass, 49-52 lanes, so the figure is if anything low -- not a count)
        tfl = float(f"{fp64['flops_per_step'] / (ms_per_step * 1e-3) / 1e12:.3g}")
        roofline = {"bound": "valu-fp64", "kernel": kname, "achieved": tfl, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                    "frac": float(f"{tfl / FP64_VECTOR_PEAK_TF:.3g}"), "traffic": traffic, "avg_launch_ms": dom_ms,
                    "fp64": fp64, "valu_issue": valu, "hbm": hbm,
                    "note": "the path is bound by FP64 VALU issue, not by HBM or MFMA: `achieved` = FP64 operations "
                            "actually performed by active lanes per second (PMC instruction mix x lane share, live "
                            "timing) against the vector FP64 peak; `valu_issue` = how close the step is to the floor its "
                            "wave-instruction count allows; `hbm` = the algorithmic-bytes figure"}
    else:  # no PMC summary of THIS build for this workload: only the HBM figure can be stated
        roofline = {"bound": "hbm", "kernel": kname, **hbm, "traffic": traffic, "avg_launch_ms": dom_ms,
                    "note": hbm["note"] + (" (the PMC summary under profiles/ was measured on other kernel sources than this "
                                           "build's: the VALU figures are withheld)" if pmc_matches_build is False else
                                           " (no PMC summary under profiles/ matches this workload, so the VALU figures are absent)")}
    roofline["pmc_matches_build"] = pmc_matches_build
    roofline["pmc_file"] = pmc_file
    if args.staged:
        roofline["stage_ms"] = [float(v) for v in mean_stage_ms]
        roofline["path_GBps"] = BYTES_PATH * units_per_step / (mean_stage_ms.sum() * 1e-3) / 1e9

    frames_txt = "1M" if S_total * T == 1_000_000 else f"{S_total * T:,}".replace(",", " ")

    def make_out():
        """The JSON line's headline part (everything that is known once the timed region and the roofline are done)."""
        return {
            "metric": "leg-IK solves/s (frames x 6 legs); max |d theta| vs reference in `parity`",
            "value": units_all * args.steps / elapsed,
            "unit": "leg-frame solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"config 3: synthetic {frames_txt} frames x 6 legs, in-workspace targets" if world == 1 else
                                    f"config 3 (weak-scaling variant): synthetic {frames_txt} frames x 6 legs PER GPU, in-workspace targets"
                                    if args.scaling == "weak" else
                                    f"config 3 literally: the FIXED problem of synthetic {frames_txt} frames x 6 legs IN TOTAL ({S_total} "
                                    f"sequences of {T} frames), sequences split over the {world} ranks, joint angles gathered on rank 0"),
                       "frames_total": S_total * T * (world if args.scaling == "weak" else 1),
                       "leg_frames_per_step_all_ranks": int(units_all),
                       "variant": args.variant, "frames_per_gpu": S * T, "legs": L, "sequences_per_gpu": S,
                       "frames_per_sequence": T, "chains_per_gpu": S * L, "warm_start": "previous frame",
                       "outputs": "7 angles + 9x3 FK per leg-frame", "device_layout": "planar",
                       "streams": len(streams),
                       "pipeline": f"{len(streams)} independent batches in flight (consecutive steps overlap); "
                                   "`single_job` is one launch at a time",
                       "launches_per_step": 4 if args.staged else 1,
                       "parity_note": ("iid poses span several equivalent leg configurations: the reference itself moves 30 % of "
                                       "such leg-frames by more than 1e-4 rad under a 1-ulp change of its input "
                                       "(profiles/r02_perturbation_report.json), so on this variant parity means HIP == C restatement "
                                       "bit for bit; `variants.smooth` is the realistic workload, `parity` the shipped recordings"
                                       if args.variant == "iid" else
                                       "temporally continuous poses (the realistic variant); `parity` holds the shipped recordings"),
                       "parallelism": f"sequence-sharded x{world}, angle gather to rank 0" if world > 1 else "1 GPU",
                       "stage_pipeline": batch.pipeline,
                       **({"depth_calibration": depth_calibration} if depth_calibration else {}),
                       **({"gather": gather_how} if gather_how else {}),
                       **({"gather_calibration": gather_calibration} if gather_calibration else {})},
            "roofline": roofline,
            "verified": "after timing: every angle / FK buffer written by the overlapped launches == one launch made "
                        "alone, bit for bit (smoke() and tests/ compare that launch with the oracle)",
        }

    # ---- N > 1: who took part, how even the ranks were, both gathers, and the other scaling mode -- in the same run ----
    multi = None
    if dist and world > 1:
        import socket
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "host": socket.gethostname(), "device": device_index,
                                      "pid": os.getpid(), "ms_per_step": elapsed_mine / args.steps * 1e3})
        multi = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (ranks share a GPU: rehearsal)"),
                 "ranks_seen": [{k: r[k] for k in ("rank", "host", "device")} for r in seen],
                 "rank_ms_per_step": {"min": min(r["ms_per_step"] for r in seen), "max": max(r["ms_per_step"] for r in seen),
                                      "by_rank": [r["ms_per_step"] for r in seen]}}
        if not args.no_extras:
            # The headline is measured.  The legs below are collectives over all ranks: if one of them ever hangs (a rank that
            # fails where the others do not and never enters the collective they wait in), the line must still come out --
            # the lifeline (armed on every rank alike) prints the headline with the legs finished so far.
            legs_deadline_s = float(os.environ.get("SEQIK_BENCH_LEGS_TIMEOUT", "300"))

            def line_with_legs_so_far():
                o = make_out()
                try:
                    o["multi_gpu"] = dict(multi, legs_timed_out_after_s=legs_deadline_s)
                except RuntimeError:     # (the main thread was adding a leg at this very moment)
                    o["multi_gpu"] = {"legs_timed_out_after_s": legs_deadline_s}
                return o

            lifeline.arm(legs_deadline_s, line_with_legs_so_far, "the extra legs")

            def guarded(name, fn):
                """One extra leg.  The headline above is measured and must survive whatever happens here: a leg that
                raises on any rank is dropped on ALL ranks (consensus by all-reduce, so nobody waits in a collective the
                others never enter) and reported as an error string."""
                err = None
                try:
                    res = fn()
                except Exception as exc:  # noqa: BLE001
                    res, err = None, f"{type(exc).__name__}: {exc}"
                ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) < 0.5:
                    multi[name] = {"error": err or "failed on another rank"}
                else:
                    multi[name] = res

            def rank_ms(mine_s, k):
                """this leg's per-step time of every rank (ms), gathered over the process group"""
                got = [None] * world
                dist.all_gather_object(got, mine_s / k * 1e3)
                return {"min": min(got), "max": max(got), "by_rank": got}

            k_cmp = max(4, min(20, args.steps))
            headline_kind = "peer" if isinstance(gather, peer_gather.PeerWriteGather) else "rccl"

            def leg_n1_reference():
                # the WHOLE fixed problem (config 3: 1M frames x 6 legs) on rank 0's GPU alone, same pipeline, no gather (one
                # GPU has nobody to gather from), while the other ranks wait in the leg's consensus all-reduce: what the
                # scaling efficiencies of THIS run are measured against
                res = None
                if rank == 0:
                    # at ITS best depth (what an N = 1 run calibrates for itself), not at the shares': a reference that is
                    # slower than it could be would flatter the efficiencies
                    k1, first, res = args.steps, None, None
                    for n_st, pipe, lat in depth_candidates(args.steps, S_total * L):
                        if lat is not None:
                            continue
                        b1 = Batch(whole["pose"], params, args, n_st, pipeline=pipe, like=first)
                        first = first or b1
                        bufs1 = [b1.angle_buffer() for _ in range(len(b1.streams))]
                        dt1 = timed_steps(b1, bufs1, k1, len(b1.streams), warmup=args.warmup)
                        if res is None or dt1 / k1 * 1e3 < res["ms_per_step"]:
                            res = {"value": b1.units * k1 / dt1, "unit": "leg-frame solves/s", "ms_per_step": dt1 / k1 * 1e3, "steps": k1,
                                   "streams": len(b1.streams), "stage_pipeline": pipe, "leg_frames_per_step": int(b1.units),
                                   "what": "the whole fixed problem on rank 0's GPU alone (the other ranks idle), same run, at the "
                                           "best of the depths an N = 1 run calibrates among"}
                        del bufs1
                    del b1, first
                    torch.cuda.empty_cache()
                got = [None] * world
                dist.all_gather_object(got, res)
                return got[0]

            def leg_gather_compare():
                # the final joint-angle gather, both ways, same batch, same process group: copy-engine peer writes into
                # rank 0's exported buffers vs grouped RCCL point-to-point (the north star's "RCCL over xGMI")
                cmp_ = {"steps": k_cmp}
                for how in ("peer", "rccl"):
                    g2, desc = peer_gather.make_gather(dist, world, rank, d_ang[0], n_buffers=n_buf, prefer=how)
                    tm, _ = timed_region(batch, d_ang, g2, k_cmp, min(2, args.warmup))
                    cmp_[how] = {"ms_per_step": tm / k_cmp * 1e3, "value": units_all * k_cmp / tm, "ran_as": desc}
                    if hasattr(g2, "close"):
                        g2.close()
                    del g2
                tm, _ = timed_region(batch, d_ang, None, k_cmp, min(2, args.warmup))
                cmp_["no_gather"] = {"ms_per_step": tm / k_cmp * 1e3, "value": units_all * k_cmp / tm}
                return cmp_

            other_scaling = "strong" if args.scaling == "weak" else "weak"

            def leg_other_scaling():
                # the other scaling mode beside the headline: strong = config 3 literally (the fixed 1M-frame problem split
                # over the ranks), weak = 1M frames per GPU
                pose2, _, _, _, units_all2 = workload_for(other_scaling)
                pad2 = max(sharding.rank_share(S_total, world, r, other_scaling)[1] - sharding.rank_share(S_total, world, r, other_scaling)[0]
                           for r in range(world))
                # (its own depth: the headline's may be 20 steps of a 1/8 share in flight; 20 whole problems are 22 x 7 x 336 MB of
                # receive buffers on rank 0 for nothing -- a batch that fills the GPU runs 3 deep on the library's kernel choice)
                n_st2 = min(n_streams, depth_cap(int(pose2.shape[0]) * L))
                pipe2 = batch.pipeline if n_st2 > 3 else 0
                n_buf2 = max(2, n_st2 + 2)
                b2 = Batch(pose2, params, args, n_st2, pipeline=pipe2, s_pad=pad2)
                bufs2 = [b2.angle_buffer() for _ in range(n_buf2)]
                g2, desc = peer_gather.make_gather(dist, world, rank, bufs2[0], n_buffers=n_buf2, min_gbps=8.0, prefer=headline_kind)
                k2 = max(4, min(40, args.steps))
                tm, mine2 = timed_region(b2, bufs2, g2, k2, min(3, args.warmup))
                res = {"value": units_all2 * k2 / tm, "unit": "leg-frame solves/s", "ms_per_step": tm / k2 * 1e3,
                       "steps": k2, "scaling": other_scaling, "streams": n_st2, "stage_pipeline": pipe2, "sequences_per_gpu": int(pose2.shape[0]),
                       "leg_frames_per_step_all_ranks": int(units_all2), "gather": desc, "rank_ms_per_step": rank_ms(mine2, k2)}
                if hasattr(g2, "close"):
                    g2.close()
                return res

            def leg_one_recording():
                # config 3 read literally: ONE recording frame-sharded over the ranks (`--one-recording` runs it alone), and
                # the same recording on rank 0's GPU alone (frame chunks, no exchange) as its one-GPU reference
                torch.cuda.empty_cache()
                res = one_recording_leg(dist, world, rank, args.frames, max(3, min(10, args.steps)), 1, coll_dev)
                n1 = None
                if rank == 0:
                    torch.cuda.empty_cache()
                    n1 = single_recording(args.frames, steps=4)["ms_per_step"]
                got = [None] * world
                dist.all_gather_object(got, n1)
                res["n1_reference_ms"] = got[0]
                res["efficiency_vs_n1"] = got[0] / res["ms_per_step"] / world
                return res

            def leg_config5():
                # BASELINE config 5 on N GPUs: 10 M frames x 6 legs streamed from pinned host memory with the alignment fused,
                # PCIe-inclusive, every rank over its own PCIe link.  (a) independent 64-frame sequences: rank r streams its 1/N
                # of them, no coordination; (b) ONE recording, contiguous slabs per rank, warm start carried from slab to slab
                # and across the rank boundaries (stream_sharding), whole-recording alignment statistics on every rank.
                import importlib.util
                from types import SimpleNamespace
                spec = importlib.util.spec_from_file_location("stream_config5", os.path.join(ROOT, "scripts", "stream_config5.py"))
                sc5 = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(sc5)
                torch.cuda.empty_cache()
                frames5 = int(os.environ.get("SEQIK_BENCH_CONFIG5_FRAMES", "10000000"))
                slab = min(500_000, max(64, (frames5 // world // 64) * 64))
                a5 = SimpleNamespace(frames=max(slab, frames5 // world), slab_frames=slab, frames_per_seq=64, unique=1, slots=3, no_fk=False,
                                     pageable=False, check=False, gpu_stats=False)
                sync_all()
                mine = sc5.synthetic_sequences(a5)
                got = [None] * world
                dist.all_gather_object(got, {"seconds": mine["seconds"], "leg_frames": mine["leg_frames"], "pcie_GBps": mine["pcie_GBps_total"]})
                seq = {"value": sum(g_["leg_frames"] for g_ in got) / max(g_["seconds"] for g_ in got), "unit": "leg-frame solves/s",
                       "leg_frames": sum(g_["leg_frames"] for g_ in got), "seconds_slowest_rank": max(g_["seconds"] for g_ in got),
                       "by_rank": got, "what": "independent 64-frame sequences, 1/N of them per rank, each rank over its own PCIe link; "
                                               "PCIe-inclusive (H2D 120 B, D2H 272 B per leg-frame), alignment fused"}
                slab_r = max(1000, (frames5 // (2 * world) // 1000) * 1000)    # two slabs per rank; a multiple of the fixture's 1000 frames
                a5r = SimpleNamespace(frames=frames5, slab_frames=slab_r, slots=3, no_fk=False, gpu_stats=False)
                rec = sc5.one_recording_over_ranks_core(a5r, dist, world, rank, backend)
                box = [rec]
                dist.broadcast_object_list(box, src=0)
                return {"workload": "config 5: %d frames x 6 legs streamed from pinned host memory, AlignPose.align_leg fused, N GPUs" % frames5,
                        "synthetic_sequences": seq, "one_recording": box[0]}

            guarded("n1_reference", leg_n1_reference)
            guarded("gather_compare", leg_gather_compare)
            guarded(other_scaling, leg_other_scaling)
            guarded("one_recording", leg_one_recording)
            guarded("config5", leg_config5)
            # scaling efficiencies against the one-GPU run of the SAME job in the SAME run: value / (N x value at N = 1)
            n1 = multi.get("n1_reference")
            if n1 and "value" in n1:
                multi["efficiency_vs_n1"] = (units_all * args.steps / elapsed) / (world * n1["value"])
                multi["speedup_vs_n1"] = (units_all * args.steps / elapsed) / n1["value"]
                oth = multi.get(other_scaling)
                if oth and "value" in oth:
                    oth["efficiency_vs_n1"] = oth["value"] / (world * n1["value"])
    if lifeline is not None:
        lifeline.disarm()

    if rank == 0:
        out = make_out()
        if multi:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_extras:
            # (the headline above is measured and must be printed whatever happens below: a leg that raises ends the extras,
            # the line carries what was finished and `extras_error`)
            try:
                # ---- one launch at a time --------------------------------------------------------------------------
                n1 = max(4, min(16, args.steps // 6))
                lone = Batch(None, params, args, 1, pipeline=0, like=batch)       # ONE call at a time: the library's own kernel choice
                dt1 = timed_steps(lone, d_ang, n1, 1)
                out["single_job"] = {"value": units_per_step * n1 / dt1, "unit": "leg-frame solves/s", "ms_per_step": dt1 / n1 * 1e3,
                                     "steps": n1, "streams": 1,
                                     "note": "one 1M-frame x 6-leg batch at a time: 1 465 full waves on 1 024 SIMDs cannot hide "
                                             "FP64 latency; `value` above is the pipelined rate"}
                # ---- fixed 1M-frame problem split N ways: the per-rank share timed on this GPU, BEFORE the legs that create streams of
                # their own (config 5's pipeline, the pooled contexts of the host-buffer calls): run behind them the same pipelines
                # share queues with those streams (1/8 share 2.1 -> 4.3 ms per step) ---------------------------------------------
                proj = {"note": "per-rank share of the fixed problem (S/N sequences) timed on ONE GPU; no gather; "
                                "projected_value = 6M leg-frames / that time.  `streams` / `stage_pipeline`: the fastest of the depth "
                                "candidates for that share (what an N-GPU run calibrates for itself): a share of 1/N brings 1/N of the "
                                "wavefronts per step, so as many more steps must be in flight to fill the GPU; lone_job_ms = ONE launch at "
                                "a time (the library's own kernel choice), what a rank gets when every step waits for the one before",
                        "by_n_gpus": {}}
                for n in (2, 4, 8):
                    best, tried = None, []
                    for n_st, pipe, lat in (depth_candidates(args.steps, (S // n) * L) if not explicit_depth else ((n_streams, batch.pipeline, None),)):
                        sub = Batch(pose[: S // n], params, args, n_st, pipeline=pipe)
                        sub.lat_range = lat
                        bufs = [sub.angle_buffer() for _ in range(len(sub.streams))]
                        k = args.steps             # the same region as the headline's: fill and drain of the pipeline included
                        dt = timed_steps(sub, bufs, k, len(sub.streams), warmup=args.warmup)
                        row = {"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat, "ms_per_step": dt / k * 1e3, "projected_value": units_per_step / (dt / k),
                               "speedup_vs_1": (elapsed / args.steps) / (dt / k), "chains_per_gpu": sub.S * L}
                        tried.append({"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat, "ms_per_step": row["ms_per_step"]})
                        if best is None or row["ms_per_step"] < best["ms_per_step"]:
                            best = row
                        del sub, bufs
                    best["candidates"] = tried
                    # the floor of a share: ONE launch at a time, no second step to overlap with (what a rank can do at best when
                    # every step has to wait for the one before it)
                    sub = Batch(pose[: S // n], params, args, 1, pipeline=0)     # ONE call: the library's own choice of kernel
                    bufs = [sub.angle_buffer()]
                    k = max(8, min(40, args.steps // 2))
                    dt = timed_steps(sub, bufs, k, 1, warmup=2)
                    best["lone_job_ms"] = dt / k * 1e3
                    best["lone_job_speedup_vs_single_job"] = out["single_job"]["ms_per_step"] / best["lone_job_ms"]
                    best["ideal_ms"] = ms_per_step / n
                    best["efficiency"] = best["speedup_vs_1"] / n
                    fl = share_floor(n)
                    if fl:
                        best["lone_job_issue_floor_ms"] = fl["issue_floor_ms"]
                        best["lone_job_issue_floor_frac"] = fl["issue_floor_ms"] / best["lone_job_ms"]
                        best["floor_source"] = fl["source"]
                    del sub, bufs
                    proj["by_n_gpus"][str(n)] = best
                out["strong_projection"] = proj
                # ---- the other synthetic variant -------------------------------------------------------------------
                other = "smooth" if args.variant == "iid" else "iid"
                _, _, pose_o, _ = make_workload(S, T, other, synthetic.SEED_BASE)
                bo = Batch(pose_o, params, args, n_streams, pipeline=batch.pipeline)
                ko = args.steps
                dto = timed_steps(bo, d_ang, ko, len(bo.streams), warmup=args.warmup)
                ms_o = dto / ko * 1e3
                _, valu_o, fp64_o, match_o, file_o = pmc_roofline(other, args.staged, key, bo.units, ms_o, device_index)
                roof_o = {"pmc_matches_build": match_o, "pmc_file": file_o}
                if fp64_o:
                    tfl_o = fp64_o["flops_per_step"] / (ms_o * 1e-3) / 1e12
                    roof_o.update({"bound": "valu-fp64", "achieved": tfl_o, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": tfl_o / FP64_VECTOR_PEAK_TF, "frac_of_valu_issue_floor": valu_o["frac_of_valu_issue_floor"],
                                   "lane_utilisation": valu_o["lane_utilisation"], "valu_insts_per_step": valu_o["valu_insts_per_step"]})
                out["variants"] = {other: {"value": bo.units * ko / dto, "unit": "leg-frame solves/s", "ms_per_step": ms_o,
                                           "steps": ko, "streams": len(bo.streams), "roofline": roof_o},
                                   "note": "smooth = temporally continuous targets (band-limited random walk): the realistic "
                                           "case; iid = every frame an unrelated pose"}
                del bo
                # ---- config 3 as ONE recording (frame chunks) ------------------------------------------------------
                del d_ang[1:]
                torch.cuda.empty_cache()
                del pose_o
                out["single_recording"] = single_recording(args.frames)
                # ---- parity vs the committed reference fixtures ----------------------------------------------------
                out["parity"] = parity_report()
                out["value_single_job"] = out["single_job"]["value"]
                # ---- every BASELINE config, reference-shaped calls, in this one line -------------------------------
                if not args.no_configs:
                    del batch, d_ang
                    torch.cuda.empty_cache()
                    out["configs"] = reference_configs()
                    # config 3 is the headline of this line: the same figures under its key, so that all five configs read alike
                    out["configs"]["3"].update({
                        "leg_frames": units_per_step, "ms_per_step_three_batches_in_flight": ms_per_step, "leg_frames_per_s": out["value"],
                        "ms_one_job_at_a_time": out["single_job"]["ms_per_step"], "leg_frames_per_s_one_job_at_a_time": out["value_single_job"],
                        "smooth_variant_leg_frames_per_s": out["variants"].get("smooth", {}).get("value"),
                        "one_recording_1M_frames_leg_frames_per_s": out["single_recording"]["value"],
                        "parity": "every buffer of the timed region == one launch made alone, bit for bit (`verified`); that launch == the C "
                                  "restatement bit for bit on sampled chains (tests/test_gpu_parity.py::test_full_size_synthetic_properties); "
                                  "against the reference itself see `parity_note` in `config`"})
                _lib.check_faults()   # the device entry points do not synchronise: a kernel fault of any launch above raises here
                # ---- a handful of scalars at the TOP level of the line: the driver's record keeps top-level scalars only --------
                def dig(obj, *path):
                    for k in path:
                        if not isinstance(obj, dict) or k not in obj:
                            return None
                        obj = obj[k]
                    return obj
                cf = out.get("configs", {})
                par = out["parity"]
                out.update({
                    "config1_default_ms": dig(cf, "1", "default", "ms"), "config1_auto_ms": dig(cf, "1", "frame_parallel_auto", "ms"),
                    "config2_default_ms": dig(cf, "2", "default", "ms"), "config2_auto_ms": dig(cf, "2", "frame_parallel_auto", "ms"),
                    "config2_64_recordings_leg_frames_per_s": dig(cf, "2", "default_64_recordings_one_call", "leg_frames_per_s"),
                    "config4_default_ms": dig(cf, "4", "default", "ms"), "config4_auto_ms": dig(cf, "4", "frame_parallel_auto", "ms"),
                    "config5_leg_frames_per_s": dig(cf, "5", "one_recording", "value"),
                    "config5_sequences_leg_frames_per_s": dig(cf, "5", "synthetic_sequences", "value"),
                    "generic_6000_frames_s": (dig(cf, "generic", "ms") or 0.0) / 1e3 or None,
                    "generic_batch_leg_frames_per_s": dig(cf, "generic", "batch", "leg_frames_per_s"),
                    "head_kernel_hbm_frac": dig(cf, "4", "head_kernel", "roofline", "frac"),
                    "head_kernel_frac_of_box_copy": dig(cf, "4", "head_kernel", "frac_of_box_copy"),
                    "parity_max_abs_dtheta": max(par[n]["serial_walk"]["max_abs_dtheta"] for n in ("anipose_shipped", "df3d_1000")),
                    "parity_p99.9_abs_dtheta": max(par[n]["serial_walk"]["p99.9_abs_dtheta"] for n in ("anipose_shipped", "df3d_1000")),
                    "parity_values_over_5e-5": sum(par[n]["serial_walk"]["values_over_5e-5"] for n in ("anipose_shipped", "df3d_1000")),
                    "parity_leg_frames_over_1e-4_outside_lf_window":
                        sum(par[n]["serial_walk"]["leg_frames_over_1e-4_outside_lf_window"] for n in ("anipose_shipped", "df3d_1000")),
                    "parity_auto_max_abs_dtheta": max(par[n]["frame_chunks"]["max_abs_dtheta"] for n in ("anipose_shipped", "df3d_1000")),
                    "smooth_variant_value": dig(out, "variants", "smooth", "value"),
                    "single_recording_value": dig(out, "single_recording", "value"),
                    "strong_share_n8_ms_per_step": dig(out, "strong_projection", "by_n_gpus", "8", "ms_per_step"),
                    "strong_share_n8_lone_job_ms": dig(out, "strong_projection", "by_n_gpus", "8", "lone_job_ms"),
                    "strong_projected_speedup_n8": dig(out, "strong_projection", "by_n_gpus", "8", "speedup_vs_1"),
                    "roofline_frac": roofline.get("frac"), "roofline_traffic_bytes": roofline.get("traffic"),
                })
            except Exception as exc:  # noqa: BLE001
                import traceback
                out["extras_error"] = f"{type(exc).__name__}: {exc}"
                sys.stderr.write("bench.py: an extra leg failed, the headline is printed without the rest:\n" + traceback.format_exc())
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(pose, legs, body, args.cpu_sample_seqs, not args.no_python_baseline)
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
                out["cpu_baseline_value"] = out["cpu_baseline"]["value"]
                out["cpu_baseline_cores"] = out["cpu_baseline"]["cores"]
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist:
        if hasattr(gather, "close"):
            gather.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
