#!/usr/bin/env python3
"""Benchmark of the sequential leg-IK hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Metric (BASELINE.json): leg-IK solves/s, one solve = one (frame, leg) = 4 stage sub-solves ->
7 joint angles (+ the stage-4 forward kinematics).  Workload at every N: BASELINE config 3,
"synthetic 1M frames x 6 legs, random in-workspace target key points", PER GPU (weak scaling):
1,000,000 frames are cut into 15,625 independent sequences of 64 frames (frame t of a sequence is
warm-started from frame t-1, frame 0 from the seeds -- the reference's semantics applied to many
recordings), 6 legs each = 93,750 chains.  A step is one pass of the hot path (one launch in which every
wave takes its chains through stages 1-4; `--staged`: the 4 stage kernels) over that batch with inputs resident in HBM; for N > 1 every step also sends the rank's joint angles to rank 0
(copy-engine peer writes over xGMI into rank 0's IPC-exported buffers, an 8-byte RCCL all-reduce as completion flag;
grouped RCCL point-to-point if the peer path is unavailable; overlapped with the next steps' kernels; `config.gather`
says which ran).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, live HIP-event timing) and, at
N = 1, `cpu_baseline` (the C oracle on the host cores, bounded sample of the same workload).
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "sequential-inverse-kinematics_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (loads the HIP runtime that libseqik_hip.so binds to)

from seqikpy_amd import _lib, data, peer_gather, sharding, synthetic, utils  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# Algorithmic HBM bytes per leg-frame (DESIGN.md "Kernels"):
BYTES_PATH = 120 + 56 + 216   # key points in, 7 angles out, 9x3 FK out
BYTES_STAGE = {1: 48 + 16 + 96, 2: 48 + 96 + 16 + 96 + 48, 3: 48 + 96 + 16 + 96 + 24, 4: 48 + 96 + 8 + 144}
# stage k reads the origin + its key point (48 B) and the 96-byte prefix frame the previous stage left in
# the workspace, writes its own angles, the next prefix frame (96 B) and its rows of the 9 x 3 FK record


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 100 steps = 1.4 s: the first and last launches of a run overlap with fewer neighbours, and with 20 steps that
    # edge still costs 5 % (20 steps 4.11e8, 400 steps 4.30e8, 1500 steps 4.31e8 solves/s)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (x 6 legs)")
    ap.add_argument("--frames-per-seq", type=int, default=64)
    ap.add_argument("--variant", default="iid", choices=["iid", "smooth"])
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--lanes-per-wave", type=int, default=0, help="chains per wavefront (0 = automatic)")
    ap.add_argument("--interleave-legs", type=int, default=0,
                    help="1 = consecutive chains per wave (legs interleaved) instead of leg-pure, longest-leg-first waves")
    ap.add_argument("--staged", action="store_true",
                    help="one launch per stage (SeqikOptions.reserved[1] = 1) instead of the default single launch in "
                         "which every wave takes its chains through the four stages in turn")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on round-robin (consecutive batches overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-python-baseline", action="store_true",
                    help="skip the Python + scipy process-pool leg of the CPU baseline (about 20 s)")
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
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
    n_dev = torch.cuda.device_count()
    device_index = local_rank % max(n_dev, 1)  # (more ranks than GPUs only happens in the gloo dry run)
    torch.cuda.set_device(device_index)
    dist = None
    # SEQIK_BENCH_FORCE_DIST=1: run the process-group + gather path with a single rank too (rehearsal of the RCCL
    # code path on a one-GPU box; the gather is then a device-to-device copy)
    use_dist = world > 1 or os.environ.get("SEQIK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SEQIK_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo": dry runs
        if backend == "nccl":
            # the solver keeps every CU busy for the whole step: give RCCL's stream priority so that the gather's
            # few workgroups are dispatched as soon as a slot frees up instead of behind the queued solver waves
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index),
                                    pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    T = args.frames_per_seq
    S = args.frames // T
    legs, body, pose, params = make_workload(S, T, args.variant, synthetic.SEED_BASE + 1000 * rank)
    L = len(legs)
    units_per_step = S * L * T  # leg-frames per GPU per step

    # planar device layout (include/seqik.h, SeqikLayout): pose [S][L][5][T][3], angles [S][L][7][T]
    layout = _lib.planar_layout(T)
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_fk = torch.zeros((S, L, T, 9, 3), dtype=torch.float64, device="cuda")
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream() for _ in range(max(0, args.streams - 1))]
    # angle buffers: one per batch in flight + two spare, so that a gather that is still draining (it only gets
    # CU slots as solver waves retire) does not hold back the launch that wants to reuse its buffer
    n_buf = len(streams) + (2 if use_dist else 0)
    n_buf = max(2, n_buf)
    d_ang = [torch.zeros((S, L, 7, T), dtype=torch.float64, device="cuda") for _ in range(n_buf)]
    d_fks = [d_fk] + [torch.zeros_like(d_fk) for _ in range(len(streams) - 1)]
    # final joint-angle gather: peer writes over xGMI when every rank can map rank 0's buffers and the copies are
    # not pathologically slow (a block needs 336 MB / 14 ms = 24 GB/s per link to keep up; a link that cannot do
    # that is no faster under RCCL, and the peer writes at least leave the root's compute units alone), grouped
    # RCCL point-to-point otherwise; the measured link rate is reported in config.gather
    gather, gather_how = (peer_gather.make_gather(dist, world, rank, d_ang[0], n_buffers=n_buf, min_gbps=8.0)
                          if use_dist else (None, None))

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    for row in ev:          # torch creates the underlying hipEvent_t lazily, on the first record()
        for e in row:
            e.record(main_stream)

    def step(i, events=None):
        b = i % n_buf
        buf = d_ang[b]
        stream = streams[i % len(streams)]
        with torch.cuda.stream(stream):
            if gather:
                gather.wait_buffer(b)  # the gather that last read this buffer has completed
            # ONE C-ABI call = the four stage kernels; the library records the given HIP events between them
            _lib.solve_seq_device(d_pose.data_ptr(), S, L, T, params, buf.data_ptr(),
                                  d_fks[i % len(streams)].data_ptr(), stream=stream.cuda_stream,
                                  block_size=args.block, layout=layout, lanes_per_wave=args.lanes_per_wave, staged=int(args.staged), interleave_legs=args.interleave_legs,
                                  stage_events=[e.cuda_event for e in events] if events else None)
            if gather:
                gather.submit(b, buf)

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if gather:
        gather.drain()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i])
    if gather:
        gather.drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # Outside the timed region: every buffer the overlapped launches wrote must hold, bit for bit, what one launch
    # made alone writes (all steps solve the same batch) -- a measurement of launches that disturbed each other
    # would be worthless.
    chk_ang, chk_fk = torch.zeros_like(d_ang[0]), torch.zeros_like(d_fk)
    _lib.solve_seq_device(d_pose.data_ptr(), S, L, T, params, chk_ang.data_ptr(), chk_fk.data_ptr(),
                          stream=main_stream.cuda_stream, block_size=args.block, layout=layout,
                          lanes_per_wave=args.lanes_per_wave, staged=int(args.staged), interleave_legs=args.interleave_legs)
    torch.cuda.synchronize()
    used = range(min(n_buf, args.steps + args.warmup))
    if not all(torch.equal(d_ang[b], chk_ang) for b in used) or \
            not all(torch.equal(f, chk_fk) for f in d_fks[:min(len(d_fks), args.steps + args.warmup)]):
        raise SystemExit("bench: overlapped launches did not reproduce a launch made alone -- result invalid")
    del chk_ang, chk_fk

    # per-kernel durations from the HIP events recorded on the launch stream inside the timed region
    stage_ms = np.array([[ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(4)] for i in range(args.steps)])
    mean_stage_ms = stage_ms.mean(0)
    if args.staged:
        dom = int(np.argmax(mean_stage_ms)) + 1
        kname, key, bytes_unit, dom_ms = f"seqik_stage_kernel<{dom}, ...>", f"stage{dom}", BYTES_STAGE[dom], float(mean_stage_ms[dom - 1])
    else:  # one kernel per step: event [0] is recorded in front of it, [1] behind it
        kname, key, bytes_unit, dom_ms = "seqik_fused_kernel<true>", "fused", BYTES_PATH, float(mean_stage_ms[0])
    ach = bytes_unit * units_per_step / (dom_ms * 1e-3) / 1e9
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_r01_staged.json" if args.staged else "traffic_r01.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("units_per_launch") == units_per_step and tj.get("variant") == args.variant:
            traffic = tj.get(f"{key}_hbm_bytes_per_launch")
            names = [f"stage{k}" for k in (1, 2, 3, 4)] if args.staged else ["fused"]
            insts = [tj.get(f"{n}_valu_insts_per_launch") for n in names]
            if all(v is not None for v in insts):
                # What actually bounds the path: VALU issue.  A wave64 VALU instruction occupies its SIMD's 16
                # f64 lanes for >= 4 cycles (f64 FMA/MUL/ADD: exactly 4; rcp/rsq seeds: more), so the step cannot
                # be shorter than  instructions x 4 / (SIMDs x clock).
                n_cu, clock_khz, _ = _lib.device_attributes(device_index)
                simds, clock_hz = n_cu * 4, clock_khz * 1e3
                floor_ms = sum(insts) * 4.0 / (simds * clock_hz) * 1e3
                valu = {"valu_insts_per_step": sum(insts), "simds": simds, "clock_MHz": clock_khz / 1e3,
                        "issue_floor_ms_per_step": floor_ms, "measured_ms_per_step": elapsed / args.steps * 1e3,
                        "frac_of_valu_issue_peak": floor_ms / (elapsed / args.steps * 1e3),
                        "lane_utilisation": [tj.get(f"{n}_valu_lane_utilisation") for n in names],
                        "source": "SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU per launch from "
                                  f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc), timing live"}
                mix = {c: [tj.get(f"{n}_f64_{c}_insts_per_launch") for n in names] for c in ("add", "mul", "fma", "trans")}
                utils_ = valu["lane_utilisation"]
                if all(v is not None for vs in mix.values() for v in vs) and all(u is not None for u in utils_):
                    # FP64 operations actually performed: wave-level instruction counts by class x 64 lanes x the
                    # share of active lanes (FMA = 2 flops), against the 78.6 TFLOP/s FP64 vector peak
                    flops = sum((mix["add"][i] + mix["mul"][i] + mix["trans"][i] + 2.0 * mix["fma"][i]) * 64.0 * utils_[i]
                                for i in range(len(names)))
                    tfl = flops / (elapsed / args.steps) / 1e12
                    peak = simds * 16 * 2 * clock_hz / 1e12
                    valu["fp64"] = {"f64_insts_per_step": {c: sum(vs) for c, vs in mix.items()},
                                    "thread_level_TFLOPs": tfl, "vector_peak_TFLOPs": peak, "frac": tfl / peak,
                                    "note": "lane share taken from all VALU instructions (SQ_THREAD_CYCLES_VALU)"}
    roofline = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_unit": bytes_unit, "avg_launch_ms": dom_ms,
                "valu": valu,
                "note": "FP64-VALU-issue-bound solver: ~1e4 f64 instructions per 392 B; HBM fraction << 1% by "
                        "construction (SURVEY 8d); `valu` is the roofline that binds"}
    if args.staged:
        roofline["stage_ms"] = [float(v) for v in mean_stage_ms]
        roofline["path_GBps"] = BYTES_PATH * units_per_step / (mean_stage_ms.sum() * 1e-3) / 1e9

    if rank == 0:
        total_units = units_per_step * world * args.steps
        out = {
            "metric": "leg-IK solves/s (frames x 6 legs)",
            "value": total_units / elapsed,
            "unit": "leg-frame solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 3: synthetic 1M frames x 6 legs per GPU, in-workspace targets",
                       "variant": args.variant, "frames_per_gpu": S * T, "legs": L, "sequences_per_gpu": S,
                       "frames_per_sequence": T, "chains_per_gpu": S * L, "warm_start": "previous frame",
                       "outputs": "7 angles + 9x3 FK per leg-frame", "device_layout": "planar",
                       "streams": len(streams), "launches_per_step": 4 if args.staged else 1,
                       "parallelism": f"sequence-sharded x{world}, angle gather to rank 0" if world > 1 else "1 GPU",
                       **({"gather": gather_how} if gather_how else {})},
            "roofline": roofline,
            "verified": "after timing: every angle / FK buffer written by the overlapped launches == one launch made "
                        "alone, bit for bit (smoke() and tests/ compare that launch with the oracle)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pose, legs, body, args.cpu_sample_seqs, not args.no_python_baseline)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
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
