#!/usr/bin/env python3
"""Benchmark of the sequential leg-IK hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--detail] [--legs all]

Metric (BASELINE.json): leg-IK solves/s, one solve = one (frame, leg) = 4 stage sub-solves -> 7 joint angles + the stage-4
forward kinematics.  Workload: BASELINE config 3, "synthetic 1M frames x 6 legs, random in-workspace target key points,
1 -> 8 MI355X frame-sharded": ONE fixed problem of 1,000,000 frames cut into 15,625 independent sequences of 64 frames (frame t
of a sequence is warm-started from frame t-1: the reference's semantics, examples/example_leg_inv_kinematics_parallel.py:143-198
applied to many recordings), 6 legs each = 93,750 chains; rank r solves sequences [r S/N, (r+1) S/N) (`--scaling strong`, the
default; `weak` = 1M frames per GPU).  A step = one pass of the hot path over the rank's batch, inputs resident in HBM;
consecutive steps overlap on HIP streams (`--streams 0`: the depth is calibrated over the very region that is measured).

OUTPUT.  The LAST line of stdout is ONE compact JSON object (< 4 KB: `compact_line`) -- the contract's keys, `roofline`,
`cpu_baseline`, `parity_max_abs_dtheta`, `value_single_job`, `value_smooth` and, at N > 1, `multi_gpu`.  The full record (every
calibration candidate, the VALU issue floor, per-rank times, `--detail` legs) goes to `bench_detail.json` and to stderr.

N > 1: started by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one rank per GPU) or plainly as
`python bench.py --gpus N`, in which case this process only LAUNCHES that command as a child, relays rank 0's line and exits
with the child's code.  The default N > 1 run is: provisional headline (3 steps in flight, RCCL point-to-point gather) ->
depth calibration -> headline (every step gathers the rank's joint angles on rank 0: grouped RCCL point-to-point over xGMI, the
north star's gather) -> bit check against a lone launch -> `n1_reference` (the WHOLE problem on rank 0's GPU alone, same run) ->
`ranks_seen`.  `--legs all` adds the legs of bench_extras.py (both gathers, the other scaling mode, the frame-sharded recording,
config 5 over the ranks); `--detail` (N = 1) adds the other BASELINE configs, the full parity report and the share projection.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "sequential-inverse-kinematics_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def launch_ranks_if_needed(argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process becomes the LAUNCHER.  It starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child (one rank per GPU; the shape of
    the reference's own parallel script, which builds its pool and merges the results itself:
    examples/example_leg_inv_kinematics_parallel.py:163-198), relays rank 0's JSON line and exits with the child's code --
    non-zero when any rank failed (75: only the provisional headline came out), 124 on time-out, 3 when no JSON line came
    back.  It runs BEFORE torch is imported and never touches the GPU (a process that has initialised the GPU must not be
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
        if '"provisional":true' in lines[-1] and rc in (0, 1):
            rc = 75   # (torch.distributed.run reports any failed rank as 1: the line says which failure it was)
    elif rc == 0:
        rc = 3
    sys.exit(rc)


def set_runtime_env():
    """The environment this measurement runs in, set BEFORE the HIP runtime starts and recorded in `config.env`:
    `seqikpy_amd.recommended_env(steps_in_flight=20)` -- 22 hardware queues, because the per-rank shares of the fixed problem only
    fill the GPU with up to 20 steps in flight, each stream on a queue of its own -- unless the caller has set
    the variables already.  The package itself never edits os.environ."""
    import seqikpy_amd
    for k, v in seqikpy_amd.recommended_env(steps_in_flight=20).items():
        os.environ.setdefault(k, v)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])
    set_runtime_env()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_support as bs  # noqa: E402
from bench_support import BYTES_PATH, BYTES_STAGE, FP64_VECTOR_PEAK_TF, HBM_PEAK_GBS, Batch  # noqa: E402
from seqikpy_amd import _lib, runtime_env, sharding, synthetic  # noqa: E402

LINE_LIMIT = 4096        # bytes of the compact line (the driver's parser lost a 26 KB line in round 5)
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")


def parse(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="100 steps = 1.2 s; with 20 the fill and drain of the pipeline still cost 5 %%")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): --frames in total, split over the ranks (BASELINE config 3: ONE fixed problem); weak: --frames per GPU")
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames (x 6 legs) per GPU (weak) or in total (strong)")
    ap.add_argument("--frames-per-seq", type=int, default=64)
    ap.add_argument("--variant", default="iid", choices=["iid", "smooth"])
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--lanes-per-wave", type=int, default=0, help="chains per wavefront (0 = automatic)")
    ap.add_argument("--interleave-legs", type=int, default=0, help="1 = consecutive chains per wave instead of leg-pure waves")
    ap.add_argument("--stage-pipeline", type=int, default=-1,
                    help="SeqikOptions.reserved[3]: 0 = the library's choice, 1 = lane-per-chain kernels, 2 = stage pipeline; "
                         "-1 (default) = 0 with an explicit --streams, calibrated together with the depth otherwise")
    ap.add_argument("--staged", action="store_true", help="one launch per stage (SeqikOptions.reserved[1] = 1) instead of the single launch")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams the steps are issued on round-robin; 0 (default) = calibrated in the run")
    ap.add_argument("--detail", action="store_true", help="N = 1: also run the legs of bench_extras.py (other configs, parity report, shares)")
    ap.add_argument("--legs", default="default", choices=["default", "all"], help="N > 1: `all` adds the legs of bench_extras.py")
    ap.add_argument("--one-recording", action="store_true", help="config 3 as ONE recording, frame-sharded over the ranks (bench_extras.py)")
    ap.add_argument("--detail-path", default=os.path.join(ROOT, "bench_detail.json"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-python-baseline", action="store_true", help="skip the Python + scipy process-pool leg of the CPU baseline (about 20 s)")
    ap.add_argument("--no-extras", action="store_true", help="headline and roofline only (profiling runs)")
    ap.add_argument("--cpu-sample-seqs", type=int, default=8192,
                    help="sequences of the batch the CPU baseline solves (8192 x 6 x 64 = 3.1 M leg-frames: 10-20 s on 16 cores)")
    return ap.parse_args(argv)


def compact_line(rec):
    """The ONE line of stdout: the contract's keys + roofline + cpu_baseline + a handful of scalars, as a string < LINE_LIMIT
    bytes.  Everything else of `rec` stays in bench_detail.json."""
    roof, cfg, cpu, multi = rec.get("roofline") or {}, rec.get("config") or {}, rec.get("cpu_baseline") or {}, rec.get("multi_gpu")
    line = {k: rec.get(k) for k in HEAD_KEYS}
    line["config"] = {k: cfg[k] for k in ("workload", "variant", "streams", "stage_pipeline", "chain_queue", "provisional", "gather", "env",
                                             "frames_per_rank", "boundary_rounds") if k in cfg}
    valu = roof.get("valu_issue") or {}
    line["roofline"] = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")}
    hbm = roof.get("hbm") or roof          # (bound "hbm": the HBM figures ARE the top level)
    line["roofline"].update({"hbm_frac": hbm.get("frac"), "bytes_per_unit": hbm.get("bytes_per_unit"),
                             "traffic_over_algorithmic": roof.get("traffic_over_algorithmic"),
                             "valu_issue_floor_frac": (valu.get("measured_costs") or {}).get("frac_of_valu_issue_floor"),
                             "lane_utilisation": (valu.get("lane_utilisation") or [None])[0],
                             "pmc_file": roof.get("pmc_file"), "pmc_matches_build": roof.get("pmc_matches_build")})
    if cpu:
        line["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cpu}
        line["cpu_baseline"]["python_scipy_pool_value"] = (cpu.get("python_scipy_pool") or {}).get("value")
    for k in ("parity_max_abs_dtheta", "parity_tolerance", "value_single_job", "value_smooth", "value_iid", "gpu_over_cpu", "verified", "extras_error",
              "check", "detail", "detail_scalars"):
        if rec.get(k) is not None:
            line[k] = rec[k]
    if multi:
        n1 = multi.get("n1_reference") or {}
        line["multi_gpu"] = {"backend": multi.get("backend"), "ranks_seen_n": len(multi.get("ranks_seen") or []),
                             "rccl_ranks": multi.get("rccl_ranks"), "devices_distinct": multi.get("devices_distinct"),
                             "efficiency_vs_n1": multi.get("efficiency_vs_n1"), "speedup_vs_n1": multi.get("speedup_vs_n1"),
                             "n1_value": n1.get("value"), "n1_ms_per_step": n1.get("ms_per_step"), "gather": cfg.get("gather"),
                             "rank_ms_per_step_min_max": [(multi.get("rank_ms_per_step") or {}).get(k) for k in ("min", "max")],
                             "legs": sorted(k for k in multi if k in ("gather_compare", "weak", "strong", "one_recording", "config5")),
                             **({"timed_out": multi["timed_out"]} if multi.get("timed_out") else {})}
    s = json.dumps(bs.round_numbers(line), separators=(",", ":"))
    for drop in ("detail_scalars", "verified"):          # never reached with today's fields; the limit holds whatever is added later
        if len(s) >= LINE_LIMIT and drop in line:
            del line[drop]
            s = json.dumps(bs.round_numbers(line), separators=(",", ":"))
    if len(s) >= LINE_LIMIT:
        raise RuntimeError(f"bench.py: compact line of {len(s)} bytes")
    return s


def build_roofline(args, batch, ev, ms_per_step, device_index):
    """`roofline` of the dominant kernel: its average launch duration from the HIP events the library recorded around it on its
    launch stream inside the timed region; algorithmic bytes x units per launch / that against HBM; and -- what actually binds --
    the FP64 VALU figures from the committed PMC summary of THIS build (bench_support.pmc_roofline)."""
    main_steps = [i for i in range(args.steps) if not bs.in_lat_range(batch, i)]
    stage_ms = np.array([[ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(4)] for i in main_steps]).mean(0)
    if args.staged:
        dom = int(np.argmax(stage_ms)) + 1
        kname, key, bytes_unit, dom_ms = f"seqik_stage_kernel<{dom}, ...>", f"stage{dom}", BYTES_STAGE[dom], float(stage_ms[dom - 1])
    else:  # one kernel per step: event [0] is recorded in front of it, [1] behind it
        queued = batch.pool > 0
        kname, key, bytes_unit, dom_ms = ("seqik_fused_queue_kernel<true>" if queued else "seqik_fused_kernel<true>"), "fused", BYTES_PATH, float(stage_ms[0])
    ach_gbs = bytes_unit * batch.units / (dom_ms * 1e-3) / 1e9
    traffic, valu, fp64, matches, pmc_file = bs.pmc_roofline(args.variant, args.staged, key, batch.units, ms_per_step, device_index, batch.pool)
    hbm = {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS, "bytes_per_unit": bytes_unit,
           "note": f"algorithmic bytes x units per launch / the kernel's average launch duration (launches of {len(batch.streams)} steps "
                   f"overlap, so a launch lasts ~{len(batch.streams)}x a step); HBM is not what binds: ~1e4 f64 instructions per 392 B"}
    if fp64:
        tfl = float(f"{fp64['flops_per_step'] / (ms_per_step * 1e-3) / 1e12:.3g}")
        roof = {"bound": "valu-fp64", "kernel": kname, "achieved": tfl, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                "frac": float(f"{tfl / FP64_VECTOR_PEAK_TF:.3g}"), "traffic": traffic, "avg_launch_ms": dom_ms, "fp64": fp64,
                "valu_issue": valu, "hbm": hbm,
                "note": "bound by FP64 VALU issue, not HBM or MFMA: `achieved` = FP64 operations performed by active lanes per second "
                        "(PMC instruction mix x lane share, live timing) against the vector FP64 peak; `valu_issue` = how close the step "
                        "is to the floor its wave-instruction count allows; `hbm` = the algorithmic-bytes figure"}
    else:  # no PMC summary of THIS build for this workload: only the HBM figure can be stated
        roof = {"bound": "hbm", "kernel": kname, **hbm, "traffic": traffic, "avg_launch_ms": dom_ms}
    roof.update({"pmc_matches_build": matches, "pmc_file": pmc_file,
                 "traffic_over_algorithmic": traffic / (bytes_unit * batch.units) if traffic else None})
    if args.staged:
        roof["stage_ms"] = [float(v) for v in stage_ms]
    return roof


def main():
    args = parse()
    sys.stdout.flush()
    json_fd = os.dup(1)          # stdout carries the ONE line and nothing else (RCCL prints a banner to fd 1): the rest goes to stderr
    os.dup2(2, 1)
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (the launcher starts one rank per GPU)")
    n_dev = torch.cuda.device_count()
    device_index = local_rank % max(n_dev, 1)  # (more ranks than GPUs only happens in the one-GPU rehearsal)
    torch.cuda.set_device(device_index)
    dist, backend = None, None
    use_dist = world > 1 or os.environ.get("SEQIK_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: the process-group path with one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" = RCCL over xGMI, one GPU per rank; ranks that share a GPU (rehearsal on a one-GPU box) talk over gloo
        backend = os.environ.get("SEQIK_BENCH_BACKEND") or ("nccl" if n_dev >= world else "gloo")
        if backend == "nccl":   # the solver keeps every CU busy: RCCL's stream gets priority so the gather is dispatched as slots free up
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index),
                                    pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ranks = bs.Ranks(dist, world, rank, backend)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    if args.one_recording:
        import bench_extras
        rec = bench_extras.one_recording_main(args, ranks)
        if rank == 0:
            sys.stderr.write(json.dumps(rec) + "\n")
            emit(compact_line(rec))
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    T, L = args.frames_per_seq, 6
    S_total = args.frames // T
    lo, hi, S_job = sharding.rank_share(S_total, world, rank, args.scaling)
    seed = synthetic.SEED_BASE + (1000 * rank if args.scaling == "weak" else 0)
    legs, body, pose_all, params = bs.make_workload(S_total, T, args.variant, seed)   # strong: generated alike on every rank
    pose = pose_all[lo:hi] if args.scaling == "strong" else pose_all
    whole_pose = pose_all if (rank == 0 and world > 1) else None      # rank 0 keeps the fixed problem: the one-GPU reference
    del pose_all
    S, units_all = pose.shape[0], S_job * L * T
    s_pad = max(b - a for a, b, _ in (sharding.rank_share(S_total, world, r, args.scaling) for r in range(world)))  # equal gather blocks
    explicit = args.streams > 0
    first = (args.streams, max(0, args.stage_pipeline)) if explicit else bs.DEPTH_CANDIDATES[0][:2]
    batch = Batch(pose, params, args, first[0], pipeline=first[1], s_pad=s_pad)
    gather_kind = (os.environ.get("SEQIK_GATHER") or "rccl") if use_dist else None
    frames_txt = "1M" if S_total * T == 1_000_000 else f"{S_total * T:,}".replace(",", " ")

    def buffers_for(bt):   # one angle buffer per step in flight + two spare, so a draining gather does not hold a launch back
        return [bt.angle_buffer() for _ in range(max(2, len(bt.streams) + (2 if use_dist else 0)))]

    def make_gather(like, n_buffers):
        """The final joint-angle gather of a step: grouped RCCL point-to-point over xGMI (the north star's gather) unless
        SEQIK_GATHER=peer asks for the copy-engine peer writes (seqikpy_amd/peer_gather.py; `--legs all` times both)."""
        if not use_dist:
            return None, None
        if gather_kind == "peer":
            from seqikpy_amd import peer_gather
            return peer_gather.make_gather(dist, world, rank, like, n_buffers=n_buffers, min_gbps=8.0, prefer="peer")
        return sharding.GatherPipeline(dist, world, rank, like, dst=0, n_buffers=n_buffers), "grouped RCCL point-to-point"

    def record(ms_per_step, elapsed, bt, roofline, extra_cfg=None):
        """The record's headline part (the contract's keys + config) for a measurement of `elapsed` seconds over args.steps steps."""
        workload = (f"config 3: synthetic {frames_txt} frames x 6 legs, in-workspace targets" if world == 1 else
                    f"config 3 (weak-scaling variant): synthetic {frames_txt} frames x 6 legs PER GPU" if args.scaling == "weak" else
                    f"config 3 literally: the FIXED problem of synthetic {frames_txt} frames x 6 legs IN TOTAL ({S_total} sequences of {T} "
                    f"frames), sequences split over the {world} ranks, joint angles gathered on rank 0")
        return {"metric": "leg-IK solves/s (frames x 6 legs); max |d theta| vs reference in parity_max_abs_dtheta",
                "value": units_all * args.steps / elapsed, "unit": "leg-frame solves/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": workload, "variant": args.variant, "streams": len(bt.streams), "stage_pipeline": bt.pipeline,
                           "chain_queue": bt.pool,
                           "frames_total": S_total * T * (world if args.scaling == "weak" else 1), "leg_frames_per_step_all_ranks": int(units_all),
                           "frames_per_gpu": S * T, "legs": L, "sequences_per_gpu": S, "frames_per_sequence": T, "chains_per_gpu": S * L,
                           "warm_start": "previous frame", "outputs": "7 angles + 9x3 FK per leg-frame", "device_layout": "planar",
                           "launches_per_step": 4 if args.staged else 1, "env": runtime_env(),
                           "parallelism": f"sequence-sharded x{world}, angle gather to rank 0" if world > 1 else "1 GPU", **(extra_cfg or {})},
                "roofline": roofline}

    lifeline = bs.Lifeline(rank, json_fd) if (use_dist and world > 1) else None
    if lifeline is not None:
        # A PROVISIONAL headline first, on the plainest path there is: should a calibration below ever get stuck, this line comes
        # out (marked provisional, exit code 75 on every rank) instead of nothing.
        bufs0 = buffers_for(batch)
        g0, how0 = make_gather(bufs0[0], len(bufs0))
        tm0, _ = ranks.timed_region(batch, bufs0, g0, args.steps, args.warmup)
        ach0 = BYTES_PATH * batch.units / (tm0 / args.steps) / 1e9
        prov = record(tm0 / args.steps * 1e3, tm0, batch,
                      {"bound": "hbm", "kernel": None, "achieved": ach0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach0 / HBM_PEAK_GBS,
                       "traffic": None, "avg_launch_ms": None, "bytes_per_unit": BYTES_PATH,
                       "note": "provisional: algorithmic bytes of this rank's share / wall time per step (no kernel events)"},
                      {"provisional": True, "gather": how0})
        if hasattr(g0, "close"):
            g0.close()
        del bufs0, g0
        lifeline.arm(float(os.environ.get("SEQIK_BENCH_STAGE_TIMEOUT", "300")), lambda: compact_line(prov), "the calibrations / the headline",
                     bs.Lifeline.EXIT_PROVISIONAL)

    depth_calibration = None
    if not explicit:
        # How many steps to keep in flight, and on which kernel family: measured over exactly the region that will be measured
        # (warm-up + steps between two synchronisations), with the gather running; every rank takes the candidate that is
        # fastest on the SLOWEST rank.  The whole problem fills the GPU at depth 3 (the only candidate then); a 1/8 share needs 20.
        depth_calibration = {"candidates": [], "rule": "fastest ms per step on the slowest rank over the same region as the headline"}
        best = None
        plain = args.lanes_per_wave == 0 and not args.staged and not args.interleave_legs
        for n_st, pipe, lat, pool in [c + (0,) for c in bs.depth_candidates(args.steps, S * L)] + \
                [(d, 1 if p else 0, None, p) for d, p in (bs.queue_candidates(S * L) if plain else [])]:
            bt, ms = None, float("inf")
            try:
                bt = Batch(None, params, args, n_st, pipeline=pipe, like=batch, pool=pool)
                bt.lat_range = lat
                bufs = buffers_for(bt)
                local_ok = True
            except Exception as exc:  # noqa: BLE001  (e.g. no memory for 20 FK buffers)
                sys.stderr.write(f"bench.py rank {rank}: depth candidate {(n_st, pipe)} failed ({type(exc).__name__}: {exc})\n")
                local_ok, bufs = False, None
            if ranks.all_ok(local_ok):       # consensus BEFORE any collective of the candidate: nobody waits for a rank that failed
                g, _ = make_gather(bufs[0], len(bufs))
                tm, _ = ranks.timed_region(bt, bufs, g, args.steps, args.warmup)
                if hasattr(g, "close"):
                    g.close()
                ms = tm / args.steps * 1e3
                if best is None or ms < best[0]:
                    best = (ms, n_st, pipe, bt, lat, pool)
            depth_calibration["candidates"].append({"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat, "chain_queue": pool,
                                                    "ms_per_step": None if ms == float("inf") else ms})
            del bufs
        if best is None:
            raise SystemExit("bench: no pipeline depth could be run")
        depth_calibration["chosen"] = {"streams": best[1], "stage_pipeline": best[2], "latency_kernel_steps": best[4], "chain_queue": best[5]}
        batch = best[3]
        del batch.d_fks[len(batch.streams):]
        torch.cuda.empty_cache()
    d_ang = buffers_for(batch)
    gather, gather_how = make_gather(d_ang[0], len(d_ang))

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    for row in ev:          # torch creates the underlying hipEvent_t lazily, on the first record()
        for e in row:
            e.record(batch.main)
    elapsed, elapsed_mine = ranks.timed_region(batch, d_ang, gather, args.steps, args.warmup, ev)

    # Outside the timed region: every buffer the overlapped launches wrote must hold, bit for bit, what one launch made alone
    # writes (all steps solve the same batch) -- a measurement of launches that disturbed each other would be worthless.
    chk_ang, chk_fk = torch.zeros_like(d_ang[0]), torch.zeros_like(batch.d_fks[0])
    _lib.solve_seq_device(batch.d_pose.data_ptr(), S, L, T, params, chk_ang.data_ptr(), chk_fk.data_ptr(),
                          stream=batch.main.cuda_stream, block_size=args.block, layout=batch.layout,
                          lanes_per_wave=min(args.lanes_per_wave, 64),    # (a chain-queue run is checked against the PLAIN launch)
                          staged=int(args.staged), interleave_legs=args.interleave_legs)
    torch.cuda.synchronize()
    n_used = max(args.steps, args.warmup)  # warm-up and timed steps both count from 0
    same = all(torch.equal(d_ang[b], chk_ang) for b in range(min(len(d_ang), n_used))) and \
        all(torch.equal(f, chk_fk) for f in batch.d_fks[:min(len(batch.d_fks), n_used)])
    _lib.check_faults()
    if not ranks.all_ok(same):
        raise SystemExit("bench: overlapped launches did not reproduce a launch made alone -- result invalid")
    del chk_ang, chk_fk

    ms_per_step = elapsed / args.steps * 1e3
    rec = record(ms_per_step, elapsed, batch, build_roofline(args, batch, ev, ms_per_step, device_index),
                 {**({"depth_calibration": depth_calibration} if depth_calibration else {}), **({"gather": gather_how} if gather_how else {})})
    rec["verified"] = "every angle / FK buffer of the timed region == one launch made alone, bit for bit"
    ctx = dict(args=args, ranks=ranks, batch=batch, d_ang=d_ang, params=params, pose=pose, legs=legs, body=body, rec=rec, S_total=S_total,
               T=T, L=L, units_all=units_all, elapsed=elapsed, ms_per_step=ms_per_step, device_index=device_index, gather=gather,
               whole_pose=whole_pose, buffers_for=buffers_for, gather_kind=gather_kind)

    if dist and world > 1:
        import socket
        seen = [None] * world
        props = torch.cuda.get_device_properties(device_index)
        dist.all_gather_object(seen, {"rank": rank, "host": socket.gethostname(), "device": device_index,
                                      "gpu": str(getattr(props, "uuid", "")) or getattr(props, "pci_bus_id", None),
                                      "ms_per_step": elapsed_mine / args.steps * 1e3})
        ones = torch.ones(1, dtype=torch.float64, device=ranks.coll_dev)
        dist.all_reduce(ones)       # how many ranks the communicator itself (RCCL on a real job) sums over
        multi = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (ranks share a GPU: rehearsal)"),
                 "ranks_seen": [{k: r[k] for k in ("rank", "host", "device", "gpu")} for r in seen], "rccl_ranks": int(ones.item()),
                 "devices_distinct": len({(r["host"], r["gpu"] or r["device"]) for r in seen}),
                 "rank_ms_per_step": {"min": min(r["ms_per_step"] for r in seen), "max": max(r["ms_per_step"] for r in seen),
                                      "by_rank": [r["ms_per_step"] for r in seen]}}
        rec["multi_gpu"] = multi
        if not args.no_extras:
            # The headline is measured AND verified: from here on a leg that does not finish costs only itself (exit code 0)
            legs_deadline = float(os.environ.get("SEQIK_BENCH_LEGS_TIMEOUT", "300"))

            def line_so_far():
                multi["timed_out"] = f"a leg behind the headline did not finish within {legs_deadline:.0f} s"
                return compact_line(rec)
            lifeline.arm(legs_deadline, line_so_far, "a leg behind the headline", 0)
            # n1_reference: the WHOLE fixed problem on rank 0's GPU alone, same run, same pipeline, no gather (one GPU has nobody
            # to gather from), while the other ranks wait in the all-gather: what the scaling efficiency of THIS run is against
            n1 = None
            if rank == 0:
                try:
                    b1 = Batch(whole_pose, params, args, 3, pipeline=0)
                    bufs1 = [b1.angle_buffer() for _ in range(3)]
                    dt1 = bs.timed_steps(b1, bufs1, args.steps, 3, warmup=args.warmup)
                    n1 = {"value": b1.units * args.steps / dt1, "unit": "leg-frame solves/s", "ms_per_step": dt1 / args.steps * 1e3,
                          "steps": args.steps, "streams": 3, "leg_frames_per_step": int(b1.units),
                          "what": "the whole fixed problem on rank 0's GPU alone (the other ranks idle), same run, 3 steps in flight"}
                    del b1, bufs1
                    torch.cuda.empty_cache()
                except Exception as exc:  # noqa: BLE001
                    n1 = {"error": f"{type(exc).__name__}: {exc}"}
            got = [None] * world
            dist.all_gather_object(got, n1)
            multi["n1_reference"] = got[0]
            if got[0] and "value" in got[0]:
                multi["speedup_vs_n1"] = rec["value"] / got[0]["value"]
                multi["efficiency_vs_n1"] = rec["value"] / (world * got[0]["value"])
            if args.legs == "all":
                import bench_extras
                bench_extras.multi_gpu_legs(ctx, multi)
        lifeline.disarm()

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            lone = Batch(None, params, args, 1, pipeline=0, like=batch)       # ONE call at a time: the library's own kernel choice
            k1 = max(4, min(16, args.steps // 6))
            dt1 = bs.timed_steps(lone, d_ang, k1, 1)
            rec["single_job"] = {"value": batch.units * k1 / dt1, "unit": "leg-frame solves/s", "ms_per_step": dt1 / k1 * 1e3, "steps": k1,
                                 "note": "one batch at a time: 1 465 full waves on 1 024 SIMDs; `value` is the pipelined rate"}
            rec["value_single_job"] = rec["single_job"]["value"]
            other = "smooth" if args.variant == "iid" else "iid"
            _, _, pose_o, _ = bs.make_workload(S, T, other, synthetic.SEED_BASE)
            bo = Batch(pose_o, params, args, len(batch.streams), pipeline=batch.pipeline, pool=batch.pool)
            dto = bs.timed_steps(bo, d_ang, args.steps, len(bo.streams), warmup=args.warmup)
            ms_o = dto / args.steps * 1e3
            _, valu_o, fp64_o, match_o, file_o = bs.pmc_roofline(other, args.staged, "fused", bo.units, ms_o, device_index, pool=batch.pool)
            rec["variants"] = {other: {"value": bo.units * args.steps / dto, "unit": "leg-frame solves/s", "ms_per_step": ms_o, "steps": args.steps,
                                       "roofline": {"pmc_file": file_o, "pmc_matches_build": match_o,
                                                    **({"bound": "valu-fp64", "frac": fp64_o["flops_per_step"] / (ms_o * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF,
                                                        "lane_utilisation": valu_o["lane_utilisation"]} if fp64_o else {})}},
                               "note": "smooth = temporally continuous targets (the realistic case); iid = every frame an unrelated pose"}
            rec["value_" + other] = rec["variants"][other]["value"]
            del bo, pose_o
            rec["parity_max_abs_dtheta"], rec["parity_tolerance"] = bs.quick_parity(), 1e-4
            if args.detail:
                import bench_extras
                bench_extras.detail_legs(ctx)
            _lib.check_faults()
        except Exception as exc:  # noqa: BLE001  (the headline above is measured and must be printed whatever happens here)
            import traceback
            rec["extras_error"] = f"{type(exc).__name__}: {exc}"
            sys.stderr.write("bench.py: a leg behind the headline failed, the line is printed without it:\n" + traceback.format_exc())
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = bs.cpu_baseline(pose, legs, body, args.cpu_sample_seqs, not args.no_python_baseline)
            rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
        except Exception as exc:  # noqa: BLE001
            rec["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        rec["detail"] = os.path.basename(args.detail_path)
        full = json.dumps(rec)
        try:
            with open(args.detail_path, "w") as fh:
                fh.write(full + "\n")
        except OSError as exc:
            rec["detail"] = f"not written ({exc})"
        sys.stderr.write(full + "\n")
        emit(compact_line(rec))
    if dist:
        if hasattr(gather, "close"):
            gather.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
