"""Machinery of `bench.py` (the flow of the measurement is there, the parts it is made of are here): the synthetic workload,
one rank's batch resident in HBM and the launch of a step, the pipeline depths a run calibrates among, the timed region, the
PMC side of the roofline (committed counter summaries under profiles/, used only for the build they were measured on), the CPU
baseline and the deadline thread of an N > 1 run.  Imports torch: `bench.py` imports this module only after its launcher
check (a launcher process never touches the GPU)."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "sequential-inverse-kinematics_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (loads the HIP runtime that libseqik_hip.so binds to)

from seqikpy_amd import _lib, data, sharding, synthetic, utils  # noqa: E402

MAX_DEPTH = int(os.environ.get("SEQIK_BENCH_MAX_DEPTH") or 20)   # steps in flight at most (24 streams of a 1/8 share lose again)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TF = 78.6  # 256 CUs x 4 SIMDs x 16 f64 lanes x 2 flop x 2.4 GHz
# Algorithmic HBM bytes per leg-frame (DESIGN.md "Kernels"):
BYTES_PATH = 120 + 56 + 216   # key points in, 7 angles out, 9x3 FK out
BYTES_STAGE = {1: 48 + 16 + 96, 2: 48 + 96 + 16 + 96 + 48, 3: 48 + 96 + 16 + 96 + 24, 4: 48 + 96 + 8 + 144}
# stage k reads the origin + its key point (48 B) and the 96-byte prefix frame the previous stage left in
# the workspace, writes its own angles, the next prefix frame (96 B) and its rows of the 9 x 3 FK record
TRAFFIC_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")  # newest first; a summary is used only if it matches the workload AND the build
LATENCY_ROUND = "r06"     # profiles/<round>_latency_floor.json (scripts/latency_floor.py)
LF_WINDOW = (284, 302)   # tests/conftest.py::LF_DEGENERATE: the anipose LF kinematic-singularity episode


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
    workspace per (device, stream), at most 64 of them, and the hardware queues are fewer (GPU_MAX_HW_QUEUES)."""
    if not _STREAMS:
        _STREAMS.append(torch.cuda.current_stream())
    while len(_STREAMS) < n:
        _STREAMS.append(torch.cuda.Stream())
    return list(_STREAMS[:n])


class Batch:
    """One rank's batch resident in HBM (planar layout) + the launch of one step on a given stream."""

    def __init__(self, pose, params, args, n_streams, pipeline=None, like=None, s_pad=None, pool=0):
        """`like`: another Batch of the SAME key points (its device copy and streams are shared, only FK buffers are added).
        `s_pad`: sequences the ANGLE buffers are allocated for (>= this rank's own): the shares of the fixed problem differ by
        one sequence between ranks (15 625 = 8 x 1 953 + 1), and the gather moves equal blocks from every rank."""
        self.params, self.args = params, args
        self.s_pad = s_pad if s_pad is not None else (like.s_pad if like is not None else None)
        self.pipeline = max(0, getattr(args, "stage_pipeline", 0)) if pipeline is None else pipeline
        self.streams = stream_pool(n_streams)
        self.main = self.streams[0]
        self.lat_range = None     # [lo, hi) steps of a timed region launched with the library's own kernel choice (depth_candidates)
        # chain queue (SeqikOptions.reserved[0] = 128, 256, ...: chains a wavefront owns); 0 = --lanes-per-wave as given
        self.pool = int(pool) or (a_pool if (a_pool := getattr(args, "lanes_per_wave", 0)) > 64 else 0)
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
                              lanes_per_wave=(self.pool if (self.pool and not tail) else a.lanes_per_wave), staged=int(a.staged),
                              interleave_legs=a.interleave_legs, pipeline=pipe,
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


def queue_candidates(chains):
    """(steps in flight, chains per wavefront) of the fused kernel's CHAIN QUEUE the run calibrates beside depth_candidates(): a
    wavefront that owns 128 / 256 chains instead of 64 halves / quarters the wavefronts of a launch, so twice / four times the
    steps are kept in flight.  Only for batches that fill the GPU at depth 3 (a share of an N-GPU run is short of wavefronts
    as it is).  Measured (profiles/r06_queue_pool_sweep.json): smooth poses -5 / -6 % per step over 100 steps, iid -2 %; over the
    driver's 20 steps the deeper pipeline's fill and drain cost more than the queue saves -- which is why this is calibrated
    over the very region that is measured and not switched on."""
    if not chains or chains < 80000:
        return []
    # (pool 0 = the plain kernel one and two steps deeper than the depth-3 candidate: 11.28 against 11.48 ms per step at depth 4
    # over 100 steps, profiles/r05_n1_streams.jsonl)
    # ((4, 128) -- too few wavefronts in flight -- measured 13.6 ms against 11.7 and was dropped from the list)
    return [c for c in ((4, 0), (5, 0), (6, 128), (12, 256)) if c[0] <= MAX_DEPTH]


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


def measured_cost_floor(mix, n_all, simds, clock_hz, ms_per_step):
    """The same issue floor priced with MEASURED issue costs per instruction class (scripts/microbench/valu_issue.hip ->
    profiles/r03_valu_issue_costs.json, three wavefronts per SIMD): f64 add / mul / fma ~4.25 cycles per wavefront
    instruction, v_rcp_f64 / v_rsq_f64 ~16.2, and of the remaining vector instructions the share that profiles/
    r03_fused_isa.json finds to be f64-class / scalar-mask / 64-bit instructions (~4.25 cycles too) against plain 32-bit
    ones (~2.7).  None when the files are absent."""
    try:
        costs = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_issue_costs.json")))["classes"]
        isa_path = next(p for p in (os.path.join(ROOT, "profiles", f"{r}_fused_isa.json") for r in ("r06", "r05", "r04", "r03")) if os.path.exists(p))
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


def pmc_roofline(variant, staged, key, units_per_step, ms_per_step, device_index, pool=0):
    """The VALU-side roofline figures from the newest committed PMC summary (profiles/traffic_rNN*.json, written by
    scripts/summarize_profile.py) that matches this workload -- used only if it was measured on THIS build: the summary
    carries the sha256 of the solver kernels' sources, which must equal the sources the loaded library was built from.
    -> (traffic, valu, fp64, matches_build, file)"""
    suffix = ("_staged" if staged else "") + (f"_queue{pool}" if pool else "") + ("" if variant == "iid" else "_" + variant)
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



def quick_parity():
    """max |d theta| of the HIP serial walk against the committed reference fixtures (the shipped anipose outputs outside the LF
    singularity episode, the df3d reference-source run): fixtures only, no oracle.  `--detail` has the full report."""
    worst = 0.0
    for name in ("anipose_shipped", "df3d_1000"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        legs = [str(l) for l in z["legs"]]
        params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
        out = _lib.solve_seq(np.stack([z[f"{l}_pose"] for l in legs])[None], params, want_fk=False)
        err = np.abs(out["angles"][0] - np.stack([z[f"{l}_angles"] for l in legs]))
        if name == "anipose_shipped":
            err[legs.index("LF"), LF_WINDOW[0]:LF_WINDOW[1]] = 0.0
        worst = max(worst, float(err.max()))
    return worst


def round_numbers(v):
    """Numbers of the compact line: 6 significant digits (the full precision is in bench_detail.json)."""
    if isinstance(v, float):
        return float(f"{v:.6g}")
    if isinstance(v, dict):
        return {k: round_numbers(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [round_numbers(x) for x in v]
    return v


class Lifeline:
    """What an N > 1 run prints if it gets stuck.  Everything such a run does is a collective over the ranks, and a rank that
    fails where the others do not leaves them waiting for ever: the measurement that IS already made must still come out.
    Every rank arms the same deadline at the same points of the program (behind a collective); when it passes, rank 0 prints
    the best line there is so far -- `line_fn()` -- and every rank leaves with `exit_code`: EXIT_PROVISIONAL (75) while only
    the provisional headline exists (it never went through the bit-for-bit check of the headline, so the run must not look
    healthy), 0 once the verified headline is what gets printed and only a leg behind it did not finish."""

    EXIT_PROVISIONAL = 75

    def __init__(self, rank, json_fd):
        import threading
        self.rank, self.json_fd, self.deadline, self.line_fn, self.what, self.exit_code = rank, json_fd, None, None, "", self.EXIT_PROVISIONAL
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def arm(self, seconds, line_fn, what, exit_code):
        self.line_fn, self.what, self.exit_code, self.deadline = line_fn, what, exit_code, time.time() + seconds

    def disarm(self):
        self.deadline = None

    def _watch(self):
        while True:
            time.sleep(0.25)
            d = self.deadline
            if d is not None and time.time() > d:
                code = self.exit_code
                try:
                    if self.rank == 0 and self.line_fn is not None:
                        os.write(self.json_fd, (self.line_fn() + "\n").encode())
                    sys.stderr.write(f"bench.py rank {self.rank}: {self.what} did not finish in time -- the line measured so far "
                                     f"is printed, leaving with exit code {code}\n")
                except BaseException:  # noqa: BLE001  (no line could be made: that must not look like a result either)
                    code = code or 1
                finally:
                    if self.rank != 0 and code != 0:
                        time.sleep(1.5)   # the launcher ends every rank as soon as one fails: rank 0's line goes out first
                    os._exit(code)


class Ranks:
    """The process group of a run (or none) and the timed region every measurement of `bench.py` goes through."""

    def __init__(self, dist, world, rank, backend):
        self.dist, self.world, self.rank, self.backend = dist, world, rank, backend
        self.coll_dev = "cuda" if backend == "nccl" else "cpu"

    def sync_all(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def all_ok(self, ok):
        """Consensus: True only if `ok` on every rank (a collective every rank reaches whatever happened before it)."""
        if not self.dist:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item()) > 0.5

    def timed_region(self, bt, bufs, g, steps, warmup, events=None):
        """W untimed + K timed steps of batch `bt` (round-robin over its streams, gather `g` per step when there is one),
        bracketed by barrier + synchronize on both sides.  Returns (max over ranks, this rank's) seconds."""
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
        self.sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, events[i] if events else None, tail=in_lat_range(bt, i))
        if g:
            g.drain()
        self.sync_all()
        mine = time.perf_counter() - t0
        tmax = mine
        if self.dist:
            t = torch.tensor([mine], dtype=torch.float64, device=self.coll_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            tmax = float(t.item())
        return tmax, mine
