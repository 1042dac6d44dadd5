"""The legs of `bench.py` beyond its headline: imported only for `--detail` (N = 1: the per-rank shares of the fixed problem timed
on one GPU, config 3 as ONE recording, the full parity report, every BASELINE config through the reference-shaped API, the generic
chain) and for `--legs all` (N > 1: both gathers, the other scaling mode, the frame-sharded recording, config 5 over the ranks),
and for `--one-recording`.  Everything here lands in `bench_detail.json`; the compact line `bench.py` prints carries a handful of
scalars from it."""
import json
import os
import sys
import time

import numpy as np
import torch

from bench_support import (BYTES_PATH, HBM_PEAK_GBS, LATENCY_ROUND, LF_WINDOW, ROOT, Batch, depth_candidates, depth_cap, make_workload,
                           pmc_roofline, timed_steps, FP64_VECTOR_PEAK_TF)
from seqikpy_amd import _lib, data, peer_gather, sharding, synthetic, utils


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
    # (SEQIK_FRAME_LOCKSTEP=0: the round-2 protocol -- every slab settles itself first --, for the comparison in EXPERIMENTS.md 6.3)
    rec = frame_sharding.FrameShardedRecording(pose, params, want_fk=True, lockstep=os.environ.get("SEQIK_FRAME_LOCKSTEP", "1") != "0")

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
            "protocol": rec.stats.get("protocol", "settle, then resume"),
            "resume_calls_per_step_this_rank": rec.stats.get("resume_calls"),
            "data": "df3d locomotion recording (fixture, 1000 frames x 6 legs) repeated end to end",
            "exchange": "all-gather of 56 B end states per leg and rank + one padded all-gather of the joint angles; FK stays sharded",
            "check": chk}


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
                   "leg_frames_per_s = legs x frames / that; default = frame_parallel='auto', verified frame chunks (the default of "
                   "the Python API since round 6); serial_walk = frame_parallel=False, the reference's own order (bit-identical to "
                   "the C restatement); latency_floor_frac = issue floor of "
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
        for key, mode in (("serial_walk", False), ("default", "auto")):
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
                dd = np.abs(a - got["serial_walk"])
                e["max_abs_vs_serial_walk"] = float(dd[ok].max())
                if mask_lf and "LF" in legs:
                    e["max_abs_vs_serial_walk_incl_lf_window"] = float(dd.max())
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

    ref6 = np.stack([zd[f"{l}_angles"] for l in legs6])
    for key, mode in (("serial_walk_64_recordings_one_call", False), ("default_64_recordings_one_call", "auto")):
        def many():
            holder["res"] = run_ik_and_fk_many(recs, chain6, data.INITIAL_ANGLES_LOCOMOTION, frame_parallel=mode)
        ms_many = best_ms(many, reps=3)
        a_many = np.stack([np.stack([holder["res"][-1][0][f"Angle_{l}_{d}"] for d in DOFS], 1) for l in legs6])
        out["2"][key] = {
            "what": "run_ik_and_fk_many: 64 recordings x 6 legs x 1000 frames in one call, frame_parallel=%r (every recording's result "
                    "the bits it gets alone in that mode)" % (mode,), "ms": ms_many, "leg_frames_per_s": 64 * 6 * 1000 / ms_many * 1e3,
            "max_abs_dtheta_vs_fixture_last_recording": float(np.abs(a_many - ref6).max())}
    # ---- config 4: legs + head / antenna angles of the shipped 6000-frame recording in ONE submission ----------------
    e4, aligned4, chain4 = leg_entry(za, ["RF", "LF"], 6000, data.BOUNDS, data.INITIAL_ANGLES, data.NMF_TEMPLATE,
                                     "config 4: anipose_220525_aJO_Fly001_001 (6000 frames; stands in for the absent "
                                     "anipose_220807_Fly002_002), legs RF + LF + the 7 head / antenna angles", True)
    body_in = dict(aligned4, R_head=zh["R_head"], L_head=zh["L_head"], Neck=zh["Neck"])
    e4["legs_and_head_one_submission"] = {}
    for key, mode in (("serial_walk", False), ("default", "auto")):
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
    for entry, kernel_key, live_ms in ((out["4"], "config4_serial_walk", out["4"]["serial_walk"]["ms"]),
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



def one_recording_main(args, ranks):
    """`bench.py --one-recording`: config 3 read literally is the whole job of the run -- one recording, frame-sharded over the
    ranks.  Returns the record (rank 0 prints its compact line)."""
    world = ranks.world
    leg = one_recording_leg(ranks.dist, world, ranks.rank, args.frames, args.steps, args.warmup, ranks.coll_dev)
    units_rank0 = 6 * leg["frames_per_rank"][0]
    spec = leg["speculative_pass_ms_this_rank"]
    ach = BYTES_PATH * units_rank0 / (spec * 1e-3) / 1e9 if spec else None
    return {"metric": "leg-IK solves/s (frames x 6 legs); max |d theta| vs reference in `check`",
            "value": leg["value"], "unit": "leg-frame solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 3 literally: ONE recording of %d frames x 6 legs, contiguous frame slabs over "
                                   "the ranks (library frame chunks, end-state exchange, angle all-gather)" % args.frames,
                       "parallelism": f"frame-sharded x{world}" if world > 1 else "1 GPU", "backend": ranks.backend,
                       **{k: leg[k] for k in ("frames_per_rank", "frames_per_chunk", "run_in_frames", "boundary_rounds", "data", "exchange")}},
            "roofline": {"bound": "hbm", "kernel": "seqik_chunk_kernel<true, SPEC> (speculative pass of rank 0's slab)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None,
                         "traffic": None, "avg_launch_ms": spec, "bytes_per_unit": BYTES_PATH,
                         "note": "algorithmic bytes of rank 0's slab / duration of its speculative pass (HIP events on the "
                                 "launch stream); the path is FP64-issue-bound, see the default run's roofline"},
            "check": leg["check"]}


def multi_gpu_legs(ctx, multi):
    """`bench.py --gpus N --legs all`: the legs behind the verified headline and its one-GPU reference -- the same batch with the
    angle gather as peer writes / RCCL point-to-point / none, the other scaling mode, config 3 as ONE frame-sharded recording
    with its own one-GPU reference, config 5 over the ranks.  Every leg is a collective over all ranks: one that raises on any
    rank is dropped on ALL ranks by consensus and reported as an error string under its name."""
    args, ranks, batch, d_ang, params = ctx["args"], ctx["ranks"], ctx["batch"], ctx["d_ang"], ctx["params"]
    dist, world, rank = ranks.dist, ranks.world, ranks.rank
    S_total, T, L, units_all = ctx["S_total"], ctx["T"], ctx["L"], ctx["units_all"]
    n_streams, n_buf = len(batch.streams), len(d_ang)

    def guarded(name, fn):
        err = None
        try:
            res = fn()
        except Exception as exc:  # noqa: BLE001
            res, err = None, f"{type(exc).__name__}: {exc}"
        multi[name] = res if ranks.all_ok(err is None) else {"error": err or "failed on another rank"}

    def rank_ms(mine_s, k):
        got = [None] * world
        dist.all_gather_object(got, mine_s / k * 1e3)
        return {"min": min(got), "max": max(got), "by_rank": got}

    k_cmp = max(4, min(20, args.steps))
    headline_kind = "peer" if isinstance(ctx["gather"], peer_gather.PeerWriteGather) else "rccl"

    def leg_gather_compare():
        cmp_ = {"steps": k_cmp}
        for how in ("peer", "rccl"):
            g2, desc = peer_gather.make_gather(dist, world, rank, d_ang[0], n_buffers=n_buf, prefer=how)
            tm, _ = ranks.timed_region(batch, d_ang, g2, k_cmp, min(2, args.warmup))
            cmp_[how] = {"ms_per_step": tm / k_cmp * 1e3, "value": units_all * k_cmp / tm, "ran_as": desc}
            if hasattr(g2, "close"):
                g2.close()
            del g2
        tm, _ = ranks.timed_region(batch, d_ang, None, k_cmp, min(2, args.warmup))
        cmp_["no_gather"] = {"ms_per_step": tm / k_cmp * 1e3, "value": units_all * k_cmp / tm}
        return cmp_

    other_scaling = "strong" if args.scaling == "weak" else "weak"

    def leg_other_scaling():
        lo, hi, S_job2 = sharding.rank_share(S_total, world, rank, other_scaling)
        seed = synthetic.SEED_BASE + (1000 * rank if other_scaling == "weak" else 0)
        pose2 = make_workload(S_total, T, args.variant, seed)[2]
        if other_scaling == "strong":
            pose2 = pose2[lo:hi]
        units_all2 = S_job2 * L * T
        pad2 = max(b - a for a, b, _ in (sharding.rank_share(S_total, world, r, other_scaling) for r in range(world)))
        # its own depth: a batch that fills the GPU runs 3 deep on the library's kernel choice
        n_st2 = min(n_streams, depth_cap(int(pose2.shape[0]) * L))
        pipe2 = batch.pipeline if n_st2 > 3 else 0
        n_buf2 = max(2, n_st2 + 2)
        b2 = Batch(pose2, params, args, n_st2, pipeline=pipe2, s_pad=pad2)
        bufs2 = [b2.angle_buffer() for _ in range(n_buf2)]
        g2, desc = peer_gather.make_gather(dist, world, rank, bufs2[0], n_buffers=n_buf2, min_gbps=8.0, prefer=headline_kind)
        k2 = max(4, min(40, args.steps))
        tm, mine2 = ranks.timed_region(b2, bufs2, g2, k2, min(3, args.warmup))
        res = {"value": units_all2 * k2 / tm, "unit": "leg-frame solves/s", "ms_per_step": tm / k2 * 1e3,
               "steps": k2, "scaling": other_scaling, "streams": n_st2, "stage_pipeline": pipe2, "sequences_per_gpu": int(pose2.shape[0]),
               "leg_frames_per_step_all_ranks": int(units_all2), "gather": desc, "rank_ms_per_step": rank_ms(mine2, k2)}
        if hasattr(g2, "close"):
            g2.close()
        return res

    def leg_one_recording():
        torch.cuda.empty_cache()
        res = one_recording_leg(dist, world, rank, args.frames, max(3, min(10, args.steps)), 1, ranks.coll_dev)
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
        ranks.sync_all()
        mine = sc5.synthetic_sequences(a5)
        got = [None] * world
        dist.all_gather_object(got, {"seconds": mine["seconds"], "leg_frames": mine["leg_frames"], "pcie_GBps": mine["pcie_GBps_total"]})
        seq = {"value": sum(g_["leg_frames"] for g_ in got) / max(g_["seconds"] for g_ in got), "unit": "leg-frame solves/s",
               "leg_frames": sum(g_["leg_frames"] for g_ in got), "seconds_slowest_rank": max(g_["seconds"] for g_ in got),
               "by_rank": got, "what": "independent 64-frame sequences, 1/N of them per rank, each rank over its own PCIe link; "
                                       "PCIe-inclusive (H2D 120 B, D2H 272 B per leg-frame), alignment fused"}
        slab_r = max(1000, (frames5 // (2 * world) // 1000) * 1000)    # two slabs per rank; a multiple of the fixture's 1000 frames
        a5r = SimpleNamespace(frames=frames5, slab_frames=slab_r, slots=3, no_fk=False, gpu_stats=False)
        rec = sc5.one_recording_over_ranks_core(a5r, dist, world, rank, ranks.backend)
        box = [rec]
        dist.broadcast_object_list(box, src=0)
        return {"workload": "config 5: %d frames x 6 legs streamed from pinned host memory, AlignPose.align_leg fused, N GPUs" % frames5,
                "synthetic_sequences": seq, "one_recording": box[0]}

    guarded("gather_compare", leg_gather_compare)
    guarded(other_scaling, leg_other_scaling)
    guarded("one_recording", leg_one_recording)
    guarded("config5", leg_config5)
    n1 = multi.get("n1_reference")
    oth = multi.get(other_scaling)
    if n1 and "value" in n1 and oth and "value" in oth:
        oth["efficiency_vs_n1"] = oth["value"] / (world * n1["value"])


def strong_projection(ctx):
    """The fixed 1M-frame problem split N ways: the per-rank share timed on THIS GPU (no gather), at every candidate depth, + one
    job at a time, + the issue floor of the lone share.  A one-GPU PROJECTION of what a rank of an N-GPU run has to do, not a
    scaling measurement."""
    args, batch, params, pose, rec = ctx["args"], ctx["batch"], ctx["params"], ctx["pose"], ctx["rec"]
    S, L = pose.shape[0], ctx["L"]
    explicit_depth = args.streams > 0
    proj = {"note": "per-rank share of the fixed problem (S/N sequences) timed on ONE GPU; no gather; projected_value = 6M leg-frames / "
                    "that time; `streams` / `stage_pipeline`: the fastest of the depth candidates for that share; lone_job_ms = ONE launch "
                    "at a time (the library's own kernel choice)", "by_n_gpus": {}}
    for n in (2, 4, 8):
        best, tried = None, []
        for n_st, pipe, lat in (depth_candidates(args.steps, (S // n) * L) if not explicit_depth else ((len(batch.streams), batch.pipeline, None),)):
            sub = Batch(pose[: S // n], params, args, n_st, pipeline=pipe)
            sub.lat_range = lat
            bufs = [sub.angle_buffer() for _ in range(len(sub.streams))]
            k = args.steps             # the same region as the headline's: fill and drain of the pipeline included
            dt = timed_steps(sub, bufs, k, len(sub.streams), warmup=args.warmup)
            row = {"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat, "ms_per_step": dt / k * 1e3,
                   "projected_value": batch.units / (dt / k), "speedup_vs_1": (ctx["elapsed"] / args.steps) / (dt / k), "chains_per_gpu": sub.S * L}
            tried.append({"streams": n_st, "stage_pipeline": pipe, "latency_kernel_steps": lat, "ms_per_step": row["ms_per_step"]})
            if best is None or row["ms_per_step"] < best["ms_per_step"]:
                best = row
            del sub, bufs
        best["candidates"] = tried
        sub = Batch(pose[: S // n], params, args, 1, pipeline=0)     # ONE call: the library's own choice of kernel
        bufs = [sub.angle_buffer()]
        k = max(8, min(40, args.steps // 2))
        dt = timed_steps(sub, bufs, k, 1, warmup=2)
        best["lone_job_ms"] = dt / k * 1e3
        best["lone_job_speedup_vs_single_job"] = rec["single_job"]["ms_per_step"] / best["lone_job_ms"]
        best["ideal_ms"] = ctx["ms_per_step"] / n
        best["efficiency"] = best["speedup_vs_1"] / n
        fl = share_floor(n)
        if fl:
            best["lone_job_issue_floor_ms"] = fl["issue_floor_ms"]
            best["lone_job_issue_floor_frac"] = fl["issue_floor_ms"] / best["lone_job_ms"]
            best["floor_source"] = fl["source"]
        del sub, bufs
        proj["by_n_gpus"][str(n)] = best
    return proj


def detail_legs(ctx):
    """`bench.py --detail` at N = 1, behind the headline: the share projection (BEFORE the legs that create streams of their own:
    run behind them the same pipelines share queues with those streams), config 3 as ONE recording, the full parity report,
    every BASELINE config through the reference-shaped API.  Adds to ctx["rec"]; a handful of scalars go into `detail_scalars` of
    the compact line."""
    args, rec = ctx["args"], ctx["rec"]
    rec["strong_projection"] = strong_projection(ctx)
    del ctx["d_ang"][1:]
    torch.cuda.empty_cache()
    rec["single_recording"] = single_recording(args.frames)
    rec["parity"] = par = parity_report()
    rec["configs"] = cf = reference_configs()
    cf["3"].update({"leg_frames": ctx["batch"].units, "ms_per_step_pipelined": ctx["ms_per_step"], "leg_frames_per_s": rec["value"],
                    "ms_one_job_at_a_time": rec["single_job"]["ms_per_step"], "leg_frames_per_s_one_job_at_a_time": rec["value_single_job"],
                    "smooth_variant_leg_frames_per_s": rec.get("value_smooth"),
                    "one_recording_1M_frames_leg_frames_per_s": rec["single_recording"]["value"]})

    def dig(obj, *path):
        for k in path:
            if not isinstance(obj, dict) or k not in obj:
                return None
            obj = obj[k]
        return obj
    names = ("anipose_shipped", "df3d_1000")
    rec["detail_scalars"] = {
        "config1_default_ms": dig(cf, "1", "default", "ms"), "config1_serial_ms": dig(cf, "1", "serial_walk", "ms"),
        "config2_default_ms": dig(cf, "2", "default", "ms"), "config2_serial_ms": dig(cf, "2", "serial_walk", "ms"),
        "config2_64_recordings_per_s": dig(cf, "2", "default_64_recordings_one_call", "leg_frames_per_s"),
        "config2_64_recordings_serial_per_s": dig(cf, "2", "serial_walk_64_recordings_one_call", "leg_frames_per_s"),
        "config4_default_ms": dig(cf, "4", "default", "ms"), "config4_serial_ms": dig(cf, "4", "serial_walk", "ms"),
        "config5_per_s": dig(cf, "5", "one_recording", "value"), "config5_sequences_per_s": dig(cf, "5", "synthetic_sequences", "value"),
        "generic_6000_frames_s": (dig(cf, "generic", "ms") or 0.0) / 1e3 or None,
        "generic_batch_per_s": dig(cf, "generic", "batch", "leg_frames_per_s"),
        "head_kernel_hbm_frac": dig(cf, "4", "head_kernel", "roofline", "frac"),
        "head_kernel_frac_of_box_copy": dig(cf, "4", "head_kernel", "frac_of_box_copy"),
        "parity_p99.9": max(par[n]["serial_walk"]["p99.9_abs_dtheta"] for n in names),
        "parity_values_over_5e-5": sum(par[n]["serial_walk"]["values_over_5e-5"] for n in names),
        "parity_over_1e-4_outside_lf_window": sum(par[n]["serial_walk"]["leg_frames_over_1e-4_outside_lf_window"] for n in names),
        "parity_auto_max_abs_dtheta": max(par[n]["frame_chunks"]["max_abs_dtheta"] for n in names),
        "single_recording_per_s": dig(rec, "single_recording", "value"),
        "share_n8_ms_per_step": dig(rec, "strong_projection", "by_n_gpus", "8", "ms_per_step"),
        "share_n8_projected_speedup": dig(rec, "strong_projection", "by_n_gpus", "8", "speedup_vs_1")}
