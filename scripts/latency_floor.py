#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of scripts/gpu_latency_profile.sh (gpurun_out/<tag>_lat_{stats,sq,f64}) into
profiles/<rnd>_latency_floor.json: for the two launches whose time is the LATENCY of one dependent chain -- config 4's
serial walk on the stage pipeline and the generic chain -- the issue floor of their critical wavefront and how close the
measured kernel time is to it.

Model (round-3 review, item 7: "latency-bound" as a number).  A lone wavefront issues one vector instruction every
~4.6-4.9 cycles whatever its class (profiles/r03_valu_issue_costs.json, `waves_per_simd_1`: f64 add / mul / fma 4.9,
v_rcp_f64 / v_rsq_f64 16.7, everything else ~4.65 -- a second wavefront on the SIMD would fill the gaps, a single chain
has none to offer).  The floor of a launch that is ONE wavefront per chain is therefore
    floor = (f64 arithmetic x 4.9 + f64 transcendental x 16.7 + other vector instructions x 4.65) / shader clock,
with the instruction counts of that wavefront from the PMC passes (SQ_INSTS_VALU, SQ_INSTS_VALU_*_F64 / SQ_WAVES).
  * generic chain: one kernel, one wavefront per chain (lane groups of 8 inside it) -> its own counters.
  * stage pipeline: four wavefronts per chain, stage k + 1 following stage k one frame behind; the walk cannot be
    faster than its slowest stage, whose instruction stream is what the per-stage kernel of the same chain issues (launch
    C of latency_kernels_run.py, no waiting loops) -> floor = max over stages.  The comparison is with launch B (lane
    pairs off: the same instruction streams); launch A (lane pairs on, the product) splits the two joints of a pass over
    two lanes and is reported beside it.
latency_floor_frac = floor / measured kernel time: the closer to 1, the less there is to gain without a shorter
instruction stream.

    python scripts/latency_floor.py TAG r04"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
from seqikpy_amd import _lib  # noqa: E402

COST = {"f64_arith": 4.9, "f64_trans": 16.7, "other": 4.65}


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out")
    run_line = json.loads([l for l in open(f"{src}/{tag}_lat_stats.log") if l.startswith('{"frames"')][-1])
    reps, frames = run_line["reps"], run_line["frames"]
    trace = pd.read_csv(newest(f"{src}/{tag}_lat_stats/*/*_kernel_trace.csv"))
    trace["ms"] = (trace["End_Timestamp"] - trace["Start_Timestamp"]) / 1e6
    trace = trace.sort_values("Start_Timestamp")

    def pmc(kind):
        d = pd.read_csv(newest(f"{src}/{tag}_lat_{kind}/*/*_counter_collection.csv"))
        return d.sort_values("Dispatch_Id")

    sq, f64 = pmc("sq"), pmc("f64")
    clock_khz = _lib.device_attributes(0)[1] if _lib.load().seqik_device_count() > 0 else 2400000
    clock_hz = clock_khz * 1e3

    def per_dispatch(df, pattern):
        """list (dispatch order) of {counter: value} for kernels whose name matches"""
        d = df[df.Kernel_Name.str.contains(pattern, regex=True)]
        out = []
        for _, g in d.groupby("Dispatch_Id", sort=True):
            out.append({r.Counter_Name: float(r.Counter_Value) for r in g.itertuples()})
        return out

    def wave_counts(s, f):
        waves = s["SQ_WAVES"]
        arith = (f["SQ_INSTS_VALU_ADD_F64"] + f["SQ_INSTS_VALU_MUL_F64"] + f["SQ_INSTS_VALU_FMA_F64"]) / waves
        trans = f["SQ_INSTS_VALU_TRANS_F64"] / waves
        valu = s["SQ_INSTS_VALU"] / waves
        return {"waves": waves, "valu_insts_per_wave": valu, "f64_arith_per_wave": arith, "f64_trans_per_wave": trans,
                "other_valu_per_wave": valu - arith - trans, "salu_insts_per_wave": s.get("SQ_INSTS_SALU", 0.0) / waves,
                "lds_insts_per_wave": s.get("SQ_INSTS_LDS", 0.0) / waves}

    def floor_ms(c):
        cycles = c["f64_arith_per_wave"] * COST["f64_arith"] + c["f64_trans_per_wave"] * COST["f64_trans"] + \
            c["other_valu_per_wave"] * COST["other"]
        return cycles / clock_hz * 1e3

    out = {"source": "rocprofv3 --kernel-trace and --pmc (own passes) of scripts/latency_kernels_run.py; issue costs of a lone "
                     "wavefront from profiles/r03_valu_issue_costs.json (waves_per_simd_1)",
           "csrc_sha256": _lib.csrc_sha256(_lib.LATENCY_SOURCES), "csrc_files": _lib.LATENCY_SOURCES,
           "issue_cycles_per_instruction_lone_wavefront": COST, "clock_MHz": clock_khz / 1e3, "frames": frames,
           "host_wall_ms": {k: v for k, v in run_line.items() if k.endswith("_ms")}}

    # ---- generic chain --------------------------------------------------------------------------------------------
    g_sq, g_f = per_dispatch(sq, "seqik_generic_kernel"), per_dispatch(f64, "seqik_generic_kernel")
    g_ms = trace[trace.Kernel_Name.str.contains("seqik_generic_kernel")]["ms"]
    c = wave_counts(g_sq[-1], g_f[-1])
    fl = floor_ms(c)
    out["generic_rf_6000"] = {"kernel": "seqik_generic_kernel<diag = 0, grouped = 1>", "kernel_ms": float(g_ms.min()),
                              "us_per_frame": float(g_ms.min()) * 1e3 / frames, **c, "valu_insts_per_frame": c["valu_insts_per_wave"] / frames,
                              "issue_floor_ms": fl, "latency_floor_frac": fl / float(g_ms.min())}

    # ---- stage pipeline -------------------------------------------------------------------------------------------
    p_ms = trace[trace.Kernel_Name.str.contains("seqik_pipe_kernel")]["ms"].tolist()
    a_ms, b_ms = min(p_ms[:reps]), min(p_ms[reps:2 * reps])
    stages = {}
    for st in (1, 2, 3, 4):
        s_sq = per_dispatch(sq, rf"seqik_stage_kernel<{st}, ")
        s_f = per_dispatch(f64, rf"seqik_stage_kernel<{st}, ")
        s_ms = trace[trace.Kernel_Name.str.contains(rf"seqik_stage_kernel<{st}, ", regex=True)]["ms"].tolist()
        # launch C's dispatches are the FIRST `reps` of every stage kernel (launch F, the strong-scaling share, comes later)
        c = wave_counts(s_sq[reps - 1], s_f[reps - 1])
        stages[str(st)] = {**c, "valu_insts_per_frame": c["valu_insts_per_wave"] / frames, "issue_floor_ms": floor_ms(c),
                           "stage_kernel_alone_ms": float(min(s_ms[:reps]))}
    crit = max(stages, key=lambda k: stages[k]["issue_floor_ms"])
    fl = stages[crit]["issue_floor_ms"]
    p_sq = per_dispatch(sq, "seqik_pipe_kernel")
    out["config4_serial_walk"] = {
        "kernel": "seqik_pipe_kernel<fk = 1, wpe = 2> (anipose RF + LF x 6000 frames, one workgroup of four stage wavefronts per leg)",
        "kernel_ms_lane_pairs_on": a_ms, "kernel_ms_lane_pairs_off": b_ms, "us_per_frame_lane_pairs_on": a_ms * 1e3 / frames,
        "per_stage_one_wavefront": stages, "critical_stage": int(crit), "issue_floor_ms": fl,
        "sum_of_stage_floors_ms": sum(s["issue_floor_ms"] for s in stages.values()),
        "latency_floor_frac": fl / b_ms, "latency_floor_frac_vs_product_launch": fl / a_ms,
        "pipe_kernel_valu_insts_per_wave_pairs_on": p_sq[reps - 1]["SQ_INSTS_VALU"] / p_sq[reps - 1]["SQ_WAVES"],
        "pipe_kernel_valu_insts_per_wave_pairs_off": p_sq[2 * reps - 1]["SQ_INSTS_VALU"] / p_sq[2 * reps - 1]["SQ_WAVES"],
        "note": "floor = the critical stage's instruction stream as the per-stage kernel issues it (no lane pairs, Jacobian re-derived "
                "in every pass) at the lone-wavefront issue rate.  Both pipeline launches skip the re-derivation after a rejected "
                "trial (run_stage REUSE: stage 1 rejects 32 % / 10 % of its trials on RF / LF), and the product launch (lane pairs "
                "on) also splits the two joints of a pass over two lanes: they issue fewer instructions than that stream, so their "
                "fractions are upper estimates of how close they are to THEIR floors"}
    # ---- the lone 1/8 share of BASELINE config 3 (launches E / F): what ONE rank of an 8-GPU strong-scaling run does per step
    if "E_share_piped_ms" in run_line:
        n_seq, n_ch = run_line["share_sequences"], run_line["share_chains"]
        # the share's dispatches come AFTER config 4's in every CSV: E = the last `reps` pipe launches, F = the last `reps`
        # launches of every stage kernel
        e_ms = min(p_ms[-reps:])
        sh = {}
        for st in (1, 2, 3, 4):
            s_sq = per_dispatch(sq, rf"seqik_stage_kernel<{st}, ")
            s_f = per_dispatch(f64, rf"seqik_stage_kernel<{st}, ")
            s_ms = trace[trace.Kernel_Name.str.contains(rf"seqik_stage_kernel<{st}, ", regex=True)]["ms"].tolist()
            c = wave_counts(s_sq[-1], s_f[-1])
            sh[str(st)] = {**c, "issue_floor_ms": floor_ms(c), "stage_kernel_alone_ms": float(min(s_ms[-reps:]))}
        crit8 = max(sh, key=lambda k: sh[k]["issue_floor_ms"])
        out["strong_share_8"] = {
            "kernel": f"seqik_pipe_kernel<fk = 1> on {n_seq} sequences x 6 legs x 64 frames (synthetic iid, planar): {n_ch} chains, "
                      "64 per workgroup of four stage wavefronts, one launch alone on the GPU",
            "kernel_ms": e_ms, "per_stage_mean_wavefront": sh, "critical_stage": int(crit8),
            "issue_floor_ms": sh[crit8]["issue_floor_ms"],
            "sum_of_stage_floors_ms": sum(v["issue_floor_ms"] for v in sh.values()),
            "latency_floor_frac": sh[crit8]["issue_floor_ms"] / e_ms,
            "note": "floor = instruction stream of the MEAN stage-1 wavefront (SQ_INSTS_VALU / SQ_WAVES of the stage kernel) at the "
                    "lone-wavefront issue rate; the launch lasts as long as its SLOWEST wavefront, so the fraction is a lower "
                    "estimate of how close the launch is to its floor.  ideal share of the pipelined 1-GPU step: 1/8 of it"}
    dst = os.path.join(ROOT, "profiles", f"{rnd}_latency_floor.json")
    json.dump(out, open(dst, "w"), indent=1)
    trace[trace.Kernel_Name.str.contains("seqik_")][["Kernel_Name", "ms"]].to_csv(os.path.join(ROOT, "profiles", f"{rnd}_latency_kernel_trace.csv"), index=False)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)}) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
