"""Turns the rocprofv3 CSVs of one bench run (gpurun_out/<tag>_{stats,fetch,write,sq}) into the small
summaries committed under profiles/ (+ profiles/traffic_rNN.json that bench.py reads)."""
import glob, json, os, sys

def newest(pattern):
    """The most recent match (gpurun merges every run's files into the same directories)."""
    return max(glob.glob(pattern), key=os.path.getmtime)

import pandas as pd

tag, rnd = sys.argv[1], sys.argv[2]           # e.g. r01b r01
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = pd.read_csv(newest(f"{src}/{tag}_stats/*/*_kernel_stats.csv"))
bench_line = [l for l in open(f"{src}/{tag}_stats.log") if l.startswith('{"metric"')][0]
variant = json.loads(bench_line)["config"]["variant"]
vtag = "" if variant == "iid" else "_" + variant
stats.to_csv(f"{dst}/{rnd}_bench{vtag}_kernel_stats.csv", index=False)
open(f"{dst}/{rnd}_bench{vtag}_under_rocprof.json", "w").write(bench_line)
# Round 5: the default run calibrates its pipeline depth first (3 / 8 / 12 / 16 steps in flight), and a launch lasts as many
# times longer as launches share the GPU -- so rocprofv3's per-kernel average mixes four depths.  The figure that must agree
# with the bench line's `roofline.avg_launch_ms` is the average over the launches of the TIMED REGION: with --no-extras the
# solver kernel's dispatches are [calibration ...][warm-up][`steps` timed launches][one launch alone = the verification], so
# the timed region is the `steps` launches in front of the last one (kernel trace, in dispatch order).
try:
    tr = pd.read_csv(newest(f"{src}/{tag}_stats/*/*_kernel_trace.csv")).sort_values("Start_Timestamp")
    b_ = json.loads(bench_line)
    # (round 6: the run also calibrates the chain queue -- seqik_fused_queue_kernel; the timed region ran the kernel the line names)
    timed_kernel = "seqik_fused_queue_kernel" if "queue" in str(b_["roofline"].get("kernel")) else "seqik_fused_kernel|seqik_pipe_kernel"
    k = tr[tr.Kernel_Name.str.contains(timed_kernel)]
    timed = k.iloc[-(b_["steps"] + 1):-1]
    dur = (timed.End_Timestamp - timed.Start_Timestamp) / 1e6
    region = {"kernel": str(timed.Kernel_Name.iloc[0])[:120], "launches": int(len(timed)), "avg_launch_ms": float(dur.mean()),
              "min_launch_ms": float(dur.min()), "max_launch_ms": float(dur.max()),
              "span_ms_first_start_to_last_end": float((timed.End_Timestamp.max() - timed.Start_Timestamp.min()) / 1e6),
              "bench_line_avg_launch_ms": b_["roofline"].get("avg_launch_ms"), "bench_line_ms_per_step": b_["ms_per_step"],
              "all_launches_of_the_run": int(len(k)), "streams": b_["config"].get("streams"),
              "source": "rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --no-extras`: the `steps` launches in front of the "
                        "last one (the verification launch made alone)"}
    json.dump(region, open(f"{dst}/{rnd}_bench{vtag}_timed_region_kernel.json", "w"), indent=1)
    print(json.dumps(region))
except Exception as exc:  # noqa: BLE001
    print("timed-region summary skipped:", exc)

def pmc(kind):
    d = pd.read_csv(newest(f"{src}/{tag}_{kind}/*/*_counter_collection.csv"))
    d = d[d.Kernel_Name.str.contains("seqik_stage_kernel|seqik_fused_kernel")].copy()
    st = d.Kernel_Name.str.extract(r"seqik_stage_kernel<(\d), ")[0]
    d["kernel"] = ("stage" + st).where(st.notna(), "fused")
    return d.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean")

fetch, write, sq = pmc("fetch"), pmc("write"), pmc("sq")
f64 = pmc("f64") if glob.glob(f"{src}/{tag}_f64/*/*_counter_collection.csv") else None
pm = pd.concat([fetch, write, sq] + ([f64] if f64 is not None else []), axis=1)
pm.to_csv(f"{dst}/{rnd}_bench{vtag}_pmc_per_launch.csv")
b = json.loads(bench_line)
units = b["config"]["sequences_per_gpu"] * b["config"]["legs"] * b["config"]["frames_per_sequence"]
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
from seqikpy_amd import _lib  # noqa: E402
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) on `python3 bench.py --no-cpu-baseline`",
       "units_per_launch": units, "variant": b["config"]["variant"],
       # the build these counters belong to: bench.py uses them only while the solver kernels' sources are unchanged
       "csrc_sha256": _lib.csrc_sha256(), "csrc_files": _lib.KERNEL_SOURCES,
       "note": "FETCH_SIZE / WRITE_SIZE are in KiB; bytes = value * 1024, mean over the launches of the run. "
               "FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950 and under-reports wide (16 B/lane) streaming reads by 2x "
               "(MI355X_MICROARCH.md); these kernels issue 8-byte per-lane loads, for which the counter is uncalibrated, "
               "so the raw value is reported.  The fused kernel's traffic includes the stage hand-off workspace "
               "(3 x (96 B written + 96 B read) per leg-frame), which is not part of the 392 algorithmic bytes."}
for k in pm.index:
    f, w = float(fetch.loc[k, "FETCH_SIZE"]) * 1024, float(write.loc[k, "WRITE_SIZE"]) * 1024
    out[f"{k}_fetch_bytes_per_launch"] = f
    out[f"{k}_write_bytes_per_launch"] = w
    out[f"{k}_hbm_bytes_per_launch"] = f + w
    # wave-level VALU instructions issued per launch, and the share of the 64 lanes that were active in them
    out[f"{k}_valu_insts_per_launch"] = float(sq.loc[k, "SQ_INSTS_VALU"])
    out[f"{k}_valu_lane_utilisation"] = float(sq.loc[k, "SQ_THREAD_CYCLES_VALU"]) / (64.0 * float(sq.loc[k, "SQ_ACTIVE_INST_VALU"]))
    if f64 is not None:  # wave-level FP64 instructions by class (FMA counts two flops)
        for c in ("ADD", "MUL", "FMA", "TRANS"):
            out[f"{k}_f64_{c.lower()}_insts_per_launch"] = float(f64.loc[k, f"SQ_INSTS_VALU_{c}_F64"])
suffix = ("_staged" if "stage1" in pm.index else "") + ("" if b["config"]["variant"] == "iid" else "_" + b["config"]["variant"])
json.dump(out, open(f"{dst}/traffic_{rnd}{suffix}.json", "w"), indent=1)
print(pm.round(0).to_string())
print(json.dumps({k: round(v / units, 1) for k, v in out.items() if k.endswith("bytes_per_launch")}, indent=0))


# ---- round 6: the chain-queue launches of the same run (bench.py calibrates them beside the plain kernel): one summary per pool size,
# told apart by the grid (ceil(sequences / pool) x legs wavefronts of 64 lanes) -> profiles/traffic_<rnd>_queue<pool>[_variant].json
def pmc_queue(kind, grid):
    d = pd.read_csv(newest(f"{src}/{tag}_{kind}/*/*_counter_collection.csv"))
    d = d[d.Kernel_Name.str.contains("seqik_fused_queue_kernel") & (d.Grid_Size == grid)]
    return d.pivot_table(index="Kernel_Id", columns="Counter_Name", values="Counter_Value", aggfunc="mean").mean(axis=0) if len(d) else None

try:
    S_, L_ = b["config"]["sequences_per_gpu"], b["config"]["legs"]
    allq = pd.read_csv(newest(f"{src}/{tag}_sq/*/*_counter_collection.csv"))
    grids = sorted(set(allq[allq.Kernel_Name.str.contains("seqik_fused_queue_kernel")].Grid_Size))
    for grid in grids:
        pools = [p_ for p_ in range(128, 4097, 64) if -(-S_ // p_) * L_ * 64 == grid]
        if not pools:
            continue
        pool = pools[0]
        parts = {kind: pmc_queue(kind, grid) for kind in ("fetch", "write", "sq", "f64")}
        if any(v is None for v in parts.values()):
            continue
        f, w = float(parts["fetch"]["FETCH_SIZE"]) * 1024, float(parts["write"]["WRITE_SIZE"]) * 1024
        q = dict(out, kernel="seqik_fused_queue_kernel<true>", chain_queue=pool)
        q.update({"fused_fetch_bytes_per_launch": f, "fused_write_bytes_per_launch": w, "fused_hbm_bytes_per_launch": f + w,
                  "fused_valu_insts_per_launch": float(parts["sq"]["SQ_INSTS_VALU"]),
                  "fused_valu_lane_utilisation": float(parts["sq"]["SQ_THREAD_CYCLES_VALU"]) / (64.0 * float(parts["sq"]["SQ_ACTIVE_INST_VALU"]))})
        for c in ("ADD", "MUL", "FMA", "TRANS"):
            q[f"fused_f64_{c.lower()}_insts_per_launch"] = float(parts["f64"][f"SQ_INSTS_VALU_{c}_F64"])
        vt = "" if b["config"]["variant"] == "iid" else "_" + b["config"]["variant"]
        json.dump(q, open(f"{dst}/traffic_{rnd}_queue{pool}{vt}.json", "w"), indent=1)
        print(f"chain queue, pool {pool}: {q['fused_valu_insts_per_launch']:.4g} VALU instructions per launch, lanes {q['fused_valu_lane_utilisation']:.3f}, "
              f"HBM {q['fused_hbm_bytes_per_launch'] / units:.0f} B per leg-frame")
except Exception as exc:  # noqa: BLE001
    print("chain-queue summary skipped:", exc)
