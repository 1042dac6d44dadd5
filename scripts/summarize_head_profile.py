#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of scripts/gpu_head_profile.sh (gpurun_out/<tag>_head_{stats,fetch,write}) into
profiles/<rnd>_head_profile.json + <rnd>_head_kernel_stats.csv: average duration of the head / antenna kernel and its HBM
traffic per frame (FETCH_SIZE doubled for the 16-byte-per-lane streaming loads, as MI355X_MICROARCH.md prescribes for
gfx950; WRITE_SIZE is exact for streaming stores).

    python scripts/summarize_head_profile.py TAG r05"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    run = json.loads([l for l in open(f"{src}/{tag}_head_stats.log") if l.startswith('{"kernel"')][-1])
    stats = pd.read_csv(newest(f"{src}/{tag}_head_stats/*/*_kernel_stats.csv"))
    stats.to_csv(f"{dst}/{rnd}_head_kernel_stats.csv", index=False)
    row = stats[stats.Name.str.contains("seqik_head_kernel")].iloc[0]

    def pmc(kind):
        d = pd.read_csv(newest(f"{src}/{tag}_head_{kind}/*/*_counter_collection.csv"))
        d = d[d.Kernel_Name.str.contains("seqik_head_kernel")]
        return d.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum").mean()

    f, w = pmc("fetch"), pmc("write")
    n = run["frames"]
    fetch_raw, write = float(f["FETCH_SIZE"]) * 1024, float(w["WRITE_SIZE"]) * 1024
    out = {"source": f"bash scripts/gpu_head_profile.sh {tag} {n}: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in "
                     "their own passes, on python3 scripts/bench_head.py",
           "kernel": "seqik_head_kernel<true, false> (staged 16-byte-per-lane loads through LDS, non-temporal)", "frames": n,
           "calls": int(row.Calls), "average_ns": float(row.AverageNs), "min_ns": float(row.MinNs), "max_ns": float(row.MaxNs),
           "FETCH_SIZE_per_launch": float(f["FETCH_SIZE"]), "TCC_EA0_RDREQ_sum_per_launch": float(f["TCC_EA0_RDREQ_sum"]),
           "TCC_EA0_WRREQ_sum_per_launch": float(w["TCC_EA0_WRREQ_sum"]), "WRITE_SIZE_per_launch": float(w["WRITE_SIZE"]),
           "fetch_bytes_per_launch_raw": fetch_raw, "fetch_bytes_per_launch_corrected": 2 * fetch_raw, "write_bytes_per_launch": write,
           "note": "FETCH_SIZE is in KiB and on gfx950 reports half the bytes of wide (16 B per lane) coalesced streaming reads "
                   "(MI355X_MICROARCH.md): doubled. WRITE_SIZE is exact for streaming stores.",
           "traffic_bytes_per_frame": (2 * fetch_raw + write) / n, "algorithmic_bytes_per_frame": 152,
           "achieved_GBps_at_average_duration": 152.0 * n / float(row.AverageNs),
           "frac_of_8TBps": 152.0 * n / float(row.AverageNs) / 8000.0,
           "live_run_under_kernel_trace": {k: run[k] for k in ("ms", "ms_min", "frac", "frac_best_launch") if k in run}}
    json.dump(out, open(f"{dst}/{rnd}_head_profile.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
