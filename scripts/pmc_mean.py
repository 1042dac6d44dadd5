#!/usr/bin/env python3
"""Mean counter value per dispatch, per kernel, of a rocprofv3 --pmc run: python scripts/pmc_mean.py DIR [kernel-substring]"""
import glob, sys
import pandas as pd
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "seqik_fused")
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(sub)]
    g = df.groupby(["Kernel_Name", "Counter_Name"]).Counter_Value.agg(["mean", "count"])
    for (k, c), r in g.iterrows():
        print(f"{k[:60]:60s} {c:28s} mean {r['mean']:.6g}  dispatches {int(r['count'])}")
