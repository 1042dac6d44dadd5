#!/usr/bin/env python3
"""AlignPose statistics (the constants of the fused alignment): host numpy vs seqik_align_stats_* on the GPU.

    python scripts/bench_align_stats.py --frames 2000000          (needs a GPU; prints one JSON line)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from seqikpy_amd import _lib, data  # noqa: E402
from seqikpy_amd.alignment import AlignPose  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2_000_000)
    ap.add_argument("--host-frames", type=int, default=0, help="frames of the host timing (0 = all)")
    args = ap.parse_args()
    legs = data.LEGS
    rng = np.random.default_rng(3)
    n = args.frames
    raw = {f"{l}_leg": rng.normal(size=(n, 5, 3)) for l in legs}
    al = AlignPose(raw, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    _lib.load()
    al_small = AlignPose({k: v[:1000] for k, v in raw.items()}, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    al_small.leg_affines(on_gpu=True)  # warm-up: library load, first launches
    t0 = time.perf_counter()
    dev = al.leg_affines(on_gpu=True)
    t_gpu = time.perf_counter() - t0
    # device-resident variant: the slab is already on the GPU (streaming pass)
    pose = torch.from_numpy(np.stack([raw[f"{l}_leg"] for l in legs])[None]).cuda()
    ranks = [int(np.floor((n - 1) * 0.45)), int(np.floor((n - 1) * 0.45)) + 1, int(np.floor((n - 1) * 0.55)), int(np.floor((n - 1) * 0.55)) + 1]
    with _lib.AlignStats(len(legs), n) as st:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.add(pose.data_ptr(), n_seq=1, n_frames=n, on_device=True)
        st.finish(ranks)
        t_gpu_resident = time.perf_counter() - t0
    hn = args.host_frames or n
    al_h = al if hn == n else AlignPose({k: v[:hn] for k, v in raw.items()}, legs, body_template=data.TEMPLATE_NMF_LOCOMOTION, log_level="ERROR")
    t0 = time.perf_counter()
    host = al_h.leg_affines()
    t_host = time.perf_counter() - t0
    same = hn == n and all(np.array_equal(host[l][0], dev[l][0]) and host[l][1] == dev[l][1] for l in legs)
    print(json.dumps({"frames": n, "legs": len(legs), "gpu_seconds_from_host_arrays": t_gpu,
                      "gpu_seconds_device_resident": t_gpu_resident, "host_numpy_seconds": t_host, "host_frames": hn,
                      "host_seconds_per_million_frames": t_host / hn * 1e6,
                      "gpu_resident_seconds_per_million_frames": t_gpu_resident / n * 1e6,
                      "bit_identical_constants": bool(same) if hn == n else None}))


if __name__ == "__main__":
    main()
