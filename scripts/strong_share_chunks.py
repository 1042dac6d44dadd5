#!/usr/bin/env python3
"""A 1/N share of the fixed 1M-frame problem (bench.py --scaling strong) does not fill a GPU: what do frame chunks INSIDE
the 64-frame sequences buy?  Times S/N sequences x 6 legs x 64 frames with several launches in flight, serial walk vs
chunked (chunk + run-in), and reports the chunk statistics and the distance from the serial walk.  Needs a GPU.

    python scripts/strong_share_chunks.py [--n 8] [--variant smooth]
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

from seqikpy_amd import _lib, data, synthetic, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--variant", default="smooth")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    T, S = 64, 15625 // a.n
    pose = synthetic.synthetic_pose(15625, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant=a.variant,
                                    seed=synthetic.SEED_BASE)[:S]
    params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    d_ang = [torch.zeros((S, 6, 7, T), dtype=torch.float64, device="cuda") for _ in streams]
    d_fk = [torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda") for _ in streams]
    d_stats = torch.zeros(_lib.N_CHUNK_STATS, dtype=torch.int32, device="cuda")
    lay = _lib.planar_layout(T)
    serial = None
    for kw in (dict(), dict(frame_chunk=32, frame_halo=8), dict(frame_chunk=16, frame_halo=8), dict(frame_chunk=16, frame_halo=4),
               dict(frame_chunk=8, frame_halo=8), dict(frame_chunk=8, frame_halo=4), dict(frame_chunk=-1)):
        def launch(i):
            k = i % a.streams
            _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang[k].data_ptr(), d_fk[k].data_ptr(), layout=lay,
                                  stream=streams[k].cuda_stream, d_chunk_stats=d_stats.data_ptr() if kw else 0, **kw)
        for i in range(a.streams):
            launch(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            launch(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        ang = d_ang[0].cpu().numpy()
        if serial is None:
            serial = ang
        st = _lib.chunk_stats_dict(d_stats.cpu().numpy()) if kw else {}
        err = np.abs(ang - serial)
        print(json.dumps({"variant": a.variant, "share": f"1/{a.n}", "chains": S * 6, "options": kw, "ms_per_step": dt * 1e3,
                          "max_abs_vs_serial": float(err.max()), "leg_frames_over_2e-5": int((err.max(2) > 2e-5).sum()),
                          "chunk_stats": {k: v for k, v in st.items() if v}}), flush=True)


if __name__ == "__main__":
    main()
