#!/usr/bin/env python3
"""One recording sharded by frame over the ranks (seqikpy_amd.frame_sharding) with the real library:

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/frame_shard_demo.py [--backend gloo]

Every rank solves its slab of the shipped 6000-frame RF + LF recording on its GPU (LOCAL_RANK modulo the number of
GPUs, so several ranks can share one card for a dry run), boundaries are verified / repaired, the angles are
gathered; rank 0 compares with the serial solve and prints one JSON line."""
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
import torch.distributed as dist  # noqa: E402

from seqikpy_amd import _lib, frame_sharding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--chunk", type=int, default=32)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    legs = ["RF", "LF"]
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    frame_sharding.solve_frame_sharded(pose[:, :, :256], params, chunk=args.chunk, device=dev)   # warm-up
    dist.barrier()
    st = {}
    t0 = time.perf_counter()
    out = frame_sharding.solve_frame_sharded(pose, params, chunk=args.chunk, device=dev, stats=st)
    dt = time.perf_counter() - t0
    if rank == 0:
        serial = _lib.solve_seq(pose, params, want_fk=True, device=dev)
        err = np.abs(out["angles"] - serial["angles"]).max(-1)[0]          # (legs, frames)
        keep = np.ones(6000, bool)
        keep[280:310] = False                                               # the LF singular episode amplifies 1e-6
        print(json.dumps({"world": world, "backend": args.backend, "seconds": dt, "slab_rank0": st["slab"],
                          "boundary_rounds": st["boundary_rounds"], "resume_calls": st["resume_calls"],
                          "max_abs_vs_serial_RF": float(err[0].max()), "max_abs_vs_serial_LF_outside_episode": float(err[1][keep].max()),
                          "fk_max_abs_vs_serial_RF": float(np.abs(out["fk"][0, 0] - serial["fk"][0, 0]).max())}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
