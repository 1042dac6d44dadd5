#!/usr/bin/env python3
"""Where ONE wavefront's time goes inside a pass of the generic-chain kernel (diagnostic build, needs a GPU).

    bash scripts/ab/build_variants.sh blk:"-DSEQIK_BLOCK_CYCLES=1"
    SEQIK_LIB=$PWD/build_ab/libseqik_blk.so python scripts/generic_block_cycles.py [--frames 6000] [--no-groups]

The reference-shaped call (LegInvKinGeneric on the shipped recording: one leg, 6000 frames) is one wavefront walking
one chain: its wall-clock IS the latency of the dependent operations of a pass.  The diagnostic build stamps the shader
clock (s_memtime) at the block boundaries of run_generic (csrc/seqik_generic.hpp) and charges the cycles to the block
that just ended.  Prints one JSON line: cycles per pass and per block, passes per frame."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from seqikpy_amd import _lib  # noqa: E402

BLOCKS = ["loop", "new_solve", "fd_jacobian", "scaling_gtol", "tr_step", "select_step", "trial_point_sincos_gather", "trial_chain_cost",
          "post_trial", "finished", "pipe_wait"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--no-groups", action="store_true")
    a = ap.parse_args()
    lib = _lib.load()
    if not hasattr(lib, "seqik_debug_block_cycles"):
        raise SystemExit("this library was not built with -DSEQIK_BLOCK_CYCLES=1 (set SEQIK_LIB)")
    z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    pose = np.stack([z["RF_pose"][:a.frames]])[None]
    p = [_lib.leg_params_from_arrays(z["RF_seg"], z["RF_bounds"], z["RF_seeds"])]
    n = 4 * (len(BLOCKS) + 1)
    buf = (ctypes.c_ulonglong * n)()
    _lib.solve_generic(pose[:, :, :50], p, lane_groups=not a.no_groups)
    lib.seqik_debug_block_cycles(None, 1)
    t0 = time.perf_counter()
    out = _lib.solve_generic(pose, p, want_diag=False, lane_groups=not a.no_groups)
    dt = time.perf_counter() - t0
    lib.seqik_debug_block_cycles(buf, 1)
    c = np.array(list(buf), dtype=np.float64).reshape(4, len(BLOCKS) + 1)[0]
    cyc, passes = c[:-1], c[-1]
    print(json.dumps({"case": f"anipose RF, {a.frames} frames, generic chain", "lane_groups": not a.no_groups, "wall_ms_with_stamps": dt * 1e3,
                      "passes": passes, "passes_per_frame": passes / a.frames, "cycles_per_pass": cyc.sum() / passes,
                      "cycles_per_pass_by_block": {b: round(float(v / passes), 1) for b, v in zip(BLOCKS, cyc) if v},
                      "share": {b: round(float(v / cyc.sum()), 4) for b, v in zip(BLOCKS, cyc) if v},
                      "angles_finite": bool(np.isfinite(out["angles"]).all())}))


if __name__ == "__main__":
    main()
