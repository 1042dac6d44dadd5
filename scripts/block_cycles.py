#!/usr/bin/env python3
"""Where a wavefront's time goes inside the solver's pass loop (diagnostic build, needs a GPU).

    bash scripts/ab/build_variants.sh blk:"-DSEQIK_BLOCK_CYCLES=1"
    SEQIK_LIB=$PWD/build_ab/libseqik_blk.so python scripts/block_cycles.py [--variant iid|smooth] [--frames 1000000]

The diagnostic build stamps the shader clock (s_memtime) at every block boundary of run_stage (csrc/seqik_core.hpp,
SEQIK_BLK) and charges the cycles to the block that just ended: WAVE time, whatever the number of active lanes.  One
launch of the benchmark batch at a time (no other launch in flight).  Prints one JSON line: per stage the share of wave
cycles per block, the cycles per pass, and the totals.  The stamps cost ~10 % and serialise a little; shares, not
absolute times, are the result."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from seqikpy_amd import _lib, data, synthetic, utils  # noqa: E402

# conditional parts of a pass whose executions the diagnostic build counts (seqik_core.hpp CNT_*)
COUNTED = ["new_solve", "feasible_slow_path", "start_evaluation", "body", "first_pass_radius", "trust_region_and_trial", "reflective",
           "accept", "finished", "fd_step_slow_path", "root_search_iteration", "root_search_evaluation", "gauss_newton_step_inside_radius", "body_with_fewer_than_16_lanes", "body_with_fewer_than_32_lanes"]
BLOCKS = ["loop", "new_solve", "fd_jacobian", "scaling_gtol", "tr_step", "in_bounds", "reflective", "trial_eval", "post_trial",
          "finished", "pipe_wait"]


def recording(a):
    lib = _lib.load()
    if not hasattr(lib, "seqik_debug_block_cycles"):
        raise SystemExit("this library was not built with -DSEQIK_BLOCK_CYCLES=1 (set SEQIK_LIB)")
    z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    legs = ["RF", "LF"]
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    n = pose.shape[2]
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((1, 2, n, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((1, 2, n, 9, 3), dtype=torch.float64, device="cuda")
    buf = (ctypes.c_ulonglong * (4 * (len(BLOCKS) + 1)))()
    ebuf = (ctypes.c_ulonglong * (4 * 2 * len(COUNTED)))()

    def run():
        _lib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=a.pipeline)
        torch.cuda.synchronize()
    run()
    lib.seqik_debug_block_cycles(None, 1)
    lib.seqik_debug_block_entries(None, 1)
    run()
    lib.seqik_debug_block_cycles(buf, 1)
    lib.seqik_debug_block_entries(ebuf, 1)
    c = np.array(list(buf), dtype=np.float64).reshape(4, len(BLOCKS) + 1)
    e = np.array(list(ebuf), dtype=np.float64).reshape(4, 2, len(COUNTED))
    out = {"case": "anipose RF + LF x 6000 frames, serial walk", "kernel": "stage pipeline" if a.pipeline >= 2 else "lane per chain",
           "frames": n, "stages": {}}
    for st in range(4):
        cyc, passes = c[st, :-1], max(c[st, -1], 1)
        out["stages"][str(st + 1)] = {"lane0_passes_both_legs": passes, "wave_cycles_both_legs": float(cyc.sum()),
                                      "cycles_per_frame_and_leg": float(cyc.sum() / (2 * n)),
                                      "cycles_per_frame_by_block": {b: round(float(v / (2 * n)), 1) for b, v in zip(BLOCKS, cyc) if v},
                                      "entries_per_frame": {nm: round(float(e[st, 0, i] / (2 * n)), 3) for i, nm in enumerate(COUNTED) if e[st, 0, i]}}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="iid")
    ap.add_argument("--frames", type=int, default=1_000_000)
    ap.add_argument("--replicate", action="store_true",
                    help="every sequence is a copy of sequence 0: the 64 lanes of a wavefront then take the same branches "
                         "(no divergence), which separates what divergence costs from what the code costs")
    ap.add_argument("--desync", action="store_true", help="every sequence = sequence 0 rotated in time by (index mod 64) frames")
    ap.add_argument("--no-fk", action="store_true", help="do not ask for the forward kinematics (no FK stores)")
    ap.add_argument("--staged", action="store_true", help="one launch per stage instead of the single launch")
    ap.add_argument("--pipeline", type=int, default=1, help="1 = lane-per-chain fused kernel (the benchmark's), 2 = stage pipeline")
    ap.add_argument("--recording", action="store_true",
                    help="the shipped anipose recording (RF + LF x 6000 frames) walked serially instead of the synthetic batch: "
                         "with --pipeline 2 this is config 4's default path, one workgroup of four stage wavefronts per leg")
    a = ap.parse_args()
    if a.recording:
        return recording(a)
    lib = _lib.load()
    if not hasattr(lib, "seqik_debug_block_cycles"):
        raise SystemExit("this library was not built with -DSEQIK_BLOCK_CYCLES=1 (set SEQIK_LIB)")
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    T = 64
    S = a.frames // T
    pose = synthetic.synthetic_pose(S, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant=a.variant,
                                    seed=synthetic.SEED_BASE)
    if a.replicate:
        pose[:] = pose[:1]
    if a.desync:   # the same frames in every lane, rotated in time: same branches statistically, frame boundaries out of step
        base = pose[:1].copy()
        for s_ in range(S):
            pose[s_] = np.roll(base[0], s_ % 64, axis=1)
    params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_ang = torch.zeros((S, 6, 7, T), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    n = 4 * (len(BLOCKS) + 1)
    buf = (ctypes.c_ulonglong * n)()

    def run():
        _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang.data_ptr(), 0 if a.no_fk else d_fk.data_ptr(),
                              layout=_lib.planar_layout(T), pipeline=a.pipeline, staged=int(a.staged))
        torch.cuda.synchronize()
    run()
    lib.seqik_debug_block_cycles(None, 1)
    have_entries = hasattr(lib, "seqik_debug_block_entries")
    ebuf = (ctypes.c_ulonglong * (4 * 2 * len(COUNTED)))()
    if have_entries:
        lib.seqik_debug_block_entries(None, 1)
    run()
    lib.seqik_debug_block_cycles(buf, 1)
    if have_entries:
        lib.seqik_debug_block_entries(ebuf, 1)
    c = np.array(list(buf), dtype=np.float64).reshape(4, len(BLOCKS) + 1)
    total = c[:, :-1].sum()
    out = {"variant": a.variant, "fk": not a.no_fk, "staged": a.staged, "replicated": a.replicate, "leg_frames": S * 6 * T, "kernel": "fused lane per chain" if a.pipeline == 1 else "stage pipeline",
           "total_wave_cycles": total, "stage_share_of_total": (c[:, :-1].sum(1) / total).round(4).tolist(), "stages": {}}
    for st in range(4):
        cyc, passes = c[st, :-1], c[st, -1]
        out["stages"][str(st + 1)] = {"wave_passes": passes, "cycles_per_pass": cyc.sum() / max(passes, 1),
                                      "share": {b: round(float(v / cyc.sum()), 4) for b, v in zip(BLOCKS, cyc) if v}}
    if have_entries:
        e = np.array(list(ebuf), dtype=np.float64).reshape(4, 2, len(COUNTED))
        for st in range(4):
            passes = max(c[st, -1], 1)
            out["stages"][str(st + 1)]["entries_per_pass"] = {n: round(float(e[st, 0, i] / passes), 4) for i, n in enumerate(COUNTED)}
            out["stages"][str(st + 1)]["active_lanes_per_entry"] = {n: round(float(e[st, 1, i] / max(e[st, 0, i], 1)), 2)
                                                                    for i, n in enumerate(COUNTED)}
    allc = c[:, :-1].sum(0)
    out["all_stages_share"] = {b: round(float(v / total), 4) for b, v in zip(BLOCKS, allc) if v}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
