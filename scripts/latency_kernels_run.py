#!/usr/bin/env python3
"""The latency-bound launches of the reference-shaped calls, one after the other, for profiling under rocprofv3
(scripts/gpu_latency_profile.sh; summarised by scripts/latency_floor.py):

  A  config 4's serial walk (anipose RF + LF x 6000 frames) as the library runs it: seqik_pipe_kernel, lane pairs on
  B  the same with lane pairs off (SeqikOptions.reserved[3] = 3)
  C  the same walk stage by stage (one seqik_stage_kernel<S> launch per stage, one wavefront per chain): the per-stage
     instruction counts of ONE wavefront without the pipeline's waiting loops -- the instruction stream B's waves issue
  D  the generic chain on the shipped recording (RF x 6000 frames): seqik_generic_kernel<diag = 0, grouped = 1>
  E  the 1/8 share of the fixed config-3 problem (1 953 sequences x 6 legs x 64 frames, synthetic iid, planar layout) as a
     LONE job, as the library runs it (automatic: stage pipeline, 64 chains per workgroup)
  F  the same share stage by stage (one seqik_stage_kernel<S> launch per stage): the per-stage instruction streams of its
     wavefronts without the pipeline's waiting loops

Every launch runs `--reps` times; prints one JSON line with the host wall-clock of each (best of reps)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from seqikpy_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    legs = ["RF", "LF"]
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    n = pose.shape[2]
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((1, 2, n, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((1, 2, n, 9, 3), dtype=torch.float64, device="cuda")

    def timed(fn):
        best = 1e9
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    out = {"frames": n, "legs": legs, "reps": a.reps}
    out["A_pipe_pairs_ms"] = timed(lambda: _lib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=2))
    ref = d_ang.clone()
    out["B_pipe_no_pairs_ms"] = timed(lambda: _lib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=3))
    out["B_equals_A_bitwise"] = bool(torch.equal(ref, d_ang))
    out["C_stage_kernels_ms"] = timed(lambda: _lib.solve_seq_device(d_pose.data_ptr(), 1, 2, n, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=1, staged=1))
    out["C_equals_A_bitwise"] = bool(torch.equal(ref, d_ang))
    gp = [params[0]]
    g_pose = np.ascontiguousarray(pose[:, :1])
    out["D_generic_ms"] = timed(lambda: _lib.solve_generic(g_pose, gp, want_fk=True))
    # ---- E / F: the lone 1/8 share of BASELINE config 3 (what a rank of an 8-GPU strong-scaling run solves per step) ----
    from seqikpy_amd import data, synthetic, utils
    legs6 = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs6)
    p6 = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs6]
    S8, T = 15625 // 8, 64
    sp = synthetic.synthetic_pose(S8, T, legs6, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="iid",
                                  seed=synthetic.SEED_BASE)
    s_pose = torch.from_numpy(np.ascontiguousarray(sp.transpose(0, 1, 3, 2, 4))).cuda()      # planar [S][L][5][T][3]
    s_ang = torch.zeros((S8, 6, 7, T), dtype=torch.float64, device="cuda")
    s_fk = torch.zeros((S8, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    lay = _lib.planar_layout(T)
    out["share_sequences"], out["share_chains"] = S8, S8 * 6
    out["E_share_piped_ms"] = timed(lambda: _lib.solve_seq_device(s_pose.data_ptr(), S8, 6, T, p6, s_ang.data_ptr(), s_fk.data_ptr(), layout=lay))
    ref8 = s_ang.clone()
    out["F_share_stage_kernels_ms"] = timed(lambda: _lib.solve_seq_device(s_pose.data_ptr(), S8, 6, T, p6, s_ang.data_ptr(), s_fk.data_ptr(),
                                                                          layout=lay, pipeline=1, staged=1))
    out["F_equals_E_bitwise"] = bool(torch.equal(ref8, s_ang))
    _lib.check_faults()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
