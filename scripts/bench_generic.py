"""Generic IK kernel: throughput of a full batch on windows of the shipped recording (device-resident), with the work
accounting that makes two builds comparable: a wavefront of the one-lane-per-chain instantiation lives as long as its slowest
lane, and which chain that is depends on the iteration path (another association order of the same arithmetic takes other
paths), so besides the wall-clock the script reports the passes the slowest wavefront made (from a diagnostics run of the same
build: nfev per frame) and the time per such pass.

    [SEQIK_LIB=build_ab/libseqik_x.so] python scripts/bench_generic.py          (needs a GPU)
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from seqikpy_amd import _lib  # noqa: E402

z = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
legs = ["RF", "LF"]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
rec = np.stack([z[f"{l}_pose"] for l in legs])          # (2, 6000, 5, 3), read ONCE (npz members are decompressed per access)
T = 32
lib = _lib.load()
arr = (_lib.SeqikLegParams * 2)(*params)
for S in (4096, 32768):
    offs = (np.arange(S) * 11) % (6000 - T)
    idx = offs[:, None] + np.arange(T)[None, :]
    pose = np.ascontiguousarray(rec[:, idx].transpose(1, 0, 2, 3, 4))     # (S, 2, T, 5, 3)
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((S, 2, T, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 2, T, 9, 3), dtype=torch.float64, device="cuda")
    d_st = torch.zeros((S, 2, T), dtype=torch.int32, device="cuda")
    d_nf = torch.zeros((S, 2, T), dtype=torch.int32, device="cuda")
    opt = _lib.SeqikOptions()

    def run(diag=False):
        rc = lib.seqik_solve_generic_device(d_pose.data_ptr(), S, 2, T, arr, d_ang.data_ptr(), d_fk.data_ptr(),
                                            d_st.data_ptr() if diag else None, d_nf.data_ptr() if diag else None,
                                            None, None, None, ctypes.byref(opt), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    run(diag=True)
    torch.cuda.synchronize()
    passes = (d_nf - 1 + (d_st == 1).int()).sum(2).cpu().numpy()          # (S, 2) passes per chain
    # leg-pure waves of 64 consecutive sequences (seqik_hip.hip chain_of_wave_lane)
    wave = np.stack([passes[:, l].reshape(-1, 64).max(1) for l in range(2)])
    print(json.dumps({"sequences": S, "legs": 2, "frames": T, "ms": best * 1e3, "M_leg_frames_per_s": S * 2 * T / best / 1e6,
                      "lane_passes_mean": float(passes.mean()), "slowest_wave_passes": int(wave.max()),
                      "mean_wave_passes": float(wave.mean()), "us_per_pass_of_the_slowest_wave": best * 1e6 / float(wave.max()),
                      "lib": os.path.basename(_lib.LIB_PATH)}), flush=True)
