"""Generic IK kernel: throughput on replicated shipped data (device-resident)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch, ctypes
from seqikpy_amd import _lib
z = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
legs = ["RF", "LF"]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
T = 32
for S in (4096, 32768):
    offs = (np.arange(S) * 11) % (6000 - T)
    pose = np.stack([np.stack([z[f"{l}_pose"][o:o + T] for l in legs]) for o in offs])
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((S, 2, T, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 2, T, 9, 3), dtype=torch.float64, device="cuda")
    lib = _lib.load()
    arr = (_lib.SeqikLegParams * 2)(*params)
    opt = _lib.SeqikOptions()
    def run():
        rc = lib.seqik_solve_generic_device(d_pose.data_ptr(), S, 2, T, arr, d_ang.data_ptr(), d_fk.data_ptr(), None, None,
                                            None, None, None, ctypes.byref(opt), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"generic IK: S={S} x 2 legs x {T} frames: {dt*1e3:.1f} ms, {S*2*T/dt/1e6:.2f} M leg-frames/s")
