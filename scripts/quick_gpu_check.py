"""Dev script: smoke + timing of the stage kernels on replicated df3d data (device-resident buffers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np
import torch
import __graft_entry__ as g
g.smoke()
from seqikpy_amd import _lib
print("lib", _lib.LIB_PATH)
z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
legs = [str(l) for l in z["legs"]]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ss = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8192, 32768]
base = np.stack([z[f"{l}_pose"] for l in legs])  # (6, 1000, 5, 3)
for S in Ss:
    offs = (np.arange(S) * 7) % (1000 - T)
    pose = np.stack([base[:, o:o + T] for o in offs])  # (S, 6, T, 5, 3)
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((S, 6, T, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for bs in [64, 256]:
        for stages in [(1, 4), (1, 1), (2, 2), (3, 3), (4, 4)]:
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.time()
                _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang.data_ptr(), d_fk.data_ptr(),
                                      first_stage=stages[0], last_stage=stages[1], stream=stream, block_size=bs)
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
            print(f"S={S} T={T} block={bs} stages={stages}: {best*1e3:.2f} ms  {S*6*T/best/1e6:.2f} M leg-frames/s")
    out = _lib.solve_seq(pose[:2], params, want_fk=True)
    assert np.array_equal(out["angles"], d_ang[:2].cpu().numpy())
    assert np.array_equal(out["fk"], d_fk[:2].cpu().numpy())
print("ok")
