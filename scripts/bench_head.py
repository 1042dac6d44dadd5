"""Head / antenna kernel (config 4 row): streaming rate on N frames, HIP-event timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch, json
from seqikpy_amd import _lib
z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_head.npz"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
reps = N // 6000
r = torch.from_numpy(z["R_head"]).cuda().repeat(reps, 1, 1)
l = torch.from_numpy(z["L_head"]).cuda().repeat(reps, 1, 1)
neck = torch.from_numpy(z["Neck"][0, 0].copy()).cuda()
n = r.shape[0]
out = torch.zeros((7, n), dtype=torch.float64, device="cuda")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = lib.seqik_head_angles_device(r.data_ptr(), l.data_ptr(), n, neck.data_ptr(), 0, float(z["rest_head_pitch"][0]),
                                      float(z["rest_antenna_pitch"][0]), 1, out.data_ptr(), st)
    assert rc == 0
for _ in range(15): run()  # the first launches after the fill kernels run ~10 % slower
K = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record()
for i in range(K):
    run(); ev[i + 1].record()
torch.cuda.synchronize()
each = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(K)])
ms = float(each.mean())
bytes_per_frame = 96 + 56
rate = lambda t: bytes_per_frame * n / t / 1e6
print(json.dumps({"kernel": "seqik_head_kernel", "frames": n, "launches": K, "ms": ms, "ms_min": float(each.min()),
                  "ms_median": float(np.median(each)), "ms_max": float(each.max()), "frames_per_s": n / ms * 1e3,
                  "algorithmic_GBps": rate(ms), "algorithmic_GBps_best_launch": rate(float(each.min())),
                  "hbm_peak_GBps": 8000, "frac": rate(ms) / 8000, "frac_best_launch": rate(float(each.min())) / 8000,
                  "ms_each": [round(float(v), 3) for v in each]}))
