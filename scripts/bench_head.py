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
for _ in range(3): run()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    run(); ev[i + 1].record()
torch.cuda.synchronize()
ms = np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(10)])
bytes_per_frame = 96 + 56
print(json.dumps({"kernel": "seqik_head_kernel", "frames": n, "ms": ms, "frames_per_s": n / ms * 1e3,
                  "algorithmic_GBps": bytes_per_frame * n / ms / 1e6, "hbm_peak_GBps": 8000,
                  "frac": bytes_per_frame * n / ms / 1e6 / 8000}))
