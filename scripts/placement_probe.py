"""How does the time of a serial walk depend on the number of one-lane waves and on the workgroup size (= how the
dispatcher places waves on SIMDs)?  Lane-per-chain kernel, W = 1, 64-frame df3d cuts."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib
z = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
legs = [str(l) for l in z["legs"]]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
T = 64
base = np.stack([np.stack([z[f"{l}_pose"][o:o + T] for l in legs]) for o in range(0, 936, 3)])
for S in (8, 16, 32, 43, 64, 86, 128, 171, 256, 342):
    pose = np.ascontiguousarray(base[np.arange(S) % base.shape[0]])
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((S, 6, T, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    row = dict(chains=S * 6)
    for W in (1, 2):
        for block in (64, 128, 256):
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=1,
                                      lanes_per_wave=W, block_size=block)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row[f"W{W}_block{block}"] = round(best, 3)
    print(json.dumps(row), flush=True)
