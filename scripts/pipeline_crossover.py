"""Serial walk: lane-per-chain kernel vs stage pipeline as the number of chains grows (64-frame sequences of the df3d
recording, device-resident, kernel time by HIP events)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib
z = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
legs = [str(l) for l in z["legs"]]
params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
T = 64
base = np.stack([np.stack([z[f"{l}_pose"][o:o + T] for l in legs]) for o in range(0, 900, 7)])  # 129 sequences
for S in (1, 4, 16, 43, 86, 171, 342, 683, 1366):
    pose = np.ascontiguousarray(base[np.arange(S) % base.shape[0]])
    d_pose = torch.from_numpy(pose).cuda()
    d_ang = torch.zeros((S, 6, T, 7), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    row = dict(chains=S * 6, frames=T)
    ref = None
    for name, pl in (("lane_per_chain", 1), ("stage_pipeline", 2)):
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang.data_ptr(), d_fk.data_ptr(), pipeline=pl)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row[name + "_ms"] = round(best, 3)
        if ref is None:
            ref = d_ang.clone()
        else:
            row["bit_identical"] = bool(torch.equal(ref, d_ang))
    print(json.dumps(row), flush=True)
