"""Single-launch time of mid-size problems (200 / 1000 / 4000 sequences x 6 legs x 64 frames) as a function of the
chains per wavefront -- the measurement behind pick_lanes_per_wave() in csrc/seqik_hip.hip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib, data, synthetic, utils
legs = data.LEGS
body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
T = 64
for S in (200, 1000, 4000):
    pose = synthetic.synthetic_pose(S, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="smooth", seed=5)
    d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
    d_ang = torch.zeros((S, 6, 7, T), dtype=torch.float64, device="cuda")
    d_fk = torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda")
    lay = _lib.planar_layout(T)
    st = torch.cuda.current_stream().cuda_stream
    auto = 64 if S * 6 >= 1024 else max(1, (S * 6 + 255) // 256)   # pick_lanes_per_wave()
    for W in sorted(set([0, 1, 2, 3, 6, 12, 24, 48, 64])):
        if W > 64: continue
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, d_ang.data_ptr(), d_fk.data_ptr(), stream=st, layout=lay, lanes_per_wave=W)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"S={S} chains={S*6} W={W or 'auto(%d)' % auto}: {best*1e3:.2f} ms")
