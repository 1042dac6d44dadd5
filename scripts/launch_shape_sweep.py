"""Serial walk of S x 6 chains x 64 frames: kernel (lane per chain / stage pipeline) x chains per wavefront (group),
device-resident, kernel time by HIP events (best of 3).  --synthetic: the benchmark's i.i.d. workload instead of cuts
of the df3d recording."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib, data, synthetic, utils
ap = argparse.ArgumentParser()
ap.add_argument("--synthetic", action="store_true")
ap.add_argument("--seqs", type=int, nargs="+", default=[171, 342, 683, 1366, 2732, 5464, 15625])
ap.add_argument("--lanes", type=int, nargs="+", default=[0, 4, 8, 16, 32, 64])
ap.add_argument("--streams", type=int, default=1)
args = ap.parse_args()
T = 64
if args.synthetic:
    legs = data.LEGS
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    params = [_lib.make_leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]
    base = synthetic.synthetic_pose(max(args.seqs), T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant="iid")
else:
    z = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    base = np.stack([np.stack([z[f"{l}_pose"][o:o + T] for l in legs]) for o in range(0, 936, 3)])
streams = [torch.cuda.Stream() for _ in range(args.streams)]
for S in args.seqs:
    pose = np.ascontiguousarray(base[np.arange(S) % base.shape[0]])
    d_pose = torch.from_numpy(pose).cuda()
    bufs = [(torch.zeros((S, 6, T, 7), dtype=torch.float64, device="cuda"), torch.zeros((S, 6, T, 9, 3), dtype=torch.float64, device="cuda"))
            for _ in streams]
    row = dict(chains=S * 6, frames=T, streams=args.streams, data="synthetic iid" if args.synthetic else "df3d cuts", ms={})
    ref = None
    for name, pl in (("lane", 1), ("pipe", 2)):
        for W in args.lanes:
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k, st in enumerate(streams):
                    st.wait_stream(torch.cuda.current_stream())
                    _lib.solve_seq_device(d_pose.data_ptr(), S, 6, T, params, bufs[k][0].data_ptr(), bufs[k][1].data_ptr(),
                                          pipeline=pl, lanes_per_wave=W, stream=st.cuda_stream)
                for st in streams:
                    torch.cuda.current_stream().wait_stream(st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row["ms"][f"{name}_W{W}"] = round(best / args.streams, 3)
            if ref is None:
                ref = bufs[0][0].clone()
            elif not torch.equal(ref, bufs[0][0]):
                row["MISMATCH"] = f"{name}_W{W}"
    print(json.dumps(row), flush=True)
