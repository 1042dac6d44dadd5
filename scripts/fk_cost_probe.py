#!/usr/bin/env python3
"""What do the forward-kinematics rows cost the single-launch kernel?  The same batch with and without the FK output (the FK adds and
stores sit in the sparse end-of-frame block of stages 2-4): a direct measurement of how much of the step is in that block's
instructions.  One JSON line.  (needs a GPU)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sequential-inverse-kinematics_amd")]
import numpy as np, torch
import bench_support as bs
from seqikpy_amd import _lib, synthetic

def main():
    S, T, L = 15625, 64, 6
    out = {}
    for variant in ("iid", "smooth"):
        legs, body, pose, params = bs.make_workload(S, T, variant, synthetic.SEED_BASE)
        d_pose = torch.from_numpy(np.ascontiguousarray(pose.transpose(0, 1, 3, 2, 4))).cuda()
        layout = _lib.planar_layout(T)
        streams = [torch.cuda.Stream() for _ in range(4)]
        angs = [torch.zeros((S, L, 7, T), dtype=torch.float64, device="cuda") for _ in streams]
        fks = [torch.zeros((S, L, T, 9, 3), dtype=torch.float64, device="cuda") for _ in streams]
        torch.cuda.synchronize()
        res = {}
        for name, with_fk in (("with_fk", True), ("without_fk", False), ("with_fk_again", True)):
            def step(i):
                k = i % 4
                _lib.solve_seq_device(d_pose.data_ptr(), S, L, T, params, angs[k].data_ptr(), fks[k].data_ptr() if with_fk else 0,
                                      stream=streams[k].cuda_stream, layout=layout)
            for i in range(8):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(60):
                step(i)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 60 * 1e3
        out[variant] = res
    print(json.dumps(out))

if __name__ == "__main__":
    main()
