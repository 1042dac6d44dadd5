"""Wall-clock of the reference-shaped calls (configs 1, 2, 4): one recording, few chains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib
def run(name, z, legs, sl=slice(None), **kw):
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"][sl] for l in legs])[None]
    _lib.solve_seq(pose[:, :, :4], params)  # warm-up (library load, allocator)
    t0 = time.perf_counter(); out = _lib.solve_seq(pose, params, want_fk=True, **kw); dt = time.perf_counter() - t0
    n = pose.shape[1] * pose.shape[2]
    print(f"{name}: {pose.shape[1]} legs x {pose.shape[2]} frames: {dt*1e3:.1f} ms wall  ({n/dt:.0f} leg-frames/s)")
    return out
za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz")); zd = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
run("config 1 (RF, 100 frames)", za, ["RF"], slice(0, 100))
run("config 2 (df3d, 6 legs, 1000 frames)", zd, [str(l) for l in zd["legs"]])
run("config 4 legs (anipose RF+LF, 6000 frames)", za, ["RF", "LF"])

from seqikpy_amd.frame_parallel import solve_frame_parallel
def run_fp(name, z, legs, chunk, halo):
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    serial = _lib.solve_seq(pose, params, want_fk=True)
    st = {}
    t0 = time.perf_counter(); out = solve_frame_parallel(pose, params, chunk=chunk, halo=halo, stats=st); dt = time.perf_counter() - t0
    n = pose.shape[1] * pose.shape[2]
    err = np.abs(out["angles"] - serial["angles"])
    print(f"{name} frame-parallel chunk={chunk} halo={halo}: {dt*1e3:.1f} ms wall ({n/dt:.0f} leg-frames/s), "
          f"max |d| vs serial {err.max():.2e}, frames > 1e-6: {(err.max(-1) > 1e-6).sum()}, {st}")
for chunk, halo in ((64, 16), (32, 16), (128, 16), (32, 8)):
    run_fp("config 2", zd, [str(l) for l in zd["legs"]], chunk, halo)
for chunk, halo in ((64, 16), (32, 16), (128, 16)):
    run_fp("config 4 legs", za, ["RF", "LF"], chunk, halo)
