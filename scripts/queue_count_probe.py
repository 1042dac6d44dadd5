"""How many hardware queues (GPU_MAX_HW_QUEUES, set by the caller BEFORE this process starts) a process can ask for before
its long-running kernels pay for it: the one-chain generic call (one wavefront, 1.67 s) and the 20-deep step pipeline of a 1/8
share, which wants a queue per stream.  One JSON line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np, torch
from seqikpy_amd import _lib
z = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
p = [_lib.leg_params_from_arrays(z["RF_seg"], z["RF_bounds"], z["RF_seeds"])]
pose = np.ascontiguousarray(z["RF_pose"])[None, None]
torch.cuda.init()
if "--touch-streams" in sys.argv:
    # torch hands out streams from a pool of 32 per device; HIP maps them onto at most GPU_MAX_HW_QUEUES hardware queues.  Touch
    # them all, so that the process HOLDS as many hardware queues as the setting allows before the timed kernels run.
    ss = [torch.cuda.Stream() for _ in range(32)]
    for st in ss:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
_lib.solve_generic(pose[:, :, :50], p)
best = 1e9
for _ in range(2):
    t0 = time.perf_counter(); _lib.solve_generic(pose, p); best = min(best, time.perf_counter() - t0)
out = {"queues": os.environ.get("GPU_MAX_HW_QUEUES"), "streams_touched": 32 if "--touch-streams" in sys.argv else 0,
       "generic_6000_frames_s": round(best, 4)}
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "125000", "--steps", "20", "--warmup", "5", "--streams", "20",
                    "--stage-pipeline", "1", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
b = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
out["share8_depth20_ms_per_step"] = round(b["ms_per_step"], 3)
print(json.dumps(out), flush=True)
