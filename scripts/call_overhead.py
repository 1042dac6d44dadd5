import os, sys, time, json
ROOT="/root/repo" if os.path.exists("/root/repo/scripts") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import latency_configs as lc
za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
for n in (1, 8, 16, 100):
    name, pose, params, ref, ok = lc.case("c", za, ["RF"], slice(0, n))
    for kw in (dict(pipeline=1), dict(pipeline=2), dict(frame_chunk=4, frame_halo=4) if n >= 16 else None):
        if kw is None: continue
        out, dt = lc.timed(pose, params, reps=20, **kw)
        print(n, kw, round(dt*1e3, 3), "ms", flush=True)
