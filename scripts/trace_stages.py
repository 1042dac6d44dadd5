"""Per-stage kernel durations of the serial walk on config 4 (for rocprofv3 --kernel-trace): what the stage pipeline
can gain at best = sum of the stage kernels / the longest of them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import latency_configs as lc
za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
name, pose, params, ref, ok = lc.case("config 4", za, ["RF", "LF"])
from seqikpy_amd import _lib
for kw in (dict(staged=1, pipeline=1), dict(pipeline=1), dict(pipeline=2)):
    out, dt = lc.timed(pose, params, reps=2, **kw)
    print(kw, round(dt * 1e3, 2), "ms", flush=True)
