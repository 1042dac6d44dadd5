"""One chunked call per reference-shaped config (for rocprofv3 --kernel-trace): which kernel takes how long."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import latency_configs as lc
za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz")); zd = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
for c in (lc.case("config 1", za, ["RF"], slice(0, 100)), lc.case("config 2", zd, [str(l) for l in zd["legs"]]),
          lc.case("config 4", za, ["RF", "LF"])):
    out, dt = lc.timed(c[1], c[2], reps=3, frame_chunk=-1)
    print(c[0], round(dt * 1e3, 3), "ms", out["chunk_stats"], flush=True)
