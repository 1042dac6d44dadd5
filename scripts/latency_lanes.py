"""Frame chunks: chains per wavefront (SeqikOptions.reserved[0]) vs wall-clock for the reference-shaped calls."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import latency_configs as lc
za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz")); zd = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
cases = [lc.case("config 1", za, ["RF"], slice(0, 100)), lc.case("config 2", zd, [str(l) for l in zd["legs"]]),
         lc.case("config 4", za, ["RF", "LF"])]
for name, pose, params, ref, ok in cases:
    for c, h in ((4, 8), (8, 8), (16, 8)):
        row = {}
        for w in (1, 2, 3, 4, 6, 8, 16, 32, 64):
            _, dt = lc.timed(pose, params, frame_chunk=c, frame_halo=h, lanes_per_wave=w)
            row[w] = round(dt * 1e3, 3)
        print(json.dumps(dict(case=name, chunk=c, halo=h, chunks=pose.shape[1] * -(-pose.shape[2] // c), ms_by_lanes_per_wave=row)), flush=True)
