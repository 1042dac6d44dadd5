"""bench.py's generic_batches leg on its own (static launch against the chain queue on batches of generic chains): for
profiling under rocprofv3 (scripts/gpu_misc_profile.sh) and for quick checks.  Prints the leg's JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import numpy as np  # noqa: E402

import bench_extras as bench  # noqa: E402

za = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
print(json.dumps(bench.generic_batches(za)))
