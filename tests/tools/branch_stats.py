"""Branch statistics of the trust-region step on the benchmark data (oracle counters; CPU only).
    python tests/tools/branch_stats.py [iid|smooth]"""
import sys, ctypes, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'sequential-inverse-kinematics_amd'))
from oracle import c_oracle
from seqikpy_amd import data, synthetic, utils
import subprocess
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "stats"])
c_oracle._SO = os.path.join(ROOT, "oracle", "_build", "libseqik_oracle_stats.so")   # the build with counters
L = c_oracle.lib()
legs = data.LEGS
body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
variant = sys.argv[1] if len(sys.argv) > 1 else "iid"
pose = synthetic.synthetic_pose(64, 64, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION, variant=variant, seed=synthetic.SEED_BASE)
L.oracle_stats_reset.argtypes=[ctypes.c_int]; L.oracle_stats_reset(1)
nfev = np.zeros(4)
for li, leg in enumerate(legs):
    seg, b, seeds = c_oracle.leg_params(leg, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION)
    for s in range(64):
        r = c_oracle.seq_leg(pose[s, li], seg, b, seeds)
        nfev += r["nfev"].sum(0)
out = (ctypes.c_longlong * 48)()
L.oracle_stats_get(out)
st = np.array(list(out)).reshape(2, 24)
n_solves = 64*64*6
print("mean nfev per solve, stages 1-4:", nfev / n_solves)
for k, name in enumerate(["stage 1 (deficient)", "stages 2-3"]):
    c = st[k]
    print(name, "TR calls", c[0], "per solve", c[0]/n_solves/(1 if k==0 else 2))
    print("  GN inside region %.3f  shortcut %.3f  loop %.3f   mean loop iterations %.2f" % (c[1]/c[0], c[2]/c[0], c[3]/c[0], c[4]/max(c[3],1)))
    print("  loop iteration histogram", (c[9:19]/max(c[3],1)).round(3))
    print("  select_step calls", c[5], "reflective share %.4f" % (c[6]/max(c[5],1)))
    p = c[3]/c[0]
    h = c[9:19]/max(c[3],1)
    # expected wave-level iterations: max over the loop lanes of a 64-lane wave (lanes independent)
    cdf = np.cumsum(h)
    import math
    exp_max = 0.0
    for it in range(1, 11):
        # P(max <= it) = prod over lanes (1 - p + p*cdf[it-1])
        pm = (1 - p + p*cdf[it-1])**64
        pm_prev = (1 - p + p*(cdf[it-2] if it >= 2 else 0.0))**64
        exp_max += it * (pm - pm_prev)
    print("  expected loop iterations per wave-pass (64 lanes): %.2f ; P(any lane reflective) %.3f" % (exp_max, 1-(1-c[6]/max(c[5],1))**64))
