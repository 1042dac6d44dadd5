"""Wall-clock of the reference-shaped calls (BASELINE configs 1, 2, 4: one recording, few chains) through the
host-buffer entry point (transfers included): serial walk vs frame chunks (SeqikOptions.frame_chunk), each with
max |d theta| vs the serial walk and vs the fixture's reference angles."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sequential-inverse-kinematics_amd"))
import numpy as np
import torch  # noqa: F401  (HIP runtime)
from seqikpy_amd import _lib

LF_WINDOW = (284, 302)  # tests/conftest.py::LF_DEGENERATE


def case(name, z, legs, sl=slice(None)):
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"][sl] for l in legs])[None]
    ref = np.stack([z[f"{l}_angles"][sl] for l in legs])[None]
    ok = np.ones(pose.shape[:3], bool)
    for i, l in enumerate(legs):
        if l == "LF" and "anipose" in name:
            ok[0, i, LF_WINDOW[0]:LF_WINDOW[1]] = False
    return name, pose, params, ref, ok


def timed(pose, params, reps=5, **kw):
    _lib.solve_seq(pose, params, want_fk=True, **kw)  # warm-up: arena, workspace, leg table
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = _lib.solve_seq(pose, params, want_fk=True, **kw)
        best = min(best, time.perf_counter() - t0)
    return out, best


def main():
    za = np.load(os.path.join(ROOT, "tests/golden/anipose_shipped.npz"))
    zd = np.load(os.path.join(ROOT, "tests/golden/df3d_1000.npz"))
    cases = [case("config 1 (anipose RF, 100 frames)", za, ["RF"], slice(0, 100)),
             case("config 2 (df3d, 6 legs x 1000 frames)", zd, [str(l) for l in zd["legs"]]),
             case("config 4 legs (anipose RF + LF, 6000 frames)", za, ["RF", "LF"])]
    # pipeline: 1 = lane-per-chain kernels, 2 = stage pipeline, 3 = stage pipeline without lane pairs, absent = the
    # library's choice (stage pipeline with lane pairs at these sizes)
    variants = [dict(pipeline=1), dict(pipeline=3), dict(pipeline=2), dict(), dict(frame_chunk=-1, pipeline=1),
                dict(frame_chunk=-1, pipeline=3), dict(frame_chunk=-1, pipeline=2), dict(frame_chunk=-1)]
    if os.environ.get("LATENCY_QUICK"):  # A/B builds: the serial walk and the automatic chunks only
        variants = [dict(pipeline=3), dict(pipeline=2), dict(frame_chunk=-1)]
    else:
        for c, h in ((4, 4), (4, 8), (8, 8), (16, 8), (32, 8)):
            for pl in (1, 2):
                variants.append(dict(frame_chunk=c, frame_halo=h, pipeline=pl))
    rows = []
    for name, pose, params, ref, ok in cases:
        serial = None
        for kw in variants:
            out, dt = timed(pose, params, **kw)
            if serial is None:
                serial = out
            d_ser = np.abs(out["angles"] - serial["angles"])
            d_ref = np.abs(out["angles"] - ref)
            row = dict(case=name, options=kw or "defaults of the C ABI (serial walk)", ms=round(dt * 1e3, 3),
                       leg_frames_per_s=round(pose.shape[1] * pose.shape[2] / dt),
                       max_abs_vs_serial_outside_LF_window=float(d_ser[ok].max()),
                       max_abs_vs_serial=float(d_ser.max()),
                       leg_frames_over_1e4_vs_serial=int((d_ser.max(-1) > 1e-4).sum()),
                       max_abs_vs_reference_outside_LF_window=float(d_ref[ok].max()),
                       leg_frames_over_1e4_vs_reference=int((d_ref.max(-1) > 1e-4).sum()),
                       chunk_stats={k: v for k, v in out["chunk_stats"].items() if v})
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
