"""The per-rank share of the fixed config-3 problem at N = 2, 4, 8 (S / N sequences x 6 legs x 64 frames, planar, 7 angles +
FK) with MANY steps in flight: kernel family (stage pipeline / lane per chain) x streams.  One JSON line per row.
GPU_MAX_HW_QUEUES must be set before the runtime starts: the script re-runs itself per queue count."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for n, frames in ((8, 125000), (4, 250000), (2, 500000)):
        for pipe in (0, 1):
            for streams in (3, 8, 16, 24):
                env = dict(os.environ, GPU_MAX_HW_QUEUES=str(max(8, streams)))
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", str(frames), "--streams", str(streams),
                       "--stage-pipeline", str(pipe), "--steps", str(max(40, 6 * streams)), "--warmup", str(streams),
                       "--no-extras", "--no-cpu-baseline"]
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
                if not line:
                    print(json.dumps({"share_of": n, "pipeline": pipe, "streams": streams, "error": r.stderr[-300:]}), flush=True)
                    continue
                b = json.loads(line[-1])
                print(json.dumps({"share_of": n, "chains": b["config"]["chains_per_gpu"], "stage_pipeline": pipe, "streams": streams,
                                  "hw_queues": env["GPU_MAX_HW_QUEUES"], "ms_per_step": round(b["ms_per_step"], 3)}), flush=True)
