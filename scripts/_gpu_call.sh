set -o pipefail
mkdir -p gpurun_out/r05m
for FR in 125000 250000 500000; do
  timeout -k 10 300 python bench.py --frames $FR --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'frames': $FR, 'steps': 20, 'ms_per_step': round(b['ms_per_step'],3), 'cal': [(c['streams'], c['stage_pipeline'], c['latency_kernel_steps'], round(c['ms_per_step'],3)) for c in b['config']['depth_calibration']['candidates']], 'chosen': b['config']['depth_calibration']['chosen']}))"
done | tee gpurun_out/r05m/k20_depths_head.jsonl
