#!/bin/bash
# Run ON THE GPU BOX: samples the shader clock / power while the default benchmark workload runs
# (the VALU-issue roofline in DESIGN.md assumes the peak clock; a power-limited clock raises the real fraction).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|mclk" | head -6
  echo "--"
  sleep 1
done > gpurun_out/clock_samples.txt
wait $BP
cat gpurun_out/clock_samples.txt | head -40
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('%.3e %.2f'%(d['value'], d['ms_per_step']))"
