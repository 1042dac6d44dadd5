#!/bin/bash
# the bench lines committed under profiles/ for a round: the driver's command, the same with --detail, and the no-flag run (100 steps)
set -o pipefail
R=${1:-r06}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_default_line.json 2> /dev/null && cp bench_detail.json gpurun_out/${R}_bench_default_detail.json || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 --detail > gpurun_out/${R}_bench_detail_line.json 2> /dev/null && cp bench_detail.json gpurun_out/${R}_bench_detail.json || exit 1
python bench.py --gpus 1 > gpurun_out/${R}_bench_k100_line.json 2> /dev/null || exit 1
cut -c1-420 gpurun_out/${R}_bench_default_line.json; echo; cut -c1-300 gpurun_out/${R}_bench_k100_line.json
