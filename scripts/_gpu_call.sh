set -o pipefail
mkdir -p gpurun_out/r05c
timeout -k 10 300 ./scripts/microbench/head_split 16000000 > gpurun_out/r05c/head_split16.jsonl 2>&1; echo "split16 rc=$?"
cat gpurun_out/r05c/head_split16.jsonl
timeout -k 10 300 ./scripts/microbench/head_split 64000000 > gpurun_out/r05c/head_split64.jsonl 2>&1; echo "split64 rc=$?"
cat gpurun_out/r05c/head_split64.jsonl
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes_per_wave or single_launch or full_size or streams" > gpurun_out/r05c/gputests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r05c/gputests.log
# lone job of the full config-3 problem: chains per wavefront x lane pairs in the fused kernel
for W in 64 48 32 24; do
  for P in 1 0; do
    SEQIK_FUSED_PAIRS=$P timeout -k 10 200 python bench.py --steps 12 --warmup 3 --streams 1 --lanes-per-wave $W --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lone W=$W pairs=$P', round(b['ms_per_step'],3), 'ms', b['value'])"
  done
done | tee gpurun_out/r05c/lone_job_lanes.txt
for W in 32 24; do
  for P in 1 0; do
    SEQIK_FUSED_PAIRS=$P timeout -k 10 200 python bench.py --steps 20 --warmup 5 --streams 3 --lanes-per-wave $W --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3 streams W=$W pairs=$P', round(b['ms_per_step'],3), 'ms', b['value'])"
  done
done | tee -a gpurun_out/r05c/lone_job_lanes.txt
