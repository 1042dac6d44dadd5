#!/bin/bash
# Run ON THE GPU BOX (through gpurun): instruction-cache counters of the latency-bound launches (scripts/latency_kernels_run.py):
# the stage-pipeline kernels carry the four stage bodies in one kernel (~95 KB of code against a 64 KB instruction cache).
set -o pipefail
TAG=${1:-icache_lat}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/${TAG}_pmc" -- python3 $ROOT/scripts/latency_kernels_run.py --reps 1 > "$OUT/${TAG}_pmc.log" 2>&1 || exit 1
echo done
