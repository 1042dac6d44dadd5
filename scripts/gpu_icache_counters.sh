#!/bin/bash
# Run ON THE GPU BOX (through gpurun): instruction-cache counters of the default bench workload (one PMC pass).
set -o pipefail
TAG=${1:-icache}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|SQC_INST\|IFETCH\|SQ_INST_CYCLES\|INST_LEVEL" | cut -c1-160 > "$OUT/${TAG}_available.txt"
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 $*"
timeout -k 10 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/${TAG}_pmc" -- $BENCH > "$OUT/${TAG}_pmc.log" 2>&1 || exit 1
echo done
