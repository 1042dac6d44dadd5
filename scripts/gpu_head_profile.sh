#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + HBM traffic counters of the head / antenna kernel at an
# HBM-bound size.  Usage: bash scripts/gpu_head_profile.sh TAG [frames]  -> gpurun_out/TAG_head_{stats,fetch,write}/
set -o pipefail
TAG=${1:-prof}
N=${2:-16000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/bench_head.py $N"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_head_stats" -- $CMD > "$OUT/${TAG}_head_stats.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/${TAG}_head_fetch" -- $CMD > "$OUT/${TAG}_head_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/${TAG}_head_write" -- $CMD > "$OUT/${TAG}_head_write.log" 2>&1 || exit 1
tail -1 "$OUT/${TAG}_head_stats.log"
