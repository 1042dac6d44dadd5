#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes (own runs, no tracing options) of the
# latency-bound launches (scripts/latency_kernels_run.py).  Usage: bash scripts/gpu_latency_profile.sh TAG
#   -> gpurun_out/TAG_lat_{stats,sq,f64}/ (+ .log); summarised by scripts/latency_floor.py TAG rNN
set -o pipefail
TAG=${1:-lat}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RUN="python3 $ROOT/scripts/latency_kernels_run.py --reps 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_lat_stats" -- $RUN > "$OUT/${TAG}_lat_stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/${TAG}_lat_sq" -- $RUN > "$OUT/${TAG}_lat_sq.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/${TAG}_lat_f64" -- $RUN > "$OUT/${TAG}_lat_f64.log" 2>&1 || exit 1
grep -h '^{"frames"' "$OUT/${TAG}_lat_stats.log" | tail -1
