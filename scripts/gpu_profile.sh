#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of the default bench
# workload.  Usage: bash scripts/gpu_profile.sh TAG [extra bench.py args, e.g. --staged]
#   -> gpurun_out/TAG_{stats,fetch,write,sq,f64}/ (+ .log)
# Summaries for profiles/ are cut from these by scripts/summarize_profile.py TAG rNN.
set -o pipefail
TAG=${1:-prof}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $*"   # --steps / --warmup of the command the driver runs
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- $BENCH > "$OUT/${TAG}_stats.log" 2>&1 || exit 1
# counters in their own runs (no tracing options), one pass each
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/${TAG}_fetch" -- $BENCH > "$OUT/${TAG}_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/${TAG}_write" -- $BENCH > "$OUT/${TAG}_write.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/${TAG}_sq" -- $BENCH > "$OUT/${TAG}_sq.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/${TAG}_f64" -- $BENCH > "$OUT/${TAG}_f64.log" 2>&1 || exit 1
grep -h '^{"metric"' "$OUT/${TAG}_stats.log" | tail -1
