#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of (a) the 1/8 share of the fixed config-3 problem in the
# driver's command form (what ONE rank of an 8-GPU strong-scaling run does), (b) batches of generic chains, static launch and
# chain queue.  Usage: bash scripts/gpu_misc_profile.sh TAG  -> gpurun_out/TAG_share8_stats/, TAG_genbatch_stats/ (+ .log)
set -o pipefail
TAG=${1:-misc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_share8_stats" -- python3 $ROOT/bench.py --frames 125000 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/${TAG}_share8_stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_genbatch_stats" -- python3 $ROOT/scripts/generic_batches_run.py > "$OUT/${TAG}_genbatch_stats.log" 2>&1 || exit 1
grep -h '^{"metric"' "$OUT/${TAG}_share8_stats.log" | tail -1 | cut -c1-300
