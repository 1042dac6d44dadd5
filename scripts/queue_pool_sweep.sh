#!/bin/bash
# Run ON THE GPU BOX: the fused kernel's chain queue (SeqikOptions.reserved[0] = chains per wavefront) against the plain launch,
# (pool, steps in flight) pairs that put about the same number of wavefronts in flight.
# Usage: bash scripts/queue_pool_sweep.sh OUT.jsonl [variant] [steps] ["pool streams" ...]
OUT=${1:-gpurun_out/queue_pool_sweep.jsonl}
VARIANT=${2:-iid}
STEPS=${3:-20}
shift 3 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
: > "$OUT"
if [ $# -eq 0 ]; then set -- "0 3" "128 3" "128 6" "256 6" "256 12" "512 12" "512 20" "1024 20"; fi
for cfg in "$@"; do
  set -- $cfg
  python3 $ROOT/bench.py --steps $STEPS --warmup 5 --variant $VARIANT --no-extras --no-cpu-baseline --lanes-per-wave $1 --streams $2 --stage-pipeline 1 \
     --detail-path /tmp/d.json 2> /tmp/err.log | python3 -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'pool': $1, 'streams': $2, 'variant': '$VARIANT', 'steps': b['steps'], 'ms_per_step': b['ms_per_step'], 'value': b['value'], 'avg_launch_ms': b['roofline']['avg_launch_ms']}))" >> "$OUT" || { tail -5 /tmp/err.log; exit 1; }
  tail -1 "$OUT"
done
