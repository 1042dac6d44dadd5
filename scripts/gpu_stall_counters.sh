#!/bin/bash
# Run ON THE GPU BOX (through gpurun): stall-side PMC passes of the default bench workload (wave cycles waiting,
# vector-memory instructions and their average time in flight, instruction fetches).  -> gpurun_out/TAG_{wait,vmem}/
set -o pipefail
TAG=${1:-stall}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 $*"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/${TAG}_wait" -- $BENCH > "$OUT/${TAG}_wait.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/${TAG}_vmem" -- $BENCH > "$OUT/${TAG}_vmem.log" 2>&1 || exit 1
echo done
