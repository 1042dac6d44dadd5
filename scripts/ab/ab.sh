#!/bin/bash
# usage: ab.sh <outdir> <lib1> [lib2 ...] [-- extra bench args]   -- bench each library variant twice, interleaved
out=$1; shift
libs=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; libs+=("$1"); shift; done
mkdir -p $out
for rep in 1 2; do
for lib in "${libs[@]}"; do
  name=$(basename $lib .so)
  SEQIK_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --steps 60 "${extra[@]}" > $out/${name}_$rep.json 2> $out/${name}_$rep.err || echo "FAILED $name"
  python -c "
import json,sys
d=json.loads(open('$out/${name}_$rep.json').read().strip().splitlines()[-1]); print('$name', $rep, round(d['ms_per_step'],3), 'ms', '%.4g'%d['value'])"
done
done
