#!/bin/bash
# Builds library variants for A/B measurements into build_ab/ (git-ignored; they travel to the GPU box with the snapshot):
#   usage: bash scripts/ab/build_variants.sh name1:"-DFLAG=1 ..." name2:"..."      -> build_ab/libseqik_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $ROOT/build_ab
cd $ROOT/sequential-inverse-kinematics_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $flags -fPIC -shared -std=c++17 -o $ROOT/build_ab/libseqik_$name.so \
      seqik_hip.hip seqik_head.hip seqik_stream.hip seqik_align.hip seqik_peer.hip &
done
wait
ls -la $ROOT/build_ab
