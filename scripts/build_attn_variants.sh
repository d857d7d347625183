#!/bin/bash
# Timing-only variants of the memory-read kernel for scripts/ab_attn.sh: one library per -D switch under experiments/ab/
# (results of these builds are wrong by construction; they only tell what each part of the tile loop costs).
# Usage: bash scripts/build_attn_variants.sh NAME=-DFLAG[,-DFLAG2] ...
cd "$(dirname "$0")/../rmem_ocu_amd/csrc" || exit 1
make -s || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=fast -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1"
mkdir -p ../../experiments/ab
rm -f ../../experiments/ab/*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=$(echo "${spec#*=}" | tr ',' ' ')
  hipcc $F $flags -c attention.hip -o /tmp/attn_$name.o 2>&1 | grep -v hip-link
  others=$(ls *.o | grep -v '^attention.o$')
  hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/attn_$name.o -o ../../experiments/ab/$name.so || exit 1
  echo "built experiments/ab/$name.so ($flags)"
done
