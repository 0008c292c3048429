#!/bin/bash
# Interleaved A/B of library builds (profiles/tools/build_variants.py) on ONE box:  ab.sh REPS "CMD" TAG [TAG ...]
# runs CMD once per tag and repetition with PRAM_HIP_LIB pointing at pram_amd/csrc/variants/libpram_hip_TAG.so (TAG "tree" = the
# in-tree library), alternating the tags inside every repetition, and prints "TAG rep: <CMD's stdout>".
reps=$1; cmd=$2; shift 2
for r in $(seq 1 $reps); do
  for t in "$@"; do
    if [ $t = tree ]; then lib=""; else lib=$PWD/pram_amd/csrc/variants/libpram_hip_$t.so; fi
    PRAM_HIP_LIB=$lib bash -c "$cmd" 2>/dev/null | grep -v amdgpu | sed "s/^/$t $r: /"
  done
done
