#!/bin/bash
set -u
OUT=gpurun_out/r4c; mkdir -p $OUT
for v in 0 2; do PRAM_ATTN_VAR=$v timeout 300 python profiles/tools/x3_attn_sum.py 2>&1 | grep -v amdgpu | sed "s/^/var=$v /"; done | tee $OUT/attn_sum.txt
for v in 0 2 10 0 2; do PRAM_ATTN_VAR=$v PRAM_PROBE_SHAPES=16x2048,32x2048,8x4096 timeout 300 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/var=$v /"; done | tee $OUT/attn_var.txt
