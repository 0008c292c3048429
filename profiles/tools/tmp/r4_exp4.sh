#!/bin/bash
set -u
OUT=gpurun_out/r4d; mkdir -p $OUT
PRAM_PROBE_SHAPES=8x2048,16x2048,24x2048,32x2048,48x2048,64x2048,16x1024,32x1024,64x1024,4x4096,8x4096,16x4096 timeout 300 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | tee $OUT/attn_batch_sweep.txt
PRAM_ATTN_WAVES=4 PRAM_PROBE_SHAPES=8x2048,16x2048,24x2048,32x2048,48x2048,64x2048 timeout 300 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/waves4 /" | tee -a $OUT/attn_batch_sweep.txt
