#!/bin/bash
for i in 1 2; do PRAM_PROBE_SHAPES=16x2048,32x2048,8x4096 timeout 300 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu; done
timeout 300 python profiles/tools/x3_attn_sum.py 2>&1 | grep -v amdgpu
