#!/bin/bash
set -u
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -k "resblock" 2>&1 | tail -15 | tee $OUT/tests.log
timeout 300 python profiles/tools/resblock_probe.py 2>&1 | grep -v amdgpu | tee $OUT/resblock_probe.txt
