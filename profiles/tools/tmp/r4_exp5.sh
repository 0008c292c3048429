#!/bin/bash
set -u
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -k "resblock" 2>&1 | tail -15 | tee $OUT/tests.log
for a in 0 1 2 4 7; do PRAM_RB_ABLATE=$a timeout 300 python profiles/tools/resblock_probe.py 2>&1 | grep "B=" | sed "s/^/abl=$a /"; done | tee $OUT/resblock_probe.txt
