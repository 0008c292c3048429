#!/bin/bash
set -u
OUT=gpurun_out/r4b; mkdir -p $OUT
python -X faulthandler -m pytest tests/test_gpu_round4.py -x -v 2>&1 | grep -v "^  File\|dist-packages" | head -150 > $OUT/tests4.log; tail -30 $OUT/tests4.log
python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -15 > $OUT/tests3.log; tail -5 $OUT/tests3.log
