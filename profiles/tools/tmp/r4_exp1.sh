#!/bin/bash
set -u
OUT=gpurun_out/r4a; mkdir -p $OUT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q 2>&1 | tail -25 > $OUT/tests.log; tail -3 $OUT/tests.log
python bench.py --cpu-queries 0 --alt off > $OUT/bench0.json 2> $OUT/bench0.err; cut -c1-400 $OUT/bench0.json
for v in 0 1 8 9; do PRAM_ATTN_VAR=$v PRAM_PROBE_SHAPES=16x2048,32x2048,8x4096 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/var=$v /"; done | tee $OUT/attn_var.txt
for v in 0 1; do PRAM_ATTN_VAR=$v python profiles/tools/x3_attn_sum.py 2>&1 | grep -v amdgpu | sed "s/^/var=$v /"; done | tee $OUT/attn_sum.txt
PRAM_ATTN_VAR=1 python bench.py --cpu-queries 0 --alt off --no-parity > $OUT/bench_var1.json 2> $OUT/bench_var1.err; cut -c1-300 $OUT/bench_var1.json
