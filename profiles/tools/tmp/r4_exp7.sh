#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_c5 -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --kpts 4096 --n-class 513 --batch-per-gpu 8 --precision f16 --cpu-queries 0 --no-parity --alt off --steps 4 --warmup 2 --inflight 1 > $OUT/kt_c5.log 2>&1
echo rc=$?; tail -2 $OUT/kt_c5.log | cut -c1-300
find $OUT/kt_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c5_f16_kernel_stats.csv
rm -rf $OUT/kt_c5
head -40 $OUT/c5_f16_kernel_stats.csv | cut -c1-160
