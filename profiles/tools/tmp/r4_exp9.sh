#!/bin/bash
OUT=gpurun_out/r4g; mkdir -p $OUT
run() { python bench.py --latency --cpu-queries 0 --no-parity "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
echo "default(512/512): $(run)"
echo "chunk256 target1024: $(PRAM_BENCH_SPLIT_TARGET=1024 run --attn-chunk-keys 256)"
echo "chunk256 target512: $(PRAM_BENCH_SPLIT_TARGET=512 run --attn-chunk-keys 256)"
echo "chunk512 target1024: $(PRAM_BENCH_SPLIT_TARGET=1024 run --attn-chunk-keys 512)"
echo "chunk1024 target256: $(PRAM_BENCH_SPLIT_TARGET=256 run --attn-chunk-keys 1024)"
echo "chunk128 target2048: $(PRAM_BENCH_SPLIT_TARGET=2048 run --attn-chunk-keys 128)"
