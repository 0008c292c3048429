#!/bin/bash
run() { python bench.py --kpts 4096 --n-class 513 --batch-per-gpu 8 --precision f16 --cpu-queries 0 --no-parity --alt off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
echo "F16_ACT=0: $(PRAM_F16_ACT=0 run)"
echo "F16_ACT=1: $(PRAM_F16_ACT=1 run)"
done
timeout 600 python -m pytest tests -m gpu -x -q -k "f16 or c5 or fp16 or half" 2>&1 | tail -3
