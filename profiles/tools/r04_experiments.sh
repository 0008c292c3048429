#!/bin/bash
# The one-off measurements of round 4 that are quoted in DESIGN.md / profiles/README.md and are not part of collect_round.sh.
# Run through gpurun from the repo root; every block prints what it measured.
#   bash profiles/tools/r04_experiments.sh [batch|latency|c5act|attnsweep|resblock]
set -u
what=${1:-all}
bench() { python bench.py --cpu-queries 0 --no-parity --alt off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('power',{}).get('socket_w_mean'), d.get('power',{}).get('sclk_mhz_mean'))"; }
if [ $what = batch ] || [ $what = all ]; then      # profiles/r04_batch_sweep.txt: batch size x steps in flight
  echo "b16 inflight3: $(bench)"; echo "b8 inflight3: $(bench --batch-per-gpu 8 --steps 40)"; echo "b8 inflight6: $(bench --batch-per-gpu 8 --inflight 6 --steps 40)"
  echo "b32 inflight2: $(bench --batch-per-gpu 32 --inflight 2 --steps 10)"; echo "b16 inflight2: $(bench --inflight 2)"
fi
if [ $what = latency ] || [ $what = all ]; then    # attention key chunks / split target in the one-query mode
  lat() { python bench.py --latency --cpu-queries 0 --no-parity "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
  echo "default (512 keys, 512 workgroups): $(lat)"; echo "256 keys, 1024: $(PRAM_BENCH_SPLIT_TARGET=1024 lat --attn-chunk-keys 256)"
  echo "1024 keys, 256: $(PRAM_BENCH_SPLIT_TARGET=256 lat --attn-chunk-keys 1024)"; echo "128 keys, 2048: $(PRAM_BENCH_SPLIT_TARGET=2048 lat --attn-chunk-keys 128)"
fi
if [ $what = c5act ] || [ $what = all ]; then      # fp16 intermediates of the C5 path
  c5() { python bench.py --kpts 4096 --n-class 513 --batch-per-gpu 8 --precision f16 --cpu-queries 0 --no-parity --alt off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
  echo "PRAM_F16_ACT=0: $(PRAM_F16_ACT=0 c5)"; echo "PRAM_F16_ACT=1: $(PRAM_F16_ACT=1 c5)"
fi
if [ $what = attnsweep ] || [ $what = all ]; then  # profiles/r04_x3_attention_batch_sweep.txt
  PRAM_PROBE_SHAPES=8x2048,16x2048,24x2048,32x2048,48x2048,64x2048,16x1024,32x1024,64x1024,4x4096,8x4096,16x4096 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu
fi
if [ $what = resblock ] || [ $what = all ]; then   # profiles/r04_resblock_probe.txt
  for a in 0 1 2 4 7; do PRAM_RB_ABLATE=$a python profiles/experiments/resblock_probe_r04.py  # (needs the round-4 tree: the fused ResBlock kernel left the library in round 5)
  2>&1 | grep "B=" | sed "s/^/abl=$a /"; done
fi
