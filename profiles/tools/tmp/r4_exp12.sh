#!/bin/bash
run() { python bench.py --cpu-queries 0 --no-parity --alt off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('power',{}).get('socket_w_mean'), d.get('power',{}).get('sclk_mhz_mean'))"; }
echo "b16 inflight3: $(run)"
echo "b8 inflight3: $(run --batch-per-gpu 8 --steps 40)"
echo "b8 inflight6: $(run --batch-per-gpu 8 --inflight 6 --steps 40)"
echo "b4 inflight6: $(run --batch-per-gpu 4 --inflight 6 --steps 80)"
echo "b32 inflight2: $(run --batch-per-gpu 32 --inflight 2 --steps 10)"
echo "b32 inflight3: $(run --batch-per-gpu 32 --inflight 3 --steps 10)"
echo "b16 inflight2: $(run --inflight 2)"
