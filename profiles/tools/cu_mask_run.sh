profiles/bin/cu_mask_probe
for r in 1 2; do
for cfg in ":3" "eo:2" "eo:4" "halves:2" "halves:4" "w:2" "w:4" ":2" ":4"; do
  m=${cfg%%:*}; n=${cfg##*:}
  PRAM_BENCH_CU_MASK=$m python bench.py --no-parity --alt off --cpu-queries 0 --inflight $n --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask=%-7s inflight=%s: %.1f q/s  %.3f ms/step  %s W' % ('$m' or 'none', '$n', d['value'], d['ms_per_step'], d.get('power',{}).get('socket_w_mean')))"
done; done
