# interleaved A/B of the one-query latency: the in-tree library against pram_amd/csrc/variants/libpram_hip_$1.so
for r in 1 2 3; do
  for t in tree $1; do
    if [ $t = tree ]; then lib=""; else lib=$PWD/pram_amd/csrc/variants/libpram_hip_$t.so; fi
    PRAM_HIP_LIB=$lib python bench.py --latency --cpu-queries 0 --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t $r: %.4f ms per query' % d['ms_per_step'])"
  done
done
