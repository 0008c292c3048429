#!/bin/bash
# Round 6's collection, in parts (each fits one gpurun call): bash profiles/tools/collect_round5.sh A|B|C   -> gpurun_out/r06/
set -u
PART=${1:-A}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
b() { name=$1; shift; python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "bench_$name rc=$? $(cut -c1-140 "$OUT/bench_$name.json")"; }
if [ $PART = A ]; then
  # PMC passes (one counter set per run, never combined with other tracing) and their summary on THIS box; then the default line
  bash profiles/tools/pmc_collect.sh "$OUT/pmc" --alt off > "$OUT/pmc.log" 2>&1; tail -4 "$OUT/pmc.log"
  python profiles/tools/pmc_summarize.py "$OUT/pmc" r06 > "$OUT/pmc_summarize.log" 2>&1; tail -1 "$OUT/pmc_summarize.log" | cut -c1-200
  cp profiles/r06_pmc_summary.md profiles/pmc_attention.json "$OUT/" 2>/dev/null
  b default
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
      python "$ROOT/bench.py" --steps 4 --warmup 2 --inflight 1 --cpu-queries 0 --no-parity --alt off > "$OUT/kt.log" 2>&1 ); echo "kernel trace rc=$?"
fi
if [ $PART = B ]; then
  b default_final
  b adagml --matcher adagml --cpu-queries 0 --alt off
  b f32 --precision f32 --cpu-queries 0 --alt off
  b inflight1 --inflight 1 --cpu-queries 0 --alt off
  b latency --latency --cpu-queries 0
  b b1_inflight3 --batch-per-gpu 1 --steps 100 --warmup 30 --cpu-queries 0 --alt off
  b secondary_512x1024 --match-kpts 512 --ref-kpts 1024 --cpu-queries 0 --alt off
  b c4 --kpts 4096 --n-class 161 --batch-per-gpu 8 --cpu-queries 0 --alt off
  b c5_f16 --kpts 4096 --n-class 513 --batch-per-gpu 8 --precision f16 --cpu-queries 0 --no-parity --alt off
  b c5_x3 --kpts 4096 --n-class 513 --batch-per-gpu 8 --cpu-queries 0 --alt off
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt_latency" -o kt --output-format csv -- \
      python "$ROOT/bench.py" --latency --steps 40 --warmup 10 --graph off --cpu-queries 0 --no-parity --alt off > "$OUT/kt_latency.log" 2>&1 ); echo "latency kernel trace rc=$?"
fi
if [ $PART = C ]; then
  python profiles/tools/x3_gemm_quick.py 2>&1 | grep -v amdgpu > "$OUT/x3_gemm_quick.txt"
  python profiles/tools/x3_mlp_tail_probe.py 2>&1 | grep -v amdgpu > "$OUT/x3_mlp_tail_probe.txt"
  python profiles/tools/x3_gemm_probe.py 2>&1 | grep -v amdgpu > "$OUT/x3_gemm_probe.txt"
  PRAM_PROBE_SHAPES=8x2048,16x2048,24x2048,32x2048,48x2048,64x2048,16x1024,32x1024,4x4096,8x4096,16x4096,1x2048,2x2048 python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu > "$OUT/x3_attention_batch_sweep.txt"
  for ck in 512 1024 2048; do
    for m in 1 2; do PRAM_ATTN_CHUNK_KEYS=$ck PRAM_ATTN_MODE=$m python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/chunk_keys=$ck /"; done
  done > "$OUT/x3_attention_chunks.txt"
fi
ls "$OUT"
