#!/bin/bash
# Everything profiles/ holds for one round, taken on ONE box from ONE tree (run it through gpurun; the outputs land in
# gpurun_out/<tag>/ and are copied into profiles/ by profiles/tools/install_round.py in the build container).
#   bash profiles/tools/collect_round.sh r03
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
b() { name=$1; shift; python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "bench_$name rc=$? $(cut -c1-160 "$OUT/bench_$name.json")"; }
# PMC passes first (one counter set per run, never combined with other tracing) and their summary on THIS box: bench.py prints
# roofline.traffic only while profiles/pmc_attention.json carries the hash of the attention source it runs
bash profiles/tools/pmc_collect.sh "$OUT/pmc" --alt off > "$OUT/pmc.log" 2>&1; tail -4 "$OUT/pmc.log"
python profiles/tools/pmc_summarize.py "$OUT/pmc" "$TAG" > "$OUT/pmc_summarize.log" 2>&1; tail -1 "$OUT/pmc_summarize.log" | cut -c1-200
b default
b adagml --matcher adagml --cpu-queries 0 --alt off
b f32 --precision f32 --cpu-queries 0 --alt off
b inflight1 --inflight 1 --cpu-queries 0 --alt off
b latency --latency --cpu-queries 0
b b1_inflight3 --batch-per-gpu 1 --steps 100 --warmup 30 --cpu-queries 0 --alt off
b secondary_512x1024 --match-kpts 512 --ref-kpts 1024 --cpu-queries 0 --alt off
b c4 --kpts 4096 --n-class 161 --batch-per-gpu 8 --cpu-queries 0 --alt off
b c5_f16 --kpts 4096 --n-class 513 --batch-per-gpu 8 --precision f16 --cpu-queries 0 --no-parity --alt off
b c5_x3 --kpts 4096 --n-class 513 --batch-per-gpu 8 --cpu-queries 0 --alt off
# kernel trace of the default command, one batch in flight (kernel names in order)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
    python "$ROOT/bench.py" --steps 4 --warmup 2 --inflight 1 --cpu-queries 0 --no-parity --alt off > "$OUT/kt.log" 2>&1 ); echo "kernel trace rc=$?"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt_latency" -o kt --output-format csv -- \
    python "$ROOT/bench.py" --latency --steps 40 --warmup 10 --graph off --cpu-queries 0 --no-parity --alt off > "$OUT/kt_latency.log" 2>&1 ); echo "latency kernel trace rc=$?"
# micro-probes
for ck in 512 1024 2048; do
  for m in 0 1 2; do PRAM_ATTN_CHUNK_KEYS=$ck PRAM_ATTN_MODE=$m python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/chunk_keys=$ck /"; done
  PRAM_ATTN_CHUNK_KEYS=$ck python profiles/tools/x3_attn_probe.py 2>&1 | grep -v amdgpu | sed "s/^/chunk_keys=$ck /"
done > "$OUT/x3_attention_chunks.txt"
python profiles/tools/x3_mlp_tail_probe.py 2>&1 | grep -v amdgpu > "$OUT/x3_mlp_tail_probe.txt"
python profiles/tools/x3_gemm_probe.py 2>&1 | grep -v amdgpu > "$OUT/x3_gemm_probe.txt"
ls "$OUT"
