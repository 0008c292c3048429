#!/bin/bash
# rocprofv3 PMC passes for one bench step, one counter set per run (never combined with sys/hip/hsa tracing).
#   bash profiles/tools/pmc_collect.sh [out_dir] [extra bench.py args...]
# then:  python profiles/tools/pmc_summarize.py <out_dir> <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/pmc}; shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$OUT/$name" -o pmc --output-format csv -- \
      python "$ROOT/bench.py" --steps 1 --warmup 1 --inflight 1 --cpu-queries 0 --no-parity "$@" > "$OUT/$name.log" 2>&1
  echo "$name rc=$? $(ls $OUT/$name 2>/dev/null | tr '\n' ' ')"
done
