#!/bin/bash
# Kernel-trace A/B of library builds inside the bench (one batch in flight): average duration of the step's top kernels per build.
#   bash profiles/tools/kt_ab.sh TAG [TAG ...]        (TAG "tree" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for t in "$@"; do
  if [ $t = tree ]; then lib=""; else lib=$ROOT/pram_amd/csrc/variants/libpram_hip_$t.so; fi
  rm -rf /tmp/kt_$t
  ( cd /tmp && export TMPDIR=/tmp && PRAM_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$t -o kt --output-format csv -- \
      python $ROOT/bench.py --steps 4 --warmup 2 --inflight 1 --cpu-queries 0 --no-parity --alt off > /tmp/kt_$t.log 2>&1 )
  f=$(find /tmp/kt_$t -name "*kernel_stats.csv" | head -1)
  echo "== $t  ($(grep -o '"value": [0-9.]*' /tmp/kt_$t.log | head -1))"
  python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:64]
    print(f"  {n:64s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
done
