#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/conv_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $set | cut -d' ' -f1)
  PYTHONPATH=$ROOT timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$name -o pmc --output-format csv -- python $ROOT/profiles/tools/x3_conv_probe.py > $OUT/$name.log 2>&1
  echo "$name rc=$?"
done
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "conv" not in k or "at::" in k: continue
            k = k.replace("(anonymous namespace)::","")[:60] + " grid=" + row.get("Grid_Size","?")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v)/1e6, 2) for c, v in cs.items()}, "n =", len(next(iter(cs.values()))))
PY
