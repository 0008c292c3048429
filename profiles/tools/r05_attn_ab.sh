#!/bin/bash
# round 5, experiment 1: the VALU diet of attention_x3_pipe_kernel — variants built by build_variants.py, A/B interleaved on one box.
#   bash profiles/tools/r05_attn_ab.sh "base scalar mix ..." [kernel reps] [bench reps]
set -u
tags=${1:-"base scalar mix swap lazy lazy8"}; kreps=${2:-4}; breps=${3:-0}
export PRAM_PROBE_SHAPES=16x2048,32x2048,8x4096
echo "## bit checksums (profiles/tools/x3_attn_sum.py)"
bash profiles/tools/ab.sh 1 "python profiles/tools/x3_attn_sum.py | tr '\n' ' '; echo" $tags
echo "## kernel timing (profiles/tools/x3_attn_probe.py)"
bash profiles/tools/ab.sh $kreps "python profiles/tools/x3_attn_probe.py | awk '{printf \"%s %s us | \", \$4 \$5, \$6} END {print \"\"}'" $tags
if [ $breps -gt 0 ]; then
  echo "## bench (16 queries per step, 20 steps, no parity gate / cpu baseline)"
  bash profiles/tools/ab.sh $breps "python bench.py --cpu-queries 0 --no-parity --alt off | python -c \"import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d.get('power',{}).get('socket_w_mean'), d.get('power',{}).get('sclk_mhz_mean'))\"" ${4:-$tags}
fi
