#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "select or keypoint or sfd2 or extract or pipeline or keep_all or topk" 2>&1 | tail -4
for i in 1 2; do python bench.py --latency --cpu-queries 0 --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('latency', d['ms_per_step'], d['value'])"; done
python bench.py --cpu-queries 0 --alt off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'], d['parity'])"
