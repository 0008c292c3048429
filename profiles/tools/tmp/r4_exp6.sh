#!/bin/bash
for a in 0 1 2 4 3 7; do PRAM_RB_ABLATE=$a timeout 300 python profiles/tools/resblock_probe.py 2>&1 | grep "B=16" | sed "s/^/abl=$a /"; done | tee gpurun_out/r4e_abl.txt
