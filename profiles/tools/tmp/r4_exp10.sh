#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -8
