#!/bin/bash
# round-5 GPU call 29: forward update a wave per row
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2 || exit 1
python3 tools/potrs_time.py 8192 2 5 2>/dev/null | tail -1
python3 tools/potrs_time.py 16384 2 5 2>/dev/null | tail -1
python3 tools/potrs_time.py 4096 2 5 2>/dev/null | tail -1
