#!/bin/bash
cd "$(dirname "$0")/../.."
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_layer.py -x -q -k "gram or cross or layer or predict" 2>&1 | tail -2
python3 tools/gram_time.py 8192 2>/dev/null
python3 tools/gram_time.py 16384 2>/dev/null
python3 tools/gram_time.py 4096 2>/dev/null
