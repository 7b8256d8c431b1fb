#!/bin/bash
# round-5 GPU call 23: the backward solve's update one cache line wide per workgroup
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2 || exit 1
python3 tools/potrs_time.py 2>/dev/null | tail -6
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['backward_solve_and_predict'], d['stage_ms']['potrf_with_carried_rows'])"
