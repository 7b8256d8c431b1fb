#!/bin/bash
# round-5 GPU call 26: backward solve -- the updated value requested up front, no transpose in front of the fused calls' backward half
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 || exit 1
for r in 1 2; do python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['backward_solve_and_predict'], d['stage_ms']['potrf_with_carried_rows'])"; done
python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1 | cut -c1-400
python3 tools/potrs_time.py 8192 2 5 2>/dev/null | tail -1
