#!/bin/bash
# round-5 GPU call 20: GPU tests with the front queue in the bench step and the staged call's third safety net
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05_bench_f.json; python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_f.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step'], d['parity_ok'], d['parity_rel_err_mean'], d['stage_ms'])"
