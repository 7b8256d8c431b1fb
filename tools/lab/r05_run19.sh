#!/bin/bash
# round-5 GPU call 19: the next block's front end on the chain's queue, beside the previous factorisation's tail (CIMRGP_BENCH_FRONT=1), one box
mkdir -p gpurun_out
one() { echo -n "$* : "; env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['stage_ms']['potrf_alone'])"; }
{
for rep in 1 2 3; do
one CIMRGP_BENCH_FRONT=0
one CIMRGP_BENCH_FRONT=1
done
} | tee gpurun_out/r05_front_queue.txt
CIMRGP_BENCH_FRONT=1 python3 bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('parity', d.get('parity_ok'), d.get('parity_rel_err_mean'), d.get('parity_rel_err_var'), d['value'])" | tee -a gpurun_out/r05_front_queue.txt
