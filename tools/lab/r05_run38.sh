#!/bin/bash
# round-5 GPU call 38: long runs of the pipelined step (no watchdog, no failure, steady rate) and the smoke entry
mkdir -p gpurun_out
for r in 1 2; do python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline 2>gpurun_out/long.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['whole_step']['frac'])" || { tail -3 gpurun_out/long.err; exit 1; }; done
python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --nccl-world1 2>gpurun_out/long.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['reduce_selfcheck_max_abs_diff'], d['drained_step_ms'])" || { tail -3 gpurun_out/long.err; exit 1; }
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --dtype f32 2>gpurun_out/long.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32', d['value'], d['ms_per_step'])" || { tail -3 gpurun_out/long.err; exit 1; }
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
