#!/bin/bash
# round-5 GPU call 11: "panel final" as a posted device word + a gate on the update's queue (CIMRGP_POST_FINAL=1) against the event (0), same box, alternating
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "potrf" 2>&1 | tail -2 || exit 1
for rep in 1 2; do for pf in 0 1; do echo "== CIMRGP_POST_FINAL=$pf"; CIMRGP_LIB_PATH=$T CIMRGP_POST_FINAL=$pf python3 tools/potrf_sweep.py --sizes 8192,12288,16384 2>/dev/null; done; done | tee gpurun_out/r05_post_final_ab.txt
for rep in 1 2; do for pf in 0 1; do echo "== bench, CIMRGP_POST_FINAL=$pf"; CIMRGP_LIB_PATH=$T CIMRGP_POST_FINAL=$pf python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; done; done | tee -a gpurun_out/r05_post_final_ab.txt
