#!/bin/bash
# round-5 GPU call 39: the tile kernels with their store addresses recomputed after the K loop (no scratch; 256 -> 236 VGPRs in the FP64 128-tile form) against the build before, one box
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or syrk or potrf" 2>&1 | tail -2 || exit 1
{
for rep in 1 2; do for v in libcimrgp_prev.so libcimrgp.so; do echo "== $v"; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$v python3 tools/potrf_sweep.py --sizes 2048,4096,8192,32768 2>/dev/null | cut -c1-100; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$v python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 2>/dev/null | cut -c1-100; done; done
for rep in 1 2; do for v in libcimrgp_prev.so libcimrgp.so; do echo -n "bench $v: "; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done; done
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_prev.so python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1 | cut -c1-120
python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1 | cut -c1-120
} | tee gpurun_out/r05_tile_noscratch_ab.txt
