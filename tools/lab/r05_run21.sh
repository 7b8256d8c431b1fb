#!/bin/bash
# round-5 GPU call 21: FP32 factorisation (config 5) -- the persistent update with more units left to the chain, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo "== $*"; env CIMRGP_LIB_PATH=$T "$@" python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 2>/dev/null | cut -c1-120; }
{
one CIMRGP_NONE=0
one CIMRGP_GEMM_PERS_F32=1
one CIMRGP_GEMM_PERS_F32=1 CIMRGP_CHAIN_CUS=48
one CIMRGP_GEMM_PERS_F32=1 CIMRGP_CHAIN_CUS=64
one CIMRGP_GEMM_PERS_F32=1 CIMRGP_CHAIN_CUS=96
one CIMRGP_GEMM_PERS_F32=1 CIMRGP_CHAIN_CUS=64 CIMRGP_TAIL_BELOW=6144
one CIMRGP_TAIL_BELOW=6144
one CIMRGP_TAIL_BELOW=3584
one CIMRGP_NONE=0
} | tee gpurun_out/r05_f32_scan.txt
