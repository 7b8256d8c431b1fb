#!/bin/bash
# round-5 GPU call 40: fewer units for the chain while the update is much the longer path (CIMRGP_CHAIN_CUS_WIDE above CIMRGP_CHAIN_WIDE_ABOVE trailing columns), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo "== $*"; env CIMRGP_LIB_PATH=$T "$@" python3 tools/potrf_sweep.py --sizes 8192,12288,16384 2>/dev/null | cut -c1-100; }
{
one CIMRGP_NONE=0
one CIMRGP_CHAIN_CUS_WIDE=24
one CIMRGP_CHAIN_CUS_WIDE=16
one CIMRGP_CHAIN_CUS_WIDE=24 CIMRGP_CHAIN_WIDE_ABOVE=6144
one CIMRGP_CHAIN_CUS_WIDE=16 CIMRGP_CHAIN_WIDE_ABOVE=7424
one CIMRGP_CHAIN_CUS_WIDE=8 CIMRGP_CHAIN_WIDE_ABOVE=7424
one CIMRGP_NONE=0
} | tee gpurun_out/r05_chain_wide.txt
