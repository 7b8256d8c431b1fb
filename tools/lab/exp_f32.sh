#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
for cfg in "CIMRGP_GEMM_PERS=256" "CIMRGP_GEMM_PERS=0" "CIMRGP_CHAIN_CUS=0" "CIMRGP_FAR_PAIR=100000" "CIMRGP_GEMM_PERS=0 CIMRGP_TAIL_BELOW=4864"; do
  echo "== $cfg"; env $cfg python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 2>/dev/null
done
