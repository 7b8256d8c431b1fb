#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_layer.py -x -q -k "batched or layer" 2>&1 | tail -2
for g in 0 1 0 1; do
  echo "== trsm_group=$g"
  CIMRGP_TRSM_GROUP=$g timeout -k 10 120 python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1
  CIMRGP_TRSM_GROUP=$g timeout -k 10 120 python3 tools/layer_time.py 64 4096 3 2>/dev/null | tail -1
  CIMRGP_TRSM_GROUP=$g timeout -k 10 120 python3 tools/layer_time.py 256 1024 5 2>/dev/null | tail -1
done
python3 tools/potrf_sweep.py --sizes 2048,4096,8192 2>/dev/null
