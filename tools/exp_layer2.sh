#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
for h in 100000 8; do
  echo "== batch_halves_min=$h"
  CIMRGP_BATCH_HALVES=$h timeout -k 10 120 python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1
  CIMRGP_BATCH_HALVES=$h timeout -k 10 120 python3 tools/layer_time.py 64 4096 3 2>/dev/null | tail -1
  CIMRGP_BATCH_HALVES=$h timeout -k 10 120 python3 tools/layer_time.py 256 1024 5 2>/dev/null | tail -1
done
