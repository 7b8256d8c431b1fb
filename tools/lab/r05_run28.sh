#!/bin/bash
# round-5 GPU call 28: which of this session's changes breaks the serialised (rocprofv3 --pmc) bench run
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
try() {
  echo "== $*"
  env CIMRGP_LIB_PATH=$T "$@" timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/dbg_pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/dbg.out 2> gpurun_out/dbg.err
  echo "rc=$?"; grep -v "^[EWI]2026" gpurun_out/dbg.err | tail -3; tail -1 gpurun_out/dbg.out | cut -c1-120
  rm -rf gpurun_out/dbg_pmc
}
try CIMRGP_POST_FINAL=0 CIMRGP_BENCH_FRONT=0
try CIMRGP_POST_FINAL=1 CIMRGP_BENCH_FRONT=0
try CIMRGP_POST_FINAL=0 CIMRGP_BENCH_FRONT=1
