#!/bin/bash
# Round 3, experiment A: the persistent trailing-update kernel stand-alone and inside potrf.
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out; mkdir -p $OUT
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
L=$OUT/r03a_gemm.log; : > $L
for pers in 0 256 240 224; do
  for k in 256 512; do
    echo "== pers=$pers k=$k" >> $L
    CIMRGP_GEMM_PERS=$pers timeout -k 10 120 python3 tools/gemm_bench.py --m 7936,5888,4096 --k $k --reps 20 --check >> $L 2>&1 || { echo "FAILED pers=$pers k=$k" >> $L; exit 1; }
  done
done
L=$OUT/r03a_potrf.log; : > $L
for cfg in "0 0" "256 0" "256 16" "256 32" "256 48" "256 64"; do
  set -- $cfg
  echo "== pers=$1 chain_cus=$2" >> $L
  for n in 8192 16384; do
    CIMRGP_GEMM_PERS=$1 CIMRGP_CHAIN_CUS=$2 timeout -k 10 120 python3 tools/potrf_time.py $n 4 >> $L 2>&1 || { echo "FAILED $cfg" >> $L; exit 1; }
  done
done
unset CIMRGP_LIB_PATH
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/r03a_tests.log 2>&1 || { tail -40 $OUT/r03a_tests.log; echo TESTS FAILED; }
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-warmups 0 --cpu-repeats 1 > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err || { tail -20 $OUT/r03a_bench.err; exit 1; }
tail -3 $OUT/r03a_tests.log
cat $OUT/r03a_gemm.log
cat $OUT/r03a_potrf.log
