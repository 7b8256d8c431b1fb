#!/bin/bash
# SQ counters of the trailing update stand-alone (M = 7936, K = 256), old kernel and persistent one.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out; TAG=${1:-x}
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
for pers in 0 256; do
  export CIMRGP_GEMM_PERS=$pers
  i=0
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
              "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_VMEM" \
              "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    d=$OUT/${TAG}_pmc_p${pers}_$i
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python3 tools/gemm_bench.py --m 7936 --k 256 --reps 5 > $d.log 2>&1 || { echo "pass failed: $pass"; tail -5 $d.log; }
  done
  python3 tools/pmc_summary.py $OUT/${TAG}_pmc_p${pers}_*/ > $OUT/${TAG}_pmc_pers${pers}.json 2> $OUT/${TAG}_pmc_pers${pers}.err || tail -3 $OUT/${TAG}_pmc_pers${pers}.err
  rm -rf $OUT/${TAG}_pmc_p${pers}_*/
done
python3 - <<PY
import json
for p in (0, 256):
    try:
        d = json.load(open("gpurun_out/${TAG}_pmc_pers%d.json" % p))
    except Exception as e:
        print("no summary", p, e); continue
    for k, v in d.items():
        if "gemm" in k:
            print(p, k, json.dumps(v))
PY
