#!/bin/bash
# round-5 GPU call 22: FP32 potrf n = 16384, kernel totals with the tile-per-workgroup update (default) and the persistent one
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
for v in 0 1; do
  CIMRGP_LIB_PATH=$T CIMRGP_GEMM_PERS_F32=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32tr_$v -- python3 tools/potrf_once.py 16384 3 0 f32 > /dev/null 2> gpurun_out/f32tr.err || { tail -5 gpurun_out/f32tr.err; exit 1; }
  echo "== CIMRGP_GEMM_PERS_F32=$v"
  f=$(find gpurun_out/f32tr_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    name = re.sub(r"\(anonymous namespace\)::|cimrgp::|void ", "", r["Name"])[:70]
    print("%-72s calls %5s total_ms %9.3f avg_us %9.1f" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  python3 tools/timeline.py gpurun_out/f32tr_$v > gpurun_out/r05_f32_timeline_pers$v.txt 2>/dev/null
  rm -rf gpurun_out/f32tr_$v
done | tee gpurun_out/r05_f32_kernel_totals.txt
