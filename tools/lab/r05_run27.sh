#!/bin/bash
# round-5 GPU call 27: L2 traffic of the persistent update stand-alone (M = 7936, K = 256) with the C stream non-temporal (n1 both, n2 loads, n3 stores)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
L=$PWD/cimrgp_amd
for v in "" _n1 _n2 _n3; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $c | cut -d' ' -f1)
    CIMRGP_LIB_PATH=$L/libcimrgp_tuning$v.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/nt_pmc${v}_$name -- python3 tools/gemm_bench.py --m 7936 --k 256 --reps 3 > /dev/null 2> gpurun_out/nt_pmc.err || { tail -5 gpurun_out/nt_pmc.err; exit 1; }
  done
  echo "== counters, variant ${v:-base}"
  python3 tools/pmc_summary.py gpurun_out/nt_pmc${v}_FETCH_SIZE/ gpurun_out/nt_pmc${v}_WRITE_SIZE/ gpurun_out/nt_pmc${v}_TCC_HIT_sum/ > gpurun_out/nt_pmc${v}.json
  python3 - gpurun_out/nt_pmc${v}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "pers" in k:
        out = {c: v[c]["mean"] for c in v if isinstance(v[c], dict) and "mean" in v[c]}
        out.update({c: v[c] for c in v if not isinstance(v[c], dict)})
        print(k, json.dumps(out))
PY
  rm -rf gpurun_out/nt_pmc${v}_FETCH_SIZE gpurun_out/nt_pmc${v}_WRITE_SIZE gpurun_out/nt_pmc${v}_TCC_HIT_sum
done | tee gpurun_out/r05_nt_traffic.txt
