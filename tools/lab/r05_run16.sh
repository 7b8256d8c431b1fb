#!/bin/bash
# round-5 GPU call 16: the persistent update's C stream non-temporal (n1: loads and stores, n2: loads, n3: stores) -- time and L2 traffic, one box
mkdir -p gpurun_out
L=$PWD/cimrgp_amd
{
for rep in 1 2; do for v in "" _n1 _n2 _n3; do echo "== variant ${v:-base}"; CIMRGP_LIB_PATH=$L/libcimrgp_tuning$v.so python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check 2>/dev/null | cut -c1-120; done; done
for v in "" _n1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    CIMRGP_LIB_PATH=$L/libcimrgp_tuning$v.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/nt_pmc${v}_$c -- python3 tools/gemm_bench.py --m 7936 --k 256 --reps 3 > /dev/null 2> gpurun_out/nt_pmc.err || { tail -5 gpurun_out/nt_pmc.err; exit 1; }
  done
  echo "== counters, variant ${v:-base}"; python3 tools/pmc_summary.py gpurun_out/nt_pmc${v}_FETCH_SIZE/ gpurun_out/nt_pmc${v}_WRITE_SIZE/ | grep pers | cut -c1-700
  rm -rf gpurun_out/nt_pmc${v}_FETCH_SIZE gpurun_out/nt_pmc${v}_WRITE_SIZE
done
for rep in 1 2; do for v in "" _n1; do echo -n "bench ${v:-base}: "; CIMRGP_LIB_PATH=$L/libcimrgp_tuning$v.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; done; done
} | tee gpurun_out/r05_nt_cstream.txt
