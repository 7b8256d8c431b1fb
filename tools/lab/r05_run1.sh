#!/bin/bash
# round-5 GPU call: final race probe set, chain kernel timings, stage stamps of the persistent update, stand-alone update, bench
mkdir -p gpurun_out
RACE_REPS=1000 timeout -k 10 400 bash tools/lab/race_probe.sh run > gpurun_out/r05_race_probe_final.txt 2>&1
echo "stale-word lines: $(grep -c 'EXACTLY the stale word' gpurun_out/r05_race_probe_final.txt)"
grep "^==\|repetitions differ" gpurun_out/r05_race_probe_final.txt
for v in e0 late; do echo "== diag_probe_$v"; timeout -k 10 250 tools/diag_probe_$v 2>&1 | grep -v "^raw\|nine waves):\|^row\|kprev 192\|workgroup"; done > gpurun_out/r05_chain_kernels.txt 2>&1
grep "us per launch\|potrf n\|^==\|beside" gpurun_out/r05_chain_kernels.txt
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_e20.so CIMRGP_GEMM_PERS=256 python3 tools/lab/pers_stamps.py 7936 > gpurun_out/r05_pers_stamps.json 2>gpurun_out/r05_pers_stamps.err
cat gpurun_out/r05_pers_stamps.json; tail -2 gpurun_out/r05_pers_stamps.err
python3 tools/gemm_bench.py --m 7936,6912 --k 256 --reps 20 --check 2>/dev/null
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_a.json
python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_a.json')); print(d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms'])"
