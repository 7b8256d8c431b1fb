#!/bin/bash
# Collect the judged profiles of the N = 8192 bench on the GPU box (run from the repo root):
#   bash tools/collect_profiles.sh r02b     -> gpurun_out/r02b_*  (copy what is kept into profiles/)
# rocprofv3: counters in their own passes with --kernel-trace only; the program itself after `--`.
tag=${1:-rXX}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 bench.py --steps 10 --warmup 2 > $out/${tag}_bench_n8192.json 2> $out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    > $out/${tag}_bench_n8192_under_rocprof.json 2> $out/${tag}_stats.err || exit 1
cp "$(ls $out/${tag}_stats/*/*kernel_stats.csv | tail -1)" $out/${tag}_bench_n8192_kernel_stats.csv
rm -rf $out/${tag}_stats
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_tr -- python3 tools/potrf_once.py 8192 3 > /dev/null 2>&1 || exit 1
python3 tools/timeline.py $out/${tag}_tr > $out/${tag}_timeline_n8192.txt; rm -rf $out/${tag}_tr
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_tr -- python3 tools/potrf_once.py 8192 3 2050 > /dev/null 2>&1 || exit 1
python3 tools/timeline.py $out/${tag}_tr > $out/${tag}_timeline_n8192_rows.txt; rm -rf $out/${tag}_tr
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/${tag}_pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> $out/${tag}_pmc_$name.err || exit 1
done
python3 tools/pmc_summary.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/${tag}_pmc_raw.json
rm -rf $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES
echo collected
