#!/bin/bash
# Collect the judged profiles of the N = 8192 bench on the GPU box (run from the repo root):
#   bash tools/collect_profiles.sh r04 [part ...]     parts: bench stats timelines pmc sweep configs rccl variants config5 chain (default: all but config5)
# -> gpurun_out/r03_*; then locally:  python3 tools/finalize_profiles.py r03   (copies what is kept into profiles/,
# stamps the commit).  rocprofv3: counters in their own passes with --kernel-trace only; the program itself after `--`.
tag=${1:-rXX}; shift || true
parts=${*:-"bench stats timelines pmc sweep configs rccl variants chain"}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
want() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
( cd cimrgp_amd/csrc && cat common.hpp gemm_tile.hpp gemm_nt.hip potrf.hip gram.hip | sha256sum ) > $out/${tag}_sources.sha256
if want bench; then
    python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_n8192.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
fi
if want stats; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline \
        > $out/${tag}_bench_n8192_under_rocprof.json 2> $out/${tag}_stats.err || { tail -5 $out/${tag}_stats.err; exit 1; }
    cp "$(ls $out/${tag}_stats/*/*kernel_stats.csv | tail -1)" $out/${tag}_bench_n8192_kernel_stats.csv
    rm -rf $out/${tag}_stats
fi
if want timelines; then
    for rows in 0 2050; do
        rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_tr -- python3 tools/potrf_once.py 8192 3 $rows > $out/${tag}_tr.log 2>&1 || { tail -5 $out/${tag}_tr.log; exit 1; }
        if grep -q "SIGSEGV\|Segmentation" $out/${tag}_tr.log; then echo "profiler run crashed at exit (rows=$rows)"; exit 1; fi
        if [ $rows = 0 ]; then f=$out/${tag}_timeline_n8192.txt; else f=$out/${tag}_timeline_n8192_rows.txt; fi
        python3 tools/timeline.py $out/${tag}_tr > $f; rm -rf $out/${tag}_tr
    done
fi
if want pmc; then
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
        name=$(echo $pass | cut -d' ' -f1)
        rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/${tag}_pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline \
            > /dev/null 2> $out/${tag}_pmc_$name.err || { tail -5 $out/${tag}_pmc_$name.err; exit 1; }
    done
    python3 tools/pmc_summary.py $out/${tag}_pmc_FETCH_SIZE/ $out/${tag}_pmc_WRITE_SIZE/ $out/${tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES/ > $out/${tag}_pmc_raw.json
    rm -rf $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES
    python3 tools/make_pmc_profile.py $out/${tag}_pmc_raw.json $out/${tag}_bench_n8192_kernel_stats.csv $out/${tag}_sources.sha256 > $out/${tag}_pmc_bench_n8192.json
fi
if want sweep; then
    python3 tools/potrf_sweep.py --sizes 1024,2048,4096,6144,8192,12288,16384,32768 > $out/${tag}_potrf_sweep.jsonl 2>/dev/null
    python3 tools/potrf_sweep.py --sizes 2048,4096,8192,16384 --rows >> $out/${tag}_potrf_sweep.jsonl 2>/dev/null
    python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 >> $out/${tag}_potrf_sweep.jsonl 2>/dev/null
    python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check > $out/${tag}_gemm_standalone.jsonl 2>/dev/null
    # the tile-per-workgroup kernel on the same sizes (tuning build: the persistent form switched off)
    CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_GEMM_PERS=0 python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check \
        > $out/${tag}_gemm_standalone_tile_kernel.jsonl 2>/dev/null
    # one layer's batched fit + prediction, stand-alone
    { python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1; python3 tools/layer_time.py 64 4096 3 2>/dev/null | tail -1; } > $out/${tag}_layer_times.jsonl
fi
if want configs; then
    python3 bench.py --config 3 --steps 3 --warmup 1 > $out/${tag}_config3_n65536.json 2> $out/${tag}_config3.err || tail -5 $out/${tag}_config3.err
    python3 bench.py --config 4 --steps 3 --warmup 1 > $out/${tag}_config4_n262144_one_gpu.json 2> $out/${tag}_config4.err || tail -5 $out/${tag}_config4.err
    CIMRGP_BENCH_REHEARSAL=gloo python3 bench.py --config 4 --gpus 2 --n 8192 --steps 2 --warmup 1 > $out/${tag}_config4_n8192_two_ranks_gloo.json 2>> $out/${tag}_config4.err
fi
if want rccl; then
    for mode in "" "--nccl-world1" "--nccl-world1 --comm torch" "--nccl-world1 --rows-queues 1"; do
        python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $mode 2>/dev/null | tail -1
    done > $out/${tag}_rccl_world_of_one.jsonl
fi
if want config5; then
    # BASELINE configs[4]: FP32 vs FP64 at n = 16384 -- info, errors against the CPU oracle, timings (several minutes of host BLAS) ...
    python3 tools/precision_sweep.py > $out/${tag}_config5_fp32_vs_fp64_n16384.jsonl 2> $out/${tag}_config5.err || tail -5 $out/${tag}_config5.err
    # ... and the matrix-core counters of one factorisation in each precision
    for dt in f64 f32; do
        rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
            -d $out/${tag}_c5pmc_$dt -- python3 tools/potrf_once.py 16384 2 0 $dt > /dev/null 2> $out/${tag}_c5pmc_$dt.err || { tail -5 $out/${tag}_c5pmc_$dt.err; exit 1; }
        python3 tools/pmc_summary.py $out/${tag}_c5pmc_$dt/ > $out/${tag}_config5_mfma_counters_$dt.json
        rm -rf $out/${tag}_c5pmc_$dt
    done
fi
if want chain && [ -x tools/diag_probe_e0 ]; then
    # the chain kernels stand-alone: both forms agree bit for bit, are repeatable (also beside a running FP32 update), launch rates
    timeout -k 10 600 tools/diag_probe_e0 | grep -v "^raw\|nine waves):\|^row\|kprev 192" > $out/${tag}_chain_kernels.txt
fi
if want variants && [ -f cimrgp_amd/libcimrgp_tuning_e1.so ]; then
    # timing-only builds of the persistent kernel (built beforehand: bash tools/lab/exp_variants.sh build)
    bash tools/lab/exp_variants.sh > $out/${tag}_pers_variants.txt 2>&1
fi
echo collected
