#!/usr/bin/env python3
"""Assemble profiles/rNN_pmc_bench_n8192.json from the passes of tools/collect_profiles.sh:
   python3 tools/make_pmc_profile.py gpurun_out/rNN_pmc_raw.json gpurun_out/rNN_bench_n8192_kernel_stats.csv \
       gpurun_out/rNN_sources.sha256 [commit] > profiles/rNN_pmc_bench_n8192.json
The counters were collected on the GPU box from the code whose source hash is in the .sha256 file (the
box has no git history); `commit` is the commit of this repository whose tree hashes to the same value
(bench.py recomputes the hash and flags the counters as stale when the kernels have changed since)."""
import csv
import json
import re
import sys


def kernel_stats(path):
    out = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            out[r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, total_us=float(r["TotalDurationNs"]) / 1e3)
    return out


def main():
    raw = json.load(open(sys.argv[1]))
    stats = kernel_stats(sys.argv[2])
    sha = open(sys.argv[3]).read().split()[0]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    n = 8192
    out = {
        "command": "tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | 'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                   "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64' --kernel-trace --output-format csv -- python3 bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline (three separate passes); durations from rocprofv3 --kernel-trace --stats "
                   "-- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline (the kernel_stats.csv beside this file)",
        "correction": "gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so read "
                      "bytes <= 2 x FETCH_SIZE x 1024 (upper bound); WRITE_SIZE x 1024 is exact for 16-B-per-lane streaming stores",
        "sources_sha256": sha,
        "sources": "sha256 over cimrgp_amd/csrc/{common.hpp,gemm_tile.hpp,gemm_nt.hip,potrf.hip,gram.hip} in that order (bench.py: _sources_sha)",
        "commit": commit,
    }
    gram_key = next((k for k in raw if k.startswith("k_rbf_gram<double,symmetric")), None)
    gram_stat = next((v for k, v in stats.items() if "k_rbf_gram_lower_wide<double" in k or "k_rbf_gram<double, true" in k), None)
    if gram_key and gram_stat:
        g = raw[gram_key]
        wbytes = g["WRITE_SIZE"]["mean"] * 1024
        rate = wbytes / (gram_stat["avg_us"] * 1e-6) / 1e9
        out["gram"] = dict(kernel="k_rbf_gram_lower_wide<double> (D1, lower triangle in 64 x 128 tiles), N=%d" % n, launches=g["WRITE_SIZE"]["launches"],
                           WRITE_SIZE_KB=g["WRITE_SIZE"]["mean"], FETCH_SIZE_KB=g["FETCH_SIZE"]["mean"],
                           algorithmic_write_bytes=n * (n + 1) // 2 * 8, avg_kernel_us=gram_stat["avg_us"],
                           hbm_write_GBps_rocprof=rate, frac_of_8TBps=rate / 8000.0, frac_of_achievable_6p29TBps=rate / 6290.0)
    fams = [k for k in raw if re.match(r"k_gemm_nt_(pers|sub)<double,lower", k)]
    tot_l = sum(raw[k]["WRITE_SIZE"]["launches"] for k in fams)
    if tot_l:
        traffic = sum(raw[k]["hbm_bytes_per_launch_upper"] * raw[k]["WRITE_SIZE"]["launches"] for k in fams) / tot_l
        out["trailing_update"] = dict(
            kernel="lower trailing updates of cimrgp_potrf (k_gemm_nt_pers<double, lower>: the combined head + bulk launches of the "
                   "look-ahead phase; k_gemm_nt_sub<double, lower, *>: the others), pooled as in bench.py's roofline",
            launches={k: raw[k]["WRITE_SIZE"]["launches"] for k in fams},
            traffic_bytes_per_launch_by_kernel={k: raw[k]["hbm_bytes_per_launch_upper"] for k in fams},
            traffic_bytes_per_launch=traffic,
            note="upper bound (2 x FETCH_SIZE + WRITE_SIZE); bench.py prints the algorithmic bytes beside it")
    mf = next((k for k in fams if "pers" in k and "SQ_VALU_MFMA_BUSY_CYCLES" in raw[k]), None) or \
        next((k for k in fams if "SQ_VALU_MFMA_BUSY_CYCLES" in raw[k]), None)
    if mf:
        r = raw[mf]
        per_xcd = r["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        # the clock the part sustains while this kernel runs: GPU-clock cycles of the launch (GRBM_GUI_ACTIVE per XCD)
        # over its duration in the kernel trace of the same command
        pers_stat = next((v for k, v in stats.items() if "k_gemm_nt_pers<double, true" in k), None)
        if pers_stat and "pers" in mf:
            clock_ghz = per_xcd / (pers_stat["avg_us"] * 1e3)
            out["sustained_clock"] = dict(
                kernel=mf, GRBM_GUI_ACTIVE_per_xcd=per_xcd, avg_kernel_us=pers_stat["avg_us"], clock_ghz=clock_ghz,
                fp64_mfma_peak_at_that_clock_tflops=78.6 * clock_ghz / 2.4,
                note="78.6 TFLOP/s is quoted at 2.4 GHz; in this kernel the part runs slower -- the in-kernel measurement "
                     "(s_memtime / s_memrealtime over a workgroup's lifetime, profiles/r05_pers_stamps.txt) reads 2.09-2.12 GHz, "
                     "in line with this counter-derived figure -- and no kernel can beat the peak at the clock it is given")
        out["trailing_update_mfma"] = dict(
            kernel=mf, launches=r["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"], SQ_VALU_MFMA_BUSY_CYCLES=r["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"],
            GRBM_GUI_ACTIVE_per_xcd=per_xcd, SQ_INSTS_VALU_MFMA_MOPS_F64=r["SQ_INSTS_VALU_MFMA_MOPS_F64"]["mean"],
            mfma_busy_frac=r["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (per_xcd * 1024.0),
            mfma_flops_per_launch=r["SQ_INSTS_VALU_MFMA_MOPS_F64"]["mean"] * 512.0,
            note="SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), GRBM_GUI_ACTIVE / 8 (reported summed over the 8 XCDs): the "
                 "fraction of ALL the chip's matrix pipes' cycles (the kernel runs on 256 - chain_cus compute units).  NOT an "
                 "independent utilisation measurement: the counter equals issued matrix flops / 32 per SIMD-cycle exactly, so this "
                 "is achieved flops / (clock x peak flops per clock)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
