#!/usr/bin/env python3
"""Copy the records tools/collect_profiles.sh left in gpurun_out/ into profiles/ and stamp the commit:
   python3 tools/finalize_profiles.py r03
The PMC profile keeps the hash of the kernel sources it was collected from; the commit written beside it is the
current HEAD, and the script refuses when the working tree's sources hash differently (collect again then)."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
h = hashlib.sha256()
for f in ("common.hpp", "gemm_tile.hpp", "gemm_nt.hip", "potrf.hip", "gram.hip"):
    h.update(open(os.path.join(root, "cimrgp_amd", "csrc", f), "rb").read())
sha = h.hexdigest()
commit = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "cimrgp_amd/csrc"], capture_output=True, text=True).stdout.strip()
keep = ["bench_n8192.json", "bench_n8192_under_rocprof.json", "bench_n8192_kernel_stats.csv", "timeline_n8192.txt",
        "timeline_n8192_rows.txt", "pmc_raw.json", "pmc_bench_n8192.json", "potrf_sweep.jsonl", "gemm_standalone.jsonl", "gemm_standalone_tile_kernel.jsonl", "pers_variants.txt", "layer_times.jsonl",
        "config3_n65536.json", "config4_n262144_one_gpu.json", "config4_n8192_two_ranks_gloo.json", "rccl_world_of_one.jsonl", "config5_fp32_vs_fp64_n16384.jsonl", "config5_mfma_counters_f64.json", "config5_mfma_counters_f32.json",
        "chain_kernels.txt"]
for name in keep:
    src = os.path.join(root, "gpurun_out", "%s_%s" % (tag, name))
    if not os.path.exists(src) or os.path.getsize(src) == 0:
        print("missing", src)
        continue
    dst = os.path.join(root, "profiles", "%s_%s" % (tag, name))
    if name == "pmc_bench_n8192.json":
        rec = json.load(open(src))
        if rec.get("sources_sha256") != sha:
            raise SystemExit("the PMC profile was collected from other kernel sources than the working tree's: collect again")
        rec["commit"] = commit + (" (+ uncommitted changes under cimrgp_amd/csrc)" if dirty else "")
        json.dump(rec, open(dst, "w"), indent=1)
    else:
        shutil.copyfile(src, dst)
    print("kept", dst)
