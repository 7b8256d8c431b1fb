"""Stand-alone rate of the trailing update  C -= A A^T (lower)  through cimrgp_syrk_lower:
   python tools/gemm_bench.py [--m 7936,5888] [--k 256] [--reps 20] [--check]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cimrgp_amd import device as dev

ap = argparse.ArgumentParser()
ap.add_argument("--m", default="7936")
ap.add_argument("--k", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--check", action="store_true")
ap.add_argument("--dtype", default="f64")
args = ap.parse_args()
dev.require_gpu()
tdt = dev.as_torch_dtype(args.dtype)
peak = 78.6 if tdt == torch.float64 else 157.3
for m in [int(v) for v in args.m.split(",")]:
    k = args.k
    c = dev.alloc_matrix(m, m, tdt, "cuda")
    a = dev.alloc_matrix(m, k, tdt, "cuda")
    torch.manual_seed(0)
    c.normal_()
    a.normal_()
    if args.check:
        want = torch.tril(c[:m, :m] - a[:m, :k] @ a[:m, :k].t())
        cc = c.clone()
        dev.syrk_lower(cc, a, m, k)
        err = float((torch.tril(cc[:m, :m]) - want).abs().max() / want.abs().max())
        upper_untouched = bool(torch.equal(torch.triu(cc[:m, :m], 129), torch.triu(c[:m, :m], 129)))
    times = []
    for it in range(args.reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dev.syrk_lower(c, a, m, k)
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            times.append(e0.elapsed_time(e1))
    med = float(np.median(times))
    rec = dict(m=m, k=k, us=round(med * 1e3, 1), tflops=round(m * (m + 1.0) * k / med / 1e9, 2),
               frac=round(m * (m + 1.0) * k / med / 1e9 / peak, 3), dtype=args.dtype)
    if args.check:
        rec.update(rel_err=err, far_upper_untouched=upper_untouched)
    print(json.dumps(rec), flush=True)
