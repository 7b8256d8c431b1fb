"""Median time of cimrgp_potrf (and of cimrgp_potrf_rows with N/4 + 2 carried rows) over sizes:
   python tools/potrf_sweep.py [--sizes 2048,4096,...] [--reps 7] [--rows]
Prints one JSON line per size.  Environment knobs of the library (CIMRGP_RESERVE_CUS, ...)
are read at first use, so compare settings across processes."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from cimrgp_amd import device as dev

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2048,4096,6144,8192,12288,16384")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--rows", action="store_true")
ap.add_argument("--dtype", default="f64")
args = ap.parse_args()
dev.require_gpu()
tdt = dev.as_torch_dtype(args.dtype)
rng = np.random.default_rng(0)
for n in [int(s) for s in args.sizes.split(",")]:
    x = torch.as_tensor(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)).to("cuda", tdt)
    kbuf = dev.alloc_matrix(n, n, tdt, "cuda")
    ws = dev.potrf_workspace(n, tdt, "cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    m = n // 4 + 2
    wbuf = dev.alloc_matrix(m, n, tdt, "cuda") if args.rows else None
    xs = torch.as_tensor(np.linspace(-1.7, 1.7, m)[:, None]).to("cuda", tdt) if args.rows else None
    times = []
    for it in range(args.reps + 2):
        dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=True, out=kbuf)
        if args.rows:
            dev.rbf_cross(xs, x, 0.1, 1.0, out=wbuf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if args.rows:
            dev.potrf_rows(kbuf, n, wbuf, m, ws, info)
        else:
            dev.potrf(kbuf, n, ws, info)
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    med = float(np.median(times))
    flops = n ** 3 / 3 + (float(n) * n * m if args.rows else 0.0)
    print(json.dumps(dict(n=n, rows=(m if args.rows else 0), dtype=args.dtype, ms=round(med, 3), min_ms=round(min(times), 3),
                          tflops=round(flops / med / 1e9, 1), info=int(info.item()),
                          reserve=os.environ.get("CIMRGP_RESERVE_CUS", "default"))), flush=True)
    del kbuf, ws, wbuf
    torch.cuda.empty_cache()
