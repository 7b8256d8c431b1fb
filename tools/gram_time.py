"""Time the Gram builder (lower tiles) at size n, and compare with the oracle:  python tools/gram_time.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cimrgp_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev.require_gpu()
rng = np.random.default_rng(0)
x = np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)
xd = dev.to_device(x, torch.float64, "cuda")
k = dev.alloc_matrix(n, n, torch.float64, "cuda")
ts = []
for it in range(25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dev.rbf_gram(xd, 0.1, 1.0, 0.01, lower_only=True, out=k); e1.record(); torch.cuda.synchronize()
    if it >= 5: ts.append(e0.elapsed_time(e1))
m = 1024
ref = np.exp(-0.5 * (x[:m] - x[:m].T) ** 2 / 0.01) + 0.01 * np.eye(m)
got = k[:m, :m].cpu().numpy()
low = np.tril_indices(m)
err = float(np.max(np.abs(got[low] - ref[low]) / ref[low]))
us = float(np.median(ts)) * 1e3
print(json.dumps(dict(n=n, us=round(us, 1), write_TBps=round(n * (n + 1) / 2 * 8 / us / 1e6, 2), frac_of_8=round(n * (n + 1) / 2 * 8 / us / 1e6 / 8, 3), max_rel_err_vs_numpy=err)))
