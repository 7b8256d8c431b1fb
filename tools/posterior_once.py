"""One warmed-up cimrgp_block_posterior at size n for kernel-trace timelines:
   rocprofv3 --kernel-trace -d gpurun_out/trace -- python3 tools/posterior_once.py [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cimrgp_amd import device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
q, ns = 2, n // 4
dev.require_gpu()
rng = np.random.default_rng(0)
t = torch.float64
x = dev.to_device(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0), t, "cuda")
y = dev.to_device(rng.normal(size=(n, q)), t, "cuda")
xs = dev.to_device(np.sort(rng.uniform(-1.7, 1.7, size=(ns, 1)), axis=0), t, "cuda")
k = dev.alloc_matrix(n, n, t, "cuda"); w = dev.alloc_matrix(ns + q, n, t, "cuda")
ws = dev.potrf_workspace(n, t, "cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
alpha = torch.zeros((n, q), dtype=t, device="cuda"); z = torch.zeros((n, q), dtype=t, device="cuda")
mean = torch.zeros((ns, q), dtype=t, device="cuda"); var = torch.zeros(ns, dtype=t, device="cuda")
for _ in range(reps):
    torch.cuda.synchronize()
    dev.block_posterior(x, y, xs, 0.1, 1.0, 0.01, k, w, ws, info, alpha, z, mean, var)
    torch.cuda.synchronize()
print("done", int(info.item()))
