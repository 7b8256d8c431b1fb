"""One warmed-up potrf (optionally with carried rows) at size n for kernel-trace timelines:
   rocprofv3 --kernel-trace -d gpurun_out/trace -- python3 tools/potrf_once.py [n] [reps] [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys

import numpy as np
import torch

from cimrgp_amd import device as dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tdt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
dev.require_gpu()
rng = np.random.default_rng(0)
x = torch.as_tensor(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)).to("cuda", tdt)
xs = torch.as_tensor(np.sort(rng.uniform(-1.7, 1.7, size=(max(rows, 1), 1)), axis=0)).to("cuda", tdt)
for _ in range(reps):
    k = dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=True)
    torch.cuda.synchronize()
    if rows:
        w = dev.alloc_matrix(rows, n, tdt, "cuda")
        dev.rbf_cross(xs, x, 0.1, 1.0, out=w)
        torch.cuda.synchronize()
        dev.potrf_rows(k, n, w, rows)
    else:
        dev.potrf(k, n)
    torch.cuda.synchronize()
print("done")
