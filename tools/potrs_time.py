"""Time cimrgp_potrs (forward + backward skinny solves, q right-hand sides) after a factorisation:
   python tools/potrs_time.py n [q] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cimrgp_amd import device as dev

n = int(sys.argv[1]); q = int(sys.argv[2]) if len(sys.argv) > 2 else 2; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev.require_gpu()
rng = np.random.default_rng(0)
x = torch.as_tensor(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)).cuda()
kbuf = dev.rbf_gram(x, 0.05, 1.0, 0.01, lower_only=True)
ws, info = dev.potrf(kbuf, n)
rhs0 = torch.as_tensor(rng.normal(size=(n, q))).cuda()
best = 1e9
for _ in range(reps + 1):
    rhs = rhs0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dev.potrs(kbuf, n, ws, rhs); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(json.dumps(dict(n=n, q=q, potrs_ms=round(best, 3), lower_triangle_read_gbps=round(2 * n * n / 2 * 8 / best / 1e6, 1), info=int(info.item()))))
