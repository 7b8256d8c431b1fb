"""Time cimrgp_potrf alone at size n (f64): python tools/potrf_time.py n [reps] [stream]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import json
import numpy as np
import torch
from cimrgp_amd import device as dev

n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
own_stream = len(sys.argv) > 3 and sys.argv[3] == "stream"      # run on a non-blocking stream instead of torch's default (null) stream
dev.require_gpu()
if own_stream:
    torch.cuda.set_stream(torch.cuda.Stream())
rng = np.random.default_rng(0)
x = torch.as_tensor(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)).cuda()
kbuf = dev.alloc_matrix(n, n, torch.float64, "cuda")
ws = dev.potrf_workspace(n, torch.float64, "cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(reps + 1):
    dev.rbf_gram(x, 0.05, 1.0, 0.01, lower_only=True, out=kbuf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dev.potrf(kbuf, n, ws, info); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(json.dumps(dict(n=n, potrf_ms=round(best, 2), tflops=round(n ** 3 / 3 / best / 1e9, 1), info=int(info.item()))))
