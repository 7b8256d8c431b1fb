"""Time the per-layer batched entry points on nb blocks of n points (d = 2, q = 2):
   python tools/layer_time.py nb n [reps]     (under rocprofv3 --kernel-trace --stats for the kernel breakdown)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cimrgp_amd import device as dev

nb, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev.require_gpu()
rng = np.random.default_rng(0)
d, q, ns = 2, 2, n // 4
x = torch.as_tensor(rng.uniform(-1.7, 1.7, size=(nb * n, d))).cuda()
y = torch.as_tensor(rng.normal(size=(nb * n, q))).cuda()
xs = torch.as_tensor(rng.uniform(-1.7, 1.7, size=(nb * ns, d))).cuda()
starts = torch.arange(nb, dtype=torch.int64, device="cuda") * n
tstarts = torch.arange(nb, dtype=torch.int64, device="cuda") * ns
ld = dev.padded_ld(n)
karena = torch.empty((nb, n, ld), dtype=torch.float64, device="cuda")
wsb = max((dev.potrf_workspace_bytes(n, torch.float64) + 15) // 16 * 16, 16)
ws = torch.empty((nb, wsb), dtype=torch.uint8, device="cuda")
info = torch.zeros(nb, dtype=torch.int32, device="cuda")
bias = torch.empty((nb, q), dtype=torch.float64, device="cuda")
noise = torch.empty(nb, dtype=torch.float64, device="cuda")
z = torch.empty((nb, n, q), dtype=torch.float64, device="cuda")
alpha = torch.empty((nb, n, q), dtype=torch.float64, device="cuda")
tout = torch.zeros_like(y)
mean = torch.zeros((nb * ns, q), dtype=torch.float64, device="cuda")
var = torch.zeros(nb * ns, dtype=torch.float64, device="cuda")
fit, pred = [], []
for it in range(reps + 1):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    dev.layer_fit(x, y, None, tout, starts, n, 0.3, 1.0, 0.01, 0.01, 1e-8, None, None, karena, ws, info, bias, noise, z, alpha)
    e[1].record()
    dev.layer_predict(x, starts, n, xs, tstarts, ns, 0.3, 1.0, karena, ws, z, bias, noise, mean, var)
    e[2].record()
    torch.cuda.synchronize()
    if it:
        fit.append(e[0].elapsed_time(e[1]))
        pred.append(e[1].elapsed_time(e[2]))
flops = nb * n ** 3 / 3.0
print(json.dumps(dict(nb=nb, n=n, fit_ms=round(float(np.median(fit)), 2), predict_ms=round(float(np.median(pred)), 2),
                      chol_tflops_of_fit=round(flops / np.median(fit) / 1e9, 1), info_max=int(info.max().item()))))
