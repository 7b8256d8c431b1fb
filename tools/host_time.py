"""Host time to ENQUEUE one N = 8192 bench step (no synchronisation inside): the margin before a step becomes host-bound."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cimrgp_amd import device as dev
n, q = 8192, 2
ns = n // 4
dev.require_gpu()
rng = np.random.default_rng(0)
x = dev.to_device(np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0), torch.float64, "cuda")
y = dev.to_device(rng.normal(size=(n, q)), torch.float64, "cuda")
xs = dev.to_device(np.sort(rng.uniform(-1.7, 1.7, size=(ns, 1)), axis=0), torch.float64, "cuda")
kbuf = dev.alloc_matrix(n, n, torch.float64, "cuda")
wbuf = dev.alloc_matrix(ns + q, n, torch.float64, "cuda")
ws = dev.potrf_workspace(n, torch.float64, "cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
mean = torch.zeros((ns, q), dtype=torch.float64, device="cuda")
var = torch.zeros(ns, dtype=torch.float64, device="cuda")
def step():
    dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=True, out=kbuf)
    dev.rbf_cross(xs, x, 0.1, 1.0, out=wbuf)
    wbuf[ns:ns + q, :n] = y.t()
    dev.potrf_rows(kbuf, n, wbuf, ns + q, ws, info)
    z = wbuf[ns:ns + q, :n].t().contiguous()
    dev.solve_lt(kbuf, n, ws, z.clone())
    dev.predict_from_w(wbuf, ns, n, z, 1.0, 0.0, None, mean, var, accumulate=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
host = []
for _ in range(10):
    t0 = time.perf_counter(); step(); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps(dict(host_enqueue_ms_per_step=round(float(np.median(host)) * 1e3, 3), back_to_back_enqueue_ms=round(t_enq / 20 * 1e3, 3),
                      wall_ms_per_step=round(t_all / 20 * 1e3, 3))))
