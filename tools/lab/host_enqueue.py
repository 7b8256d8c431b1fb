"""How far ahead of the device is the host in the pipelined step?  (lab) enqueue time vs total time of K steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import workloads
from cimrgp_amd import device as dev
n, q = 8192, 2
ns = n // 4
tdt = torch.float64
x, y = workloads.make_block(n, q, 1234)
xd, yd = dev.to_device(x, tdt, "cuda"), dev.to_device(y, tdt, "cuda")
xsd = dev.to_device(workloads.block_test_points(ns), tdt, "cuda")
def make_set():
    return dict(kbuf=dev.alloc_matrix(n, n, tdt, "cuda"), wbuf=dev.alloc_matrix(ns + q, n, tdt, "cuda"), ws=dev.potrf_workspace(n, tdt, "cuda"),
                info=torch.zeros(1, dtype=torch.int32, device="cuda"), alpha=torch.empty((n, q), dtype=tdt, device="cuda"),
                z=torch.empty((n, q), dtype=tdt, device="cuda"), scratch=torch.empty(2 * q * n, dtype=tdt, device="cuda"),
                mean=torch.zeros((ns, q), dtype=tdt, device="cuda"), var=torch.zeros(ns, dtype=tdt, device="cuda"))
sets = [make_set(), make_set()]
cur = torch.cuda.current_stream()
sq = dev.solve_queue(cur)
def step(i):
    b = sets[i % 2]
    dev.block_posterior(xd, yd, xsd, 0.1, 1.0, 0.01, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"], b["mean"], b["var"],
                        scratch=b["scratch"], streams=(cur, cur, sq))
for i in range(4): step(i)
torch.cuda.synchronize()
K = 12
t0 = time.perf_counter()
for i in range(K): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms per step, device %.2f ms per step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
