"""Which stage separation of cimrgp_block_posterior_staged helps or hurts (lab): N = 8192 steps over rotating sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
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
sets = [make_set() for _ in range(3)]

def run(tag, streams, steps=12, order_reuse=True):
    done = [None] * 3
    def step(i):
        b = sets[i % 3]
        if streams is not None and order_reuse and done[i % 3] is not None:
            streams[0].wait_event(done[i % 3])
        dev.block_posterior(xd, yd, xsd, 0.1, 1.0, 0.01, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"], b["mean"], b["var"],
                            scratch=b["scratch"], streams=streams)
        if streams is not None:
            done[i % 3] = torch.cuda.Event(); done[i % 3].record(streams[2])
    for i in range(4): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-60s %.3f ms/step  %.1f /s" % (tag, ms, 1e3 / ms), flush=True)

from cimrgp_amd import _lib
if len(sys.argv) > 1:
    _lib.set_rows_queues(int(sys.argv[1]))
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)
run("default stream, one-stream call", None)
with torch.cuda.stream(s1):
    run("non-default stream, one-stream call", None)
run("staged, all three = s1", (s1, s1, s1))
run("staged, front = s2, factor = solve = s1", (s2, s1, s1))
run("staged, front = factor = s1, solve = s2", (s1, s1, s2))
run("staged, front = s2, factor = s1, solve = s3", (s2, s1, s3))
run("staged, front = factor = s1, solve = high-priority stream", (s1, s1, hi))
with torch.cuda.stream(s1):
    sq = dev.solve_queue()
print("solve queue is its own stream:", sq.cuda_stream != s1.cuda_stream)
run("staged, front = factor = s1, solve = cimrgp_solve_queue(s1)", (s1, s1, sq))
cur = torch.cuda.current_stream()
sq0 = dev.solve_queue(cur)
run("staged, default stream, solve = cimrgp_solve_queue(default)", (cur, cur, sq0))

# whole calls alternating over K caller streams (K look-ahead contexts): independent blocks in flight together
def run_multi(tag, nstreams, steps=16, staged=False):
    ss = [torch.cuda.Stream() for _ in range(nstreams)]
    bsets = sets + [make_set() for _ in range(max(0, 2 * nstreams - len(sets)))]
    sqs = [dev.solve_queue(s) for s in ss] if staged else None
    def step(i):
        k = i % nstreams
        b = bsets[i % (2 * nstreams)] if staged else bsets[k]
        with torch.cuda.stream(ss[k]):
            dev.block_posterior(xd, yd, xsd, 0.1, 1.0, 0.01, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"], b["mean"], b["var"],
                                scratch=b["scratch"], streams=(ss[k], ss[k], sqs[k]) if staged else None)
    for i in range(2 * nstreams * 2): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-60s %.3f ms/step  %.1f /s" % (tag, ms, 1e3 / ms), flush=True)

run_multi("2 caller streams, whole calls alternating", 2)
run_multi("3 caller streams, whole calls alternating", 3)
run_multi("4 caller streams, whole calls alternating", 4)
run_multi("2 caller streams, staged (solve on each context's queue)", 2, staged=True)
