#!/usr/bin/env python3
"""Headline benchmark: GP posteriors/s and effective Cholesky GFLOP/s at N = 8192.

One "step" = one complete dense GP posterior of one (resolution, partition)
block with inputs resident in HBM: RBF Gram build (D1) -> blocked Cholesky (D2)
-> alpha = K^-1 r (D3) -> predictive mean and variance at N/4 test points
(D4/D5).  With --gpus N every rank runs the same step on its own independent
partition (weak scaling) and the per-step predictions are summed with ONE
all-reduce of the fused [mean | var] buffer (RCCL over xGMI).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (spec)
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA peak


def make_block(n, q, seed):
    """x ~ sorted U(-sqrt3, sqrt3), y_k = sin(3x+k) + 0.5 sin(17x^2) + 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(-np.sqrt(3), np.sqrt(3), size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x + k) + 0.5 * np.sin(17 * x * x) for k in range(q)]) + 0.1 * rng.normal(size=(n, q))
    return x, y


def cpu_baseline(n, ns, q, ell, sf2, noise, seed):
    """The oracle (NumPy/SciPy port) timed on the host cores on ONE posterior of the same
    workload -- a reported baseline, never the product path."""
    import oracle
    x, y = make_block(n, q, seed)
    xs = np.linspace(-1.7, 1.7, ns)[:, None]
    t0 = time.perf_counter()
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    oracle.block_predict(x, fit, xs, ell, sf2, True)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    return dict(value=1.0 / dt, unit="posteriors/s", cores=int(cores), kind="port",
                sample="1 posterior (Gram+Cholesky+solve+mean/var at N/4 points) of the N=%d block, "
                       "NumPy/SciPy FP64, %.1f s" % (n, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as td
    import cimrgp_amd as ca
    from cimrgp_amd import device as dev, dist, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # CIMRGP_BENCH_REHEARSAL=gloo: rehearse the N > 1 path on a box with ONE GPU (all ranks on
        # cuda:0, gloo carrying the device tensors); the driver's multi-GPU runs use nccl = RCCL
        rehearsal = os.environ.get("CIMRGP_BENCH_REHEARSAL", "")
        torch.cuda.set_device(0 if rehearsal else local_rank)
        td.init_process_group(rehearsal or "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    device = dev.require_gpu()
    tdt = dev.as_torch_dtype(args.dtype)
    lib = _lib.load()

    n, q = args.n, 2
    ns = n // 4
    ell, sf2, noise = 0.1, 1.0, 0.01
    x, y = make_block(n, q, 1234 + rank)
    xd = dev.to_device(x, tdt, device)
    yd = dev.to_device(y, tdt, device)
    xsd = dev.to_device(np.linspace(-1.7, 1.7, ns)[:, None], tdt, device)

    # buffers allocated once: the step itself never allocates the big matrices
    kbuf = dev.alloc_matrix(n, n, tdt, device)
    wbuf = dev.alloc_matrix(ns + q, n, tdt, device)      # carried rows: [K(X*, X); r^T]
    ws = dev.potrf_workspace(n, tdt, device)
    info = torch.zeros(1, dtype=torch.int32, device=device)
    fused = torch.zeros((q + 1, ns * world), dtype=tdt, device=device)
    mean = torch.zeros((ns, q), dtype=tdt, device=device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    stage_ms = np.zeros(5)

    def step(timed):
        if timed:
            ev[0].record()
        dev.rbf_gram(xd, ell, sf2, noise, lower_only=True, out=kbuf)               # D1
        if timed:
            ev[1].record()
        dev.rbf_cross(xsd, xd, ell, sf2, out=wbuf)                                  # rows of D5 ...
        wbuf[ns:ns + q, :n] = yd.t()                                                # ... and of D3's forward half
        if timed:
            ev[2].record()
        # D2 with the rows carried through the same panel sweep: W = K* L^-T and z^T = (L^-1 r)^T
        dev.potrf_rows(kbuf, n, wbuf, ns + q, ws, info)
        if timed:
            ev[3].record()
        z = wbuf[ns:ns + q, :n].t().contiguous()
        alpha = dev.solve_lt(kbuf, n, ws, z.clone())                                # D3 backward half
        var = fused[q, rank * ns:(rank + 1) * ns]
        dev.predict_from_w(wbuf, ns, n, z, sf2, 0.0, None, mean, var, accumulate=False)   # D4 + D5 tail
        if timed:
            ev[4].record()
        fused[:q, rank * ns:(rank + 1) * ns] = mean.t()
        dist.allreduce_sum_(fused)                                                  # the one collective
        if timed:
            ev[5].record()

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    assert int(info.item()) == 0, "Cholesky failed in warm-up"

    _lib.check(lib.cimrgp_profile_begin(), "cimrgp_profile_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
        # stage times are read after the loop from the last step's events only
    barrier()
    dt = time.perf_counter() - t0
    tr_ms, tr_fl, tr_cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.cimrgp_profile_collect(ctypes.byref(tr_ms), ctypes.byref(tr_fl), ctypes.byref(tr_cnt)),
               "cimrgp_profile_collect")
    for i in range(5):
        stage_ms[i] = ev[i].elapsed_time(ev[i + 1])
    # Cholesky alone (no carried rows) for the effective-GFLOP/s figure, timed separately
    torch.cuda.synchronize()
    chol_ms = []
    for _ in range(3):
        dev.rbf_gram(xd, ell, sf2, noise, lower_only=True, out=kbuf)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        dev.potrf(kbuf, n, ws, info)
        c1.record()
        torch.cuda.synchronize()
        chol_ms.append(c0.elapsed_time(c1))
    chol_ms = float(np.median(chol_ms))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    assert int(info.item()) == 0

    if rank == 0:
        peak = FP64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
        achieved = (tr_fl.value / (tr_ms.value * 1e-3)) / 1e12 if tr_ms.value > 0 else 0.0
        chol_gflops = (n ** 3 / 3.0) / (chol_ms * 1e-3) / 1e9
        out = {
            "metric": "GP posteriors/sec at N=%d (Gram + Cholesky + solve + predictive mean/var)" % n,
            "value": world * args.steps / dt,
            "unit": "posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 1-D, N=%d single-resolution single-partition RBF GP "
                                   "per GPU, q=2 outputs, N/4=%d test points, ell=0.1, sf2=1, noise=0.01" % (n, ns),
                       "partitions_per_gpu": 1, "parallelism": "independent partitions, 1 all-reduce/step"},
            "cholesky_gflops": chol_gflops,
            "cholesky_frac_of_peak": chol_gflops / 1e3 / peak,
            "stage_ms": {"gram": stage_ms[0], "cross_gram_and_rhs_rows": stage_ms[1],
                         "potrf_with_carried_rows": stage_ms[2], "backward_solve_and_predict": stage_ms[3],
                         "reduce": stage_ms[4], "potrf_alone": chol_ms},
            "gram_gbps_lower": (n * (n + 1) / 2 * (8 if args.dtype == "f64" else 4)) / (stage_ms[0] * 1e-3) / 1e9,
            "roofline": {"bound": "mfma", "kernel": "k_gemm_nt_sub<%s, lower> (Cholesky trailing update)"
                                                    % ("double" if args.dtype == "f64" else "float"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "launches": int(tr_cnt.value),
                         "avg_launch_ms": tr_ms.value / max(1, tr_cnt.value),
                         "avg_launch_gflop": tr_fl.value / max(1, tr_cnt.value) / 1e9,
                         "traffic": None},
        }
        # HBM traffic of that kernel comes from the committed rocprofv3 --pmc passes of this same
        # command (counters cannot be collected from inside the process being timed)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_trailing_update.json")
        if args.dtype == "f64" and n == 8192 and os.path.exists(pmc):
            with open(pmc) as fh:
                pj = json.load(fh)
            out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_trailing_update.json)"
            mm = [n - 256 * (p + 2) for p in range((n // 256) - 2)]
            out["roofline"]["algorithmic_bytes_per_launch"] = float(np.mean([m * (m + 1) / 2 * 8 * 2 + m * 256 * 8 for m in mm]))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, ns, q, ell, sf2, noise, 1234)
        print(json.dumps(out))
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
