#!/usr/bin/env python3
"""BASELINE config 5: FP32 vs FP64 Cholesky at n = 16384 per partition -- warmed median timing,
LAPACK-style `info`, and the error of BOTH precisions' predictive mean / variance against the
FP64 CPU oracle on the same inputs (SURVEY.md 8d cfg 5: noise in {1e-1..1e-4}, length-scale in
{1, 0.1}).  Prints one JSON line per cell.   python tools/precision_sweep.py [--n 16384] [--no-oracle]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def posterior(dev, torch, x, y, xs, ell, sf2, noise, tdt, reps=5):
    n, ns = x.shape[0], xs.shape[0]
    xd, yd, xsd = (dev.to_device(a, tdt, "cuda") for a in (x, y, xs))
    kbuf = dev.alloc_matrix(n, n, tdt, "cuda")
    ws = dev.potrf_workspace(n, tdt, "cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    times = []
    for it in range(reps + 2):                       # 2 warm-ups, then the median of `reps`
        dev.rbf_gram(xd, ell, sf2, noise, lower_only=True, out=kbuf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dev.potrf(kbuf, n, ws, info)
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    ms = float(np.median(times))
    alpha = yd.clone()
    z = dev.potrs(kbuf, n, ws, alpha, want_z=True)
    w = dev.rbf_cross(xsd, xd, ell, sf2)
    dev.trsm_rows(kbuf, n, ws, w, ns)
    mean = torch.zeros((ns, y.shape[1]), dtype=tdt, device="cuda")
    var = torch.zeros(ns, dtype=tdt, device="cuda")
    dev.predict_from_w(w, ns, n, z, sf2, 0.0, None, mean, var)
    return int(info.item()), ms, mean.double().cpu().numpy(), var.double().cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    import torch
    import workloads
    from cimrgp_amd import device as dev
    dev.require_gpu()
    n, ns, q = args.n, 2048, 2
    x, y = workloads.make_block(n, q)
    xs = workloads.block_test_points(ns)
    for ell in (1.0, 0.1):
        for noise in (1e-1, 1e-2, 1e-3, 1e-4):
            i64, ms64, m64, v64 = posterior(dev, torch, x, y, xs, ell, 1.0, noise, torch.float64)
            i32, ms32, m32, v32 = posterior(dev, torch, x, y, xs, ell, 1.0, noise, torch.float32)
            row = dict(n=n, ell=ell, noise=noise, info_f64=i64, info_f32=i32,
                       potrf_ms_f64=ms64, potrf_ms_f32=ms32, timing="median of 5 after 2 warm-ups",
                       tflops_f64=n ** 3 / 3 / (ms64 * 1e-3) / 1e12, tflops_f32=n ** 3 / 3 / (ms32 * 1e-3) / 1e12,
                       frac_of_peak_f64=n ** 3 / 3 / (ms64 * 1e-3) / 78.6e12, frac_of_peak_f32=n ** 3 / 3 / (ms32 * 1e-3) / 157.3e12)
            if not args.no_oracle:
                import oracle
                try:
                    fit = oracle.block_fit(x, y, ell, 1.0, noise)
                    om, ov = oracle.block_predict(x, fit, xs, ell, 1.0, True)
                    row["checker"] = "CPU oracle (scipy.linalg, FP64)"
                    if i64 == 0:
                        row["mean_relerr_f64_vs_oracle"] = float(np.max(np.abs(m64 - om)) / np.max(np.abs(om)))
                        row["var_abserr_f64_vs_oracle"] = float(np.max(np.abs(v64 - ov)))
                    if i32 == 0:
                        row["mean_relerr_f32_vs_oracle"] = float(np.max(np.abs(m32 - om)) / np.max(np.abs(om)))
                        row["var_abserr_f32_vs_oracle"] = float(np.max(np.abs(v32 - ov)))
                    row["oracle_var_min"] = float(ov.min())
                except np.linalg.LinAlgError as exc:
                    row["oracle_error"] = str(exc)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
