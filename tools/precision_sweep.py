#!/usr/bin/env python3
"""BASELINE config 5: FP32 vs FP64 Cholesky at n = 16384 per partition -- timing, LAPACK-style
`info`, and the error of the FP32 predictive mean / variance against the FP64 path on the same
inputs (SURVEY.md 8d cfg 5: noise in {1e-1..1e-4}, length-scale in {1, 0.1}).  Prints JSON lines.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def posterior(dev, torch, x, y, xs, ell, sf2, noise, tdt):
    n, ns = x.shape[0], xs.shape[0]
    xd, yd, xsd = (dev.to_device(a, tdt, "cuda") for a in (x, y, xs))
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    ws = dev.potrf_workspace(n, tdt, "cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dev.potrf(kbuf, n, ws, info)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
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
    args = ap.parse_args()
    import torch
    from cimrgp_amd import device as dev
    dev.require_gpu()
    rng = np.random.default_rng(1234)
    n, ns, q = args.n, 2048, 2
    x = np.sort(rng.uniform(-np.sqrt(3), np.sqrt(3), size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x + k) + 0.5 * np.sin(17 * x * x) for k in range(q)]) + 0.1 * rng.normal(size=(n, q))
    xs = np.linspace(-1.7, 1.7, ns)[:, None]
    for ell in (1.0, 0.1):
        for noise in (1e-1, 1e-2, 1e-3, 1e-4):
            i64, ms64, m64, v64 = posterior(dev, torch, x, y, xs, ell, 1.0, noise, torch.float64)
            i32, ms32, m32, v32 = posterior(dev, torch, x, y, xs, ell, 1.0, noise, torch.float32)
            row = dict(n=n, ell=ell, noise=noise, info_f64=i64, info_f32=i32,
                       potrf_ms_f64=ms64, potrf_ms_f32=ms32,
                       tflops_f64=n ** 3 / 3 / (ms64 * 1e-3) / 1e12, tflops_f32=n ** 3 / 3 / (ms32 * 1e-3) / 1e12)
            if i64 == 0 and i32 == 0:
                row["mean_relerr_f32"] = float(np.max(np.abs(m32 - m64)) / np.max(np.abs(m64)))
                row["var_abserr_over_sf2_f32"] = float(np.max(np.abs(v32 - v64)))
                row["var_f64_min"] = float(v64.min())
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
