"""Micro-benchmark of the three reduced-rank kernels (HBM-bound skinny passes over Phi).
Usage: python tools/bench_reduced.py [n] [m] [q]   -> one JSON line per kernel."""
import json
import sys

import numpy as np
import torch

import cimrgp_amd as ca
from cimrgp_amd import _lib, device as dev


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return beg.elapsed_time(end) / iters * 1e-3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    q = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev.require_gpu()
    lib = _lib.load()
    for dtype, code, es in ((torch.float64, _lib.F64, 8), (torch.float32, _lib.F32, 4)):
        x = torch.rand((n, 1), dtype=dtype, device="cuda") * 2 - 1
        iv = torch.tensor([1.05], dtype=torch.float64, device="cuda")
        phi = torch.empty((n, m), dtype=dtype, device="cuda")
        y = torch.randn((n, q), dtype=dtype, device="cuda")
        fbar = torch.randn((n, q), dtype=dtype, device="cuda")
        fvar = torch.rand(n, dtype=dtype, device="cuda")
        eau = torch.randn((q, m), dtype=torch.float64, device="cuda")
        bias = torch.randn(q, dtype=torch.float64, device="cuda")
        c2 = torch.rand(m, dtype=torch.float64, device="cuda")
        out = torch.empty(m * q + 2 * m + q + 2, dtype=torch.float64, device="cuda")
        scratch = torch.empty(lib.cimrgp_basis_moments_scratch_bytes(n, m, q) // 8, dtype=torch.float64, device="cuda")
        mean = torch.zeros((n, q), dtype=dtype, device="cuda")
        var = torch.zeros(n, dtype=dtype, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        t_b = timed(lambda: lib.cimrgp_laplace_basis(code, x.data_ptr(), n, 1, iv.data_ptr(), m, phi.data_ptr(), st))
        t_m = timed(lambda: lib.cimrgp_basis_moments(code, x.data_ptr(), n, 1, iv.data_ptr(), m, y.data_ptr(), fbar.data_ptr(),
                                                     fvar.data_ptr(), eau.data_ptr(), q, out.data_ptr(), scratch.data_ptr(), st))
        t_a = timed(lambda: lib.cimrgp_basis_apply(code, x.data_ptr(), n, 1, iv.data_ptr(), m, eau.data_ptr(), q,
                                                   bias.data_ptr(), c2.data_ptr(), 0.5, mean.data_ptr(), var.data_ptr(), 0, st))
        bytes_phi = n * m * es
        # algorithmic bytes: what each kernel must move (Phi is written by the first, never read by the others)
        for name, t, nbytes in (("laplace_basis", t_b, bytes_phi + n * es),
                                ("basis_moments", t_m, n * (1 + 2 * q + 1) * es),
                                ("basis_apply", t_a, n * (1 + q + 1) * es)):
            print(json.dumps(dict(kernel=name, dtype=str(dtype).split(".")[1], n=n, m=m, q=q, ms=round(t * 1e3, 4),
                                  algorithmic_GBps=round(nbytes / t / 1e9, 1), frac_of_8TBps=round(nbytes / t / 8e12, 3),
                                  fma_per_point=m * (q + 3), gfma_per_s=round(n * m * (q + 3) / t / 1e9, 1))))


if __name__ == "__main__":
    main()
