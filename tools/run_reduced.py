"""Time the reduced-rank model (fiMRGP / ciMRGP) end to end on the GPU and show where the host
time goes.  Usage: python tools/run_reduced.py [N] [resolution] [n_basis] [n_iter] [fi|ci] [--profile]"""
import cProfile
import json
import pstats
import sys
import time

import numpy as np
import torch

import cimrgp_amd as ca


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 65536
    res = int(args[1]) if len(args) > 1 else 4
    m = int(args[2]) if len(args) > 2 else 40
    n_iter = int(args[3]) if len(args) > 3 else 5
    forced = (args[4] if len(args) > 4 else "fi") == "fi"
    rng = np.random.default_rng(1234)
    x = np.sort(rng.uniform(-np.sqrt(3), np.sqrt(3), size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x + k) + 0.5 * np.sin(17 * x ** 2) for k in range(2)]) + 0.1 * rng.normal(size=(n, 2))
    ns = n // 4
    xs = np.sort(rng.uniform(-np.sqrt(3), np.sqrt(3), size=(ns, 1)), axis=0)
    t0 = time.perf_counter()
    model = ca.MultiResolutionGaussianProcess(train_xy=[x, y], n_basis=m, index_set_obj=ca.IndexSetUniform(n, res, 2),
                                              basis_function_obj=ca.LaplacianEigenpairs(),
                                              spectral_density_obj=ca.MaternKernel(nu=1, l=1, sf=1), forced_independence=forced)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prof = cProfile.Profile() if "--profile" in sys.argv else None
    if prof:
        prof.enable()
    model.fit(n_iter, None)
    torch.cuda.synchronize()
    if prof:
        prof.disable()
    t2 = time.perf_counter()
    idx_t = ca.IndexSetUniform(ns, res, 2)
    mean = model.get_predicted_mean(xs, idx_t)
    var = model.get_central_moment2(xs, idx_t)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    truth = np.hstack([np.sin(3 * xs + k) + 0.5 * np.sin(17 * xs ** 2) for k in range(2)])
    print(json.dumps(dict(workload="reduced-rank %s" % ("fiMRGP" if forced else "ciMRGP"), n=n, resolution=res,
                          blocks=sum(model.n_regions), n_basis=m, n_iter=n_iter, ctor_s=round(t1 - t0, 3),
                          fit_s=round(t2 - t1, 3), s_per_sweep=round((t2 - t1) / n_iter, 4), predict_s=round(t3 - t2, 3),
                          rmse=float(np.sqrt(np.mean((mean - truth) ** 2))), var_finite=bool(np.all(np.isfinite(var))))))
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
