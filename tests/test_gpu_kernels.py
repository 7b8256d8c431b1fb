"""Parity of every HIP kernel, called through the C ABI, with the FP64 oracle.

Tolerances (relative to the largest reference magnitude of the array):
  f64: 1e-10 for Gram/solves (blocked summation order differs from LAPACK),
  f32: stated per test -- the f32 path is a precision sweep, not a parity claim.
north_star's bar is 1e-5 relative on predictive mean / variance (f64 path).
"""
import os

import numpy as np
import pytest
import scipy.linalg as sla

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from cimrgp_amd import device
    device.require_gpu()
    return device


def _relerr(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / (np.max(np.abs(b)) + 1e-300))


def _data(n, d, seed=0, q=2):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.7, 1.7, size=(n, d))
    x = x[np.argsort(x[:, 0])]
    y = np.stack([np.sin(3 * x[:, 0]) + 0.3 * x[:, -1], np.cos(2 * x[:, 0] * x[:, -1])] +
                 [np.sin((c + 2) * x[:, 0]) for c in range(q - 2)], axis=1)
    y += 0.05 * rng.normal(size=y.shape)
    return x, y


TDT = {"f64": "float64", "f32": "float32"}


@pytest.mark.parametrize("dt,tol", [("f64", 1e-13), ("f32", 2e-6)])
@pytest.mark.parametrize("n,d", [(1, 1), (63, 1), (64, 2), (65, 3), (257, 2), (1000, 1), (130, 8)])
@pytest.mark.parametrize("lower_only", [False, True])
def test_rbf_gram(dev, dt, tol, n, d, lower_only):
    x, _ = _data(n, d, seed=n)
    tdt = getattr(torch, TDT[dt])
    xd = dev.to_device(x, tdt, "cuda")
    buf = dev.rbf_gram(xd, 0.37, 1.9, 0.05, lower_only=lower_only)
    k = buf[:n, :n].double().cpu().numpy()
    ref = oracle.rbf_gram(x, None, 0.37, 1.9, 0.05)
    if lower_only:
        k, ref = np.tril(k), np.tril(ref)
    assert _relerr(k, ref) < tol


@pytest.mark.parametrize("dt,tol", [("f64", 1e-13), ("f32", 2e-6)])
def test_rbf_cross_ragged(dev, dt, tol):
    xa, _ = _data(77, 2, seed=1)
    xb, _ = _data(201, 2, seed=2)
    tdt = getattr(torch, TDT[dt])
    buf = dev.rbf_cross(dev.to_device(xa, tdt, "cuda"), dev.to_device(xb, tdt, "cuda"), 0.8, 0.7)
    assert _relerr(buf[:77, :201].double().cpu().numpy(), oracle.rbf_gram(xa, xb, 0.8, 0.7)) < tol


def _factor(dev, x, ell, sf2, noise, tdt):
    n = x.shape[0]
    xd = dev.to_device(x, tdt, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    ws, info = dev.potrf(kbuf, n)
    return xd, kbuf, ws, info


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 191, 256, 257, 320, 511, 512, 700, 1025])
def test_potrf_f64_matches_lapack(dev, n):
    x, _ = _data(n, 1, seed=n)
    ell, sf2, noise = 0.3, 1.0, 0.01
    _, kbuf, ws, info = _factor(dev, x, ell, sf2, noise, torch.float64)
    assert int(info.item()) == 0
    lmat = np.tril(kbuf[:n, :n].cpu().numpy())
    ref, rinfo = oracle.potrf_lower(oracle.rbf_gram(x, None, ell, sf2, noise))
    assert rinfo == 0
    assert _relerr(lmat, ref) < 1e-10
    # the inverted 64x64 diagonal blocks left in the workspace
    nslab, npan = (n + 63) // 64, (n + 255) // 256
    flat = ws.view(torch.float64).cpu().numpy()
    inv = flat[:nslab * 4096].reshape(-1, 64, 64)
    for s in range((n + 63) // 64):
        w = min(64, n - 64 * s)
        blk = ref[64 * s:64 * s + w, 64 * s:64 * s + w]
        np.testing.assert_allclose(inv[s][:w, :w] @ blk, np.eye(w), atol=1e-9)
        assert np.all(np.triu(inv[s], 1) == 0)
    # the 256x256 blocks (L_pp^-1)^T the skinny solves use
    inv_t = flat[nslab * 4096:nslab * 4096 + npan * 65536].reshape(npan, 256, 256)
    for p in range(npan):
        w = min(256, n - 256 * p)
        blk = ref[256 * p:256 * p + w, 256 * p:256 * p + w]
        np.testing.assert_allclose(inv_t[p][:w, :w].T @ blk, np.eye(w), atol=1e-8)
        assert np.all(np.tril(inv_t[p], -1) == 0)


@pytest.mark.parametrize("n,m", [(5184, 0), (5377, 70), (6000, 33), (5632, 0), (5632, 260), (5500, 385)])
def test_potrf_lookahead_path_matches_lapack(dev, n, m):
    """Sizes just above the one-queue limit (5120): the look-ahead schedule with a ragged last panel
    (5184 = 20 panels + 64, 5377 = 21 + 1, 6000 = 23 + 112 columns), with and without carried rows,
    against LAPACK on the whole matrix (m = 260, 385: carried rows a little beyond a multiple of the
    128-row tile of the far update, which runs on the second rows queue)."""
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    ell, sf2, noise = 0.05, 1.0, 0.01
    xd = dev.to_device(x, torch.float64, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    lref, iref = oracle.potrf_lower(oracle.rbf_gram(x, None, ell, sf2, noise))
    assert iref == 0
    if m:
        bmat = rng.normal(size=(m, n))
        bbuf = dev.alloc_matrix(m, n, torch.float64, "cuda")
        bbuf[:m, :n] = torch.from_numpy(bmat).cuda()
        ws, info = dev.potrf_rows(kbuf, n, bbuf, m)
    else:
        ws, info = dev.potrf(kbuf, n)
    assert int(info.item()) == 0
    lmat = torch.tril(kbuf[:n, :n]).cpu().numpy()
    assert np.max(np.abs(lmat - lref)) / np.max(np.abs(lref)) < 1e-10
    if m:
        import scipy.linalg as sla
        want = sla.solve_triangular(lref, bmat.T, lower=True).T          # B L^-T
        got = bbuf[:m, :n].cpu().numpy()
        assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-9
    # the stored 256 x 256 inverses serve the skinny solves afterwards
    r = rng.normal(size=(n, 2))
    rd = dev.to_device(r, torch.float64, "cuda")
    dev.potrs(kbuf, n, ws, rd)
    import scipy.linalg as sla
    want = sla.cho_solve((lref, True), r)
    assert np.max(np.abs(rd.cpu().numpy() - want)) / np.max(np.abs(want)) < 1e-8


@pytest.mark.parametrize("n", [64, 257, 1025])
def test_potrf_f32(dev, n):
    x, _ = _data(n, 2, seed=n)
    ell, sf2, noise = 0.3, 1.0, 0.05
    _, kbuf, ws, info = _factor(dev, x, ell, sf2, noise, torch.float32)
    assert int(info.item()) == 0
    lmat = np.tril(kbuf[:n, :n].double().cpu().numpy())
    k = oracle.rbf_gram(x, None, ell, sf2, noise)
    # backward error of the factorisation: |L L^T - K| / |K|  (f32 eps ~ 6e-8, n terms)
    assert _relerr(lmat @ lmat.T, k) < 2e-5


@pytest.mark.parametrize("n,m", [(5632, 0), (5632, 130), (8192, 0), (8192, 130), (16384, 0), (16384, 130)])
def test_potrf_f32_lookahead_backward_error(dev, n, m):
    """FP32 ABOVE the one-queue limit (look-ahead schedule, persistent / paired far updates, carried rows on their own
    queues): backward error of the factor, |L L^T - K|_F / |K|_F < 8 n eps32 (the factor is formed in FP64 from the
    FP32 L on the GPU), and the carried rows W = B L^-T against the FP64 oracle at noise 0.1 -- a wrong tile shows
    as an error of order one, conditioning (~n / noise) as 1e-3."""
    rng = np.random.default_rng(n + m)
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    ell, sf2, noise = 0.05, 1.0, 0.1
    xd = dev.to_device(x, torch.float32, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    k64 = torch.tril(kbuf[:n, :n]).double()
    k64 = k64 + torch.tril(k64, -1).t()
    if m:
        bmat = rng.normal(size=(m, n))
        bbuf = dev.alloc_matrix(m, n, torch.float32, "cuda")
        bbuf[:m, :n] = torch.from_numpy(bmat).to("cuda", torch.float32)
        b64 = bbuf[:m, :n].double().cpu().numpy()                  # the rows as the kernel saw them
        ws, info = dev.potrf_rows(kbuf, n, bbuf, m)
    else:
        ws, info = dev.potrf(kbuf, n)
    assert int(info.item()) == 0
    l64 = torch.tril(kbuf[:n, :n]).double()
    resid = float(torch.linalg.norm(l64 @ l64.t() - k64) / torch.linalg.norm(k64))
    eps32 = float(np.finfo(np.float32).eps)
    assert resid < 8 * n * eps32, resid
    assert resid < 64 * np.sqrt(n) * eps32, resid                  # what a blocked factorisation really delivers
    if m:
        lref, iref = oracle.potrf_lower(k64.cpu().numpy())
        assert iref == 0
        want = sla.solve_triangular(lref, b64.T, lower=True).T      # B L^-T in FP64, from the FP32 inputs
        got = bbuf[:m, :n].double().cpu().numpy()
        err = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        assert err < 2e-3, err


def test_potrf_f32_beside_a_running_update_is_bit_equal_to_the_undisturbed_run(dev):
    """Round 5: the situation of the round-4 wrong results, run ONCE.  An FP32 factorisation above
    the one-queue limit (four-wave chain kernels beside their own FP32 trailing updates) while ANOTHER stream keeps FP32
    updates of a second matrix running on every compute unit: the factor must equal, bit for bit, the one computed with
    nothing beside it, and pass the backward-error bar.  The cause was a hand-off through LDS (tools/lab/race_probe.hip:
    the gathering wave's last ds_write instructions were not in the LDS array when the pivot wave read behind the
    barrier); a chain workgroup that shares its compute unit with an update workgroup is what exposed it."""
    n, mu, k = 5632, 6144, 256
    rng = np.random.default_rng(77)
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    ell, sf2, noise = 0.05, 1.0, 0.1
    xd = dev.to_device(x, torch.float32, "cuda")

    def factor():
        kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
        ws, info = dev.potrf(kbuf, n)
        return kbuf, info

    kref, iref = factor()
    torch.cuda.synchronize()
    assert int(iref.item()) == 0
    k64 = torch.tril(dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)[:n, :n]).double()
    k64 = k64 + torch.tril(k64, -1).t()
    cbuf = dev.alloc_matrix(mu, mu, torch.float32, "cuda")
    abuf = dev.alloc_matrix(mu, k, torch.float32, "cuda")
    cbuf.zero_(); abuf.zero_()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(40):                                   # ~40 x 0.3 ms of FP32 tile-kernel workgroups on every unit
            dev.syrk_lower(cbuf, abuf, mu, k)
    kgot, igot = factor()
    torch.cuda.synchronize()
    assert int(igot.item()) == 0
    assert torch.equal(torch.tril(kgot[:n, :n]), torch.tril(kref[:n, :n]))
    l64 = torch.tril(kgot[:n, :n]).double()
    resid = float(torch.linalg.norm(l64 @ l64.t() - k64) / torch.linalg.norm(k64))
    assert resid < 64 * np.sqrt(n) * float(np.finfo(np.float32).eps), resid


def test_more_caller_streams_than_contexts(dev):
    """Ten caller streams factor matrices above the one-queue limit at the same time: beyond the eight look-ahead
    contexts of a device callers share a context by hash -- and must not share its gate counter (only the context's
    owner stream uses the gate; round 3 let any caller of the first context reset and count on it).  Every factor
    against LAPACK."""
    n = 5376
    rng = np.random.default_rng(3)
    xs = [np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0) for _ in range(2)]
    ell, sf2, noise = 0.05, 1.0, 0.01
    refs = []
    for x in xs:
        lref, iref = oracle.potrf_lower(oracle.rbf_gram(x, None, ell, sf2, noise))
        assert iref == 0
        refs.append(lref)
    streams = [torch.cuda.Stream() for _ in range(10)]
    xds = [dev.to_device(x, torch.float64, "cuda") for x in xs]
    bufs, infos = [], []
    for rep in range(2):                                    # second round: every stream's context exists already
        bufs, infos = [], []
        torch.cuda.synchronize()
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                kbuf = dev.rbf_gram(xds[i % 2], ell, sf2, noise, lower_only=True)
                ws, info = dev.potrf(kbuf, n)
                bufs.append((kbuf, ws))
                infos.append(info)
        torch.cuda.synchronize()
        for i in range(len(streams)):
            assert int(infos[i].item()) == 0, (rep, i, int(infos[i].item()))
            lmat = torch.tril(bufs[i][0][:n, :n]).cpu().numpy()
            assert np.max(np.abs(lmat - refs[i % 2])) / np.max(np.abs(refs[i % 2])) < 1e-10, (rep, i)


def test_potrf_reports_first_bad_pivot(dev):
    x = np.array([[0.0], [0.5], [0.5], [1.0]] + [[2.0 + i] for i in range(70)])   # duplicated point, no noise
    _, _, _, info = _factor(dev, x, 1.0, 1.0, 0.0, torch.float64)
    _, ref = oracle.potrf_lower(oracle.rbf_gram(x, None, 1.0, 1.0, 0.0))
    assert ref == 3
    assert int(info.item()) == 3
    with pytest.raises(np.linalg.LinAlgError):
        dev.raise_if_not_pd(info)


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 5e-3)])
@pytest.mark.parametrize("n,q", [(1, 2), (64, 2), (65, 1), (257, 3), (700, 2), (1025, 8)])
def test_potrs(dev, dt, tol, n, q):
    x, y = _data(n, 1, seed=n, q=max(q, 2))
    y = y[:, :q]
    ell, sf2, noise = 0.3, 1.0, 0.02
    tdt = getattr(torch, TDT[dt])
    _, kbuf, ws, info = _factor(dev, x, ell, sf2, noise, tdt)
    rhs = dev.to_device(y, tdt, "cuda")
    z = dev.potrs(kbuf, n, ws, rhs, want_z=True)
    assert int(info.item()) == 0
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    assert _relerr(z.double().cpu().numpy(), fit["z"]) < tol
    assert _relerr(rhs.double().cpu().numpy(), fit["alpha"]) < tol


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 3e-3)])
@pytest.mark.parametrize("n,m", [(256, 1), (512, 15), (768, 16), (1024, 17), (1280, 129), (1300, 31), (2048, 2050)])
def test_trsm_rows_one_launch_per_panel(dev, dt, tol, n, m):
    """The rows' one-launch panel step (k_rows_step: 16 rows per workgroup, previous panel's update fused with the 256-wide
    solve) through cimrgp_trsm_rows: row counts around the 16-row workgroup, one to eight full panels, a ragged last
    panel (which keeps the two-launch form behind a fused chain), both precisions, against the oracle's triangular solve."""
    x, _ = _data(n, 2, seed=n + m)
    ell, sf2, noise = 0.5, 1.2, 0.05
    tdt = getattr(torch, TDT[dt])
    xd, kbuf, ws, info = _factor(dev, x, ell, sf2, noise, tdt)
    assert int(info.item()) == 0
    rng = np.random.default_rng(n * 7 + m)
    b = rng.normal(size=(m, n))
    w = dev.alloc_matrix(m, n, tdt, "cuda")
    w[:m, :n] = dev.to_device(b, tdt, "cuda")
    dev.trsm_rows(kbuf, n, ws, w, m)
    lref, _ = oracle.potrf_lower(oracle.rbf_gram(x, None, ell, sf2, noise))
    ref = sla.solve_triangular(lref, b.T, lower=True).T
    assert _relerr(w[:m, :n].double().cpu().numpy(), ref) < tol


@pytest.mark.parametrize("n,m", [(64, 5), (257, 130), (700, 64), (512, 1000), (2500, 70), (3072, 33)])
def test_trsm_rows_f64(dev, n, m):
    x, _ = _data(n, 2, seed=n)
    xs, _ = _data(m, 2, seed=n + 1)
    ell, sf2, noise = 0.5, 1.2, 0.01
    xd, kbuf, ws, _ = _factor(dev, x, ell, sf2, noise, torch.float64)
    w = dev.rbf_cross(dev.to_device(xs, torch.float64, "cuda"), xd, ell, sf2)
    dev.trsm_rows(kbuf, n, ws, w, m)
    lref, _ = oracle.potrf_lower(oracle.rbf_gram(x, None, ell, sf2, noise))
    ref = sla.solve_triangular(lref, oracle.rbf_gram(xs, x, ell, sf2).T, lower=True).T
    assert _relerr(w[:m, :n].cpu().numpy(), ref) < 1e-9


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 2e-3)])
@pytest.mark.parametrize("n,ns,d", [(64, 1, 1), (257, 37, 2), (600, 1000, 1)])
def test_predict_mean_and_variance(dev, dt, tol, n, ns, d):
    x, y = _data(n, d, seed=n)
    xs, _ = _data(ns, d, seed=n + 7)
    ell, sf2, noise = 0.4, 1.3, 0.02
    tdt = getattr(torch, TDT[dt])
    xd, kbuf, ws, _ = _factor(dev, x, ell, sf2, noise, tdt)
    alpha = dev.to_device(y, tdt, "cuda")
    z = dev.potrs(kbuf, n, ws, alpha, want_z=True)
    xsd = dev.to_device(xs, tdt, "cuda")
    bias = dev.to_device(np.array([0.25, -1.5]), tdt, "cuda")
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    mref, vref = oracle.block_predict(x, fit, xs, ell, sf2, True)
    # D4 fused mean, overwrite then accumulate
    m1 = dev.predict_mean(xd, alpha, xsd, ell, sf2, bias)
    assert _relerr(m1.double().cpu().numpy(), mref + np.array([0.25, -1.5])) < tol
    dev.predict_mean(xd, alpha, xsd, ell, sf2, None, out=m1, accumulate=True)
    assert _relerr(m1.double().cpu().numpy(), 2 * mref + np.array([0.25, -1.5])) < tol
    # D5 through W = K* L^-T
    w = dev.rbf_cross(xsd, xd, ell, sf2)
    dev.trsm_rows(kbuf, n, ws, w, ns)
    mean = torch.zeros((ns, 2), dtype=tdt, device="cuda")
    var = torch.zeros(ns, dtype=tdt, device="cuda")
    dev.predict_from_w(w, ns, n, z, sf2, 0.0, None, mean, var, accumulate=False)
    assert _relerr(mean.double().cpu().numpy(), mref) < tol
    # variance is a difference of near-equal numbers: compare on the scale of sf2
    assert float(np.max(np.abs(var.double().cpu().numpy() - vref))) / sf2 < tol


@pytest.mark.parametrize("dt,tol", [("f64", 1e-9), ("f32", 5e-3)])
@pytest.mark.parametrize("n,m,q", [(64, 5, 2), (300, 70, 1), (777, 130, 2), (1100, 260, 3)])
def test_potrf_rows_carries_rhs_rows(dev, dt, tol, n, m, q):
    """One sweep: K = L L^T and [K*; r^T] <- [K*; r^T] L^-T, then the backward-only solve."""
    x, y = _data(n, 2, seed=n, q=max(q, 2))
    y = y[:, :q]
    xs, _ = _data(m, 2, seed=n + 3)
    ell, sf2, noise = 0.5, 1.1, 0.02
    tdt = getattr(torch, TDT[dt])
    xd = dev.to_device(x, tdt, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    rows = dev.alloc_matrix(m + q, n, tdt, "cuda")
    dev.rbf_cross(dev.to_device(xs, tdt, "cuda"), xd, ell, sf2, out=rows)
    rows[m:m + q, :n] = dev.to_device(y, tdt, "cuda").t()
    ws, info = dev.potrf_rows(kbuf, n, rows, m + q)
    assert int(info.item()) == 0
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    wref = sla.solve_triangular(fit["L"], oracle.rbf_gram(xs, x, ell, sf2).T, lower=True).T
    got = rows[:m + q, :n].double().cpu().numpy()
    assert _relerr(got[:m], wref) < tol
    assert _relerr(got[m:].T, fit["z"]) < tol
    assert _relerr(np.tril(kbuf[:n, :n].double().cpu().numpy()), fit["L"]) < tol
    z = rows[m:m + q, :n].t().contiguous()
    alpha = dev.solve_lt(kbuf, n, ws, z)
    assert _relerr(alpha.double().cpu().numpy(), fit["alpha"]) < tol


def test_residual_chain_helpers(dev):
    rng = np.random.default_rng(3)
    n, q = 1000, 2
    y = rng.normal(size=(n, q)) * np.array([1.0, 3.0]) + np.array([0.5, -2.0])
    fbar = rng.normal(size=(n, q)) * 0.1
    yd, fd = (dev.to_device(a, torch.float64, "cuda") for a in (y, fbar))
    stats = dev.block_stats(yd, fd)
    r = y - fbar
    bias = r.mean(axis=0)
    pooled = np.mean((r - bias) ** 2)
    np.testing.assert_allclose(stats.cpu().numpy(), np.r_[bias, pooled], rtol=1e-12)
    noise = dev.noise_from_stats(stats, q, 0.01, 1e-8)
    np.testing.assert_allclose(noise.item(), 0.01 * pooled, rtol=1e-12)
    assert dev.noise_from_stats(stats, q, 0.0, 1e-8).item() == 1e-8
    rd = dev.residual(yd, fd, stats[:q])
    np.testing.assert_allclose(rd.cpu().numpy(), r - bias, rtol=0, atol=1e-14)
    alpha = rng.normal(size=(n, q))
    out = torch.ones((n, q), dtype=torch.float64, device="cuda")
    dev.train_mean(rd, dev.to_device(alpha, torch.float64, "cuda"), stats[:q], noise, out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), 1 + (r - bias) - 0.01 * pooled * alpha + bias, rtol=1e-12, atol=1e-13)
    # stats with fbar = None
    np.testing.assert_allclose(dev.block_stats(yd, None).cpu().numpy()[:q], y.mean(axis=0), rtol=1e-12)
    # add_diag and log-determinant
    x, _ = _data(300, 1, seed=4)
    xd = dev.to_device(x, torch.float64, "cuda")
    kbuf = dev.rbf_gram(xd, 0.3, 1.0, 0.0, lower_only=True)
    dev.add_diag(kbuf, 300, noise)
    kref = oracle.rbf_gram(x, None, 0.3, 1.0, float(noise.item()))
    np.testing.assert_allclose(np.diag(kbuf[:300, :300].cpu().numpy()), np.diag(kref), rtol=1e-14)
    dev.potrf(kbuf, 300)
    lref, _ = oracle.potrf_lower(kref)
    np.testing.assert_allclose(dev.logdet_half(kbuf, 300).item(), np.sum(np.log(np.diag(lref))), rtol=1e-10)


def test_dense_golden_fixtures(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "dense_oracle.npz"))
    for tag in ["n64_d1", "n257_d2", "n512_d1"]:
        x, y, xs = g[tag + "_x"], g[tag + "_y"], g[tag + "_xs"]
        ell, sf2, noise = (float(v) for v in g[tag + "_hyp"])
        n = x.shape[0]
        xd, kbuf, ws, info = _factor(dev, x, ell, sf2, noise, torch.float64)
        assert int(info.item()) == 0
        lmat = kbuf[:n, :n].cpu().numpy()
        np.testing.assert_allclose(np.diag(lmat), g[tag + "_Ldiag"], rtol=1e-9)
        np.testing.assert_allclose(lmat[-1], g[tag + "_Llast"], rtol=0, atol=1e-9)
        alpha = dev.to_device(y, torch.float64, "cuda")
        z = dev.potrs(kbuf, n, ws, alpha, want_z=True)
        assert _relerr(alpha.cpu().numpy(), g[tag + "_alpha"]) < 1e-8
        assert _relerr(z.cpu().numpy(), g[tag + "_z"]) < 1e-9
        xsd = dev.to_device(xs, torch.float64, "cuda")
        m = dev.predict_mean(xd, alpha, xsd, ell, sf2, None)
        assert _relerr(m.cpu().numpy(), g[tag + "_mean"]) < 1e-9


def test_hip_posterior_agrees_with_scikit_learn(dev, golden_dir):
    """The HIP path held directly to the scikit-learn fixtures (an implementation of the same exact
    GP that is independent of this repository's oracle; tests/golden/make_sklearn_golden.py)."""
    g = np.load(os.path.join(golden_dir, "sklearn_gp.npz"))
    for tag in [str(t) for t in g["cases"]]:
        x, y, xs = g[tag + "_x"], g[tag + "_y"], g[tag + "_xs"]
        ell, sf2, noise = (float(v) for v in g[tag + "_hyp"])
        n, ns = x.shape[0], xs.shape[0]
        xd, yd, xsd = (dev.to_device(a, torch.float64, "cuda") for a in (x, y, xs))
        kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
        ws, info = dev.potrf(kbuf, n)
        assert int(info.item()) == 0
        alpha = yd.clone()
        z = dev.potrs(kbuf, n, ws, alpha, want_z=True)
        w = dev.rbf_cross(xsd, xd, ell, sf2)
        dev.trsm_rows(kbuf, n, ws, w, ns)
        mean = torch.zeros((ns, 2), dtype=torch.float64, device="cuda")
        var = torch.zeros(ns, dtype=torch.float64, device="cuda")
        dev.predict_from_w(w, ns, n, z, sf2, 0.0, None, mean, var)
        np.testing.assert_allclose(mean.cpu().numpy(), g[tag + "_mean"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(var.cpu().numpy(), g[tag + "_var"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("dt,tol", [("f64", 1e-10), ("f32", 5e-4)])
@pytest.mark.parametrize("n,batch,m", [(1, 2, 0), (130, 3, 2), (1000, 4, 3), (2049, 2, 0), (1024, 32, 2), (2048, 16, 3), (1300, 24, 2)])
def test_potrf_rows_batched_matches_single(dev, dt, tol, n, batch, m):
    """``batch`` equal-sized factorisations in the same launches (blocks of one layer): each must equal
    LAPACK on its own matrix, carried rows and backward solve included; a non-PD member reports its own
    ``info`` without disturbing the others."""
    tdt = getattr(torch, TDT[dt])
    rng = np.random.default_rng(100 * n + batch)
    ld = dev.padded_ld(n)
    karena = torch.empty((batch, n, ld), dtype=tdt, device="cuda")
    ws_bytes = (dev.potrf_workspace_bytes(n, tdt) + 15) // 16 * 16
    ws_arena = torch.empty((batch, max(ws_bytes, 16)), dtype=torch.uint8, device="cuda")
    info = torch.full((batch,), 7, dtype=torch.int32, device="cuda")
    xs_, ks_ = [], []
    for i in range(batch):
        x = np.sort(rng.uniform(-2, 2, size=(n, 1)), axis=0)
        xs_.append(x)
        ks_.append(oracle.rbf_gram(x, None, 0.3 + 0.1 * i, 1.0, 0.05))
        dev.rbf_gram(dev.to_device(x, tdt, "cuda"), 0.3 + 0.1 * i, 1.0, 0.05, lower_only=True, out=karena[i])
    rows = bmat = None
    if m:
        bmat = rng.normal(size=(batch, m, n))
        rows = torch.zeros((batch, m, ld), dtype=tdt, device="cuda")
        rows[:, :, :n] = torch.from_numpy(bmat).to("cuda", tdt)
    dev.potrf_rows_batched(karena, n, ld, ws_arena, info, rows, m, ld if m else 0)
    assert info.cpu().tolist() == [0] * batch
    for i in range(batch):
        lref, _ = oracle.potrf_lower(ks_[i])
        got = torch.tril(karena[i][:, :n]).double().cpu().numpy()
        assert _relerr(got, lref) < tol
        if m:
            want = sla.solve_triangular(lref, bmat[i].T, lower=True).T
            assert _relerr(rows[i][:, :n].double().cpu().numpy(), want) < 10 * tol
    # batched backward halves against LAPACK
    q = 2
    zs = rng.normal(size=(batch, n, q))
    z = torch.from_numpy(zs).to("cuda", tdt).contiguous()
    dev.solve_lt_batched(karena, n, ld, ws_arena, z)
    for i in range(batch):
        lref, _ = oracle.potrf_lower(ks_[i])
        want = sla.solve_triangular(lref, zs[i], lower=True, trans="T")
        assert _relerr(z[i].double().cpu().numpy(), want) < 100 * tol
    if n >= 130 and dt == "f64":
        # member 1 made singular (two identical points, no noise): only ITS info is set
        for i in range(batch):
            x = xs_[i].copy()
            if i == 1:
                x[5] = x[4]
            dev.rbf_gram(dev.to_device(x, tdt, "cuda"), 0.3, 1.0, 0.0 if i == 1 else 0.05, lower_only=True, out=karena[i])
        dev.potrf_rows_batched(karena, n, ld, ws_arena, info)
        got = info.cpu().tolist()
        assert got[1] > 0 and all(g == 0 for j, g in enumerate(got) if j != 1)


@pytest.mark.parametrize("front", [False, True])
@pytest.mark.parametrize("nsets", [2, 1])
@pytest.mark.parametrize("n,ns,q", [(700, 50, 2), (5632, 130, 2)])
def test_block_posterior_solve_stage_on_the_idle_context_queue(dev, n, ns, q, nsets, front):
    """cimrgp_solve_queue: the solve stage of block i on the look-ahead context's idle queue, beside the front end and
    first panels of block i+1, over TWO rotating buffer sets with no ordering by the caller other than its own reads:
    bit for bit the one-stream results (small n: the factorisation itself does not use the context).  With ONE buffer set
    (a caller's mistake) the front end waits for the previous solve stage: still the right results, no overlap.
    front: the front end on cimrgp_front_queue (the context's chain queue, idle in a factorisation's last third), so that
    block i+1's Gram matrices run beside block i's factorisation -- the same results, ordered by the call's own nets."""
    tdt = torch.float64
    nblocks = 5
    rng = np.random.default_rng(n + 1)
    blocks = []
    for b in range(nblocks):
        x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
        y = np.stack([np.cos(3 * x[:, 0] + c + b) for c in range(q)], axis=1) + 0.1 * rng.normal(size=(n, q))
        xs = rng.uniform(-2.0, 2.0, size=(ns, 1))
        blocks.append(tuple(dev.to_device(a, tdt, "cuda") for a in (x, y, xs)))
    ell, sf2, noise = 0.05, 1.1, 0.02

    def buffers():
        return dict(kbuf=dev.alloc_matrix(n, n, tdt, "cuda"), wbuf=dev.alloc_matrix(ns + q, n, tdt, "cuda"),
                    ws=dev.potrf_workspace(n, tdt, "cuda"), info=torch.zeros(1, dtype=torch.int32, device="cuda"),
                    alpha=torch.zeros((n, q), dtype=tdt, device="cuda"), z=torch.zeros((n, q), dtype=tdt, device="cuda"),
                    scratch=torch.empty(2 * q * n, dtype=tdt, device="cuda"))
    means = [torch.zeros((ns, q), dtype=tdt, device="cuda") for _ in range(2 * nblocks)]
    vars_ = [torch.zeros(ns, dtype=tdt, device="cuda") for _ in range(2 * nblocks)]
    bs = buffers()
    for i, (xd, yd, xsd) in enumerate(blocks):
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, bs["kbuf"], bs["wbuf"], bs["ws"], bs["info"], bs["alpha"], bs["z"],
                            means[i], vars_[i], scratch=bs["scratch"])
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    sq = dev.solve_queue(cur)
    assert sq.cuda_stream != cur.cuda_stream
    fq = dev.front_queue(cur) if front else cur
    assert (fq.cuda_stream != cur.cuda_stream and fq.cuda_stream != sq.cuda_stream) or not front
    sets = [buffers() for _ in range(nsets)]
    for i, (xd, yd, xsd) in enumerate(blocks):
        b = sets[i % nsets]
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"],
                            means[nblocks + i], vars_[nblocks + i], scratch=b["scratch"], streams=(fq, cur, sq))
    torch.cuda.synchronize()
    for i in range(nblocks):
        assert torch.equal(means[i], means[nblocks + i]) and torch.equal(vars_[i], vars_[nblocks + i])
    assert all(int(b["info"].item()) == 0 for b in sets)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("n,ns", [(300, 0), (6144, 0), (6144, 37)])
def test_block_posterior_staged_edge_cases(dev, dt, n, ns):
    """The staged call without test points (ns = 0: fit only), with accumulation into mean / variance, in both
    precisions: bit for bit the one-stream call; equal streams ARE that call."""
    tdt = getattr(torch, TDT[dt])
    q = 2
    rng = np.random.default_rng(n + ns)
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    y = np.stack([np.sin(3 * x[:, 0] + c) for c in range(q)], axis=1) + 0.1 * rng.normal(size=(n, q))
    xd, yd = dev.to_device(x, tdt, "cuda"), dev.to_device(y, tdt, "cuda")
    xsd = dev.to_device(rng.uniform(-2.0, 2.0, size=(ns, 1)), tdt, "cuda") if ns else None
    ell, sf2, noise = 0.05, 1.1, 0.1

    def run(streams):
        kbuf = dev.alloc_matrix(n, n, tdt, "cuda")
        wbuf = dev.alloc_matrix(ns + q, n, tdt, "cuda")
        ws = dev.potrf_workspace(n, tdt, "cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        alpha = torch.zeros((n, q), dtype=tdt, device="cuda")
        z = torch.zeros((n, q), dtype=tdt, device="cuda")
        mean = torch.full((max(ns, 1), q), 0.5, dtype=tdt, device="cuda")
        var = torch.full((max(ns, 1),), 0.25, dtype=tdt, device="cuda")
        torch.cuda.synchronize()
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, kbuf, wbuf, ws, info, alpha, z, mean if ns else None, var if ns else None,
                            add_noise=True, accumulate=True, streams=streams)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        return torch.tril(kbuf[:n, :n]).clone(), alpha, z, mean, var

    cur = torch.cuda.current_stream()
    ref = run(None)
    same = run((cur, cur, cur))
    piped = run((cur, cur, dev.solve_queue(cur)))
    # ... and with the front end on cimrgp_front_queue: the first panel of the factorisation then follows it on that queue
    # (n = 6144: look-ahead; the small size owns no context and gets its own stream back)
    fronted = run((dev.front_queue(cur), cur, dev.solve_queue(cur)))
    for got in (same, piped, fronted):
        assert all(torch.equal(a, b) for a, b in zip(ref, got))
    if ns:
        assert float(ref[3].min()) != 0.5 or float(ref[3].max()) != 0.5      # accumulated onto the 0.5 / 0.25 fills
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    assert _relerr(ref[1].double().cpu().numpy(), fit["alpha"]) < (1e-8 if dt == "f64" else 5e-3)


@pytest.mark.parametrize("n,ns,q", [(700, 50, 2), (5632, 130, 2)])
def test_block_posterior_staged_pipeline_matches_the_one_stream_call(dev, n, ns, q):
    """cimrgp_block_posterior_staged: independent blocks pipelined over three streams and three rotating buffer sets (front
    end of block i+1 and solve of block i-1 beside the factorisation of block i) give, block by block, bit for bit what
    the one-stream call gives."""
    tdt = torch.float64
    nblocks, nsets = 5, 3
    rng = np.random.default_rng(n)
    blocks = []
    for b in range(nblocks):
        x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
        y = np.stack([np.sin(3 * x[:, 0] + c + b) for c in range(q)], axis=1) + 0.1 * rng.normal(size=(n, q))
        xs = rng.uniform(-2.0, 2.0, size=(ns, 1))
        blocks.append(tuple(dev.to_device(a, tdt, "cuda") for a in (x, y, xs)))
    ell, sf2, noise = 0.05, 1.1, 0.02

    def buffers():
        return dict(kbuf=dev.alloc_matrix(n, n, tdt, "cuda"), wbuf=dev.alloc_matrix(ns + q, n, tdt, "cuda"),
                    ws=dev.potrf_workspace(n, tdt, "cuda"), info=torch.zeros(1, dtype=torch.int32, device="cuda"),
                    alpha=torch.zeros((n, q), dtype=tdt, device="cuda"), z=torch.zeros((n, q), dtype=tdt, device="cuda"),
                    scratch=torch.empty(2 * q * n, dtype=tdt, device="cuda"))
    # reference: one stream -- the pipelined run's factorisation stream: the look-ahead schedule (and with it the last
    # bits of L) belongs to the stream's context -- and one buffer set
    streams = tuple(torch.cuda.Stream() for _ in range(3))
    ref = []
    bs = buffers()
    for (xd, yd, xsd) in blocks:
        mean = torch.zeros((ns, q), dtype=tdt, device="cuda")
        var = torch.zeros(ns, dtype=tdt, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[1]):
            dev.block_posterior(xd, yd, xsd, ell, sf2, noise, bs["kbuf"], bs["wbuf"], bs["ws"], bs["info"], bs["alpha"], bs["z"], mean, var,
                                scratch=bs["scratch"])
        torch.cuda.synchronize()
        assert int(bs["info"].item()) == 0
        ref.append((mean, var, bs["alpha"].clone()))
    torch.cuda.synchronize()
    # pipelined
    sets = [buffers() for _ in range(nsets)]
    done = [None] * nsets
    got = []
    for i, (xd, yd, xsd) in enumerate(blocks):
        s = i % nsets
        if done[s] is not None:
            streams[0].wait_event(done[s])                   # the set's last reader
            with torch.cuda.stream(streams[2]):
                got[i - nsets] = got[i - nsets][:2] + (sets[s]["alpha"].clone(),)
            done[s] = torch.cuda.Event(); done[s].record(streams[2])
            streams[0].wait_event(done[s])
        mean = torch.zeros((ns, q), dtype=tdt, device="cuda")
        var = torch.zeros(ns, dtype=tdt, device="cuda")
        streams[0].wait_stream(torch.cuda.current_stream())  # the zero fills above
        b = sets[s]
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"], mean, var,
                            scratch=b["scratch"], streams=streams)
        done[s] = torch.cuda.Event(); done[s].record(streams[2])
        got.append((mean, var, None))
    torch.cuda.synchronize()
    for i in range(nblocks):
        alpha = got[i][2] if got[i][2] is not None else sets[i % nsets]["alpha"]
        assert torch.equal(got[i][0], ref[i][0]) and torch.equal(got[i][1], ref[i][1]) and torch.equal(alpha, ref[i][2])
    assert all(int(b["info"].item()) == 0 for b in sets)


def test_block_posterior_staged_two_stream_pairs_on_one_device(dev):
    """Round 5: TWO caller stream pairs on one device alternate staged calls, each pair over its own two
    buffer sets in rotation with nothing ordered by the caller, then a plain call reuses a set whose solve stage may
    still be running.  The library keeps one in-flight record per (stream, stream_solve) pair and per buffer set
    (round 4: one per device, so each pair overwrote the other's and both safety nets were lost).  Every block's
    posterior against the oracle; the regions of a layer are independent (src/Posteriors.py:35-59)."""
    tdt = torch.float64
    n, ns, q = 5376, 64, 2                 # above the one-queue size: each pair's stream gets a look-ahead context
    ell, sf2, noise = 0.05, 1.1, 0.02
    rng = np.random.default_rng(55)
    nper = 4
    blocks = []
    for b in range(2 * nper + 1):
        x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
        y = np.stack([np.sin(3 * x[:, 0] + c + b) for c in range(q)], axis=1) + 0.1 * rng.normal(size=(n, q))
        xs = rng.uniform(-2.0, 2.0, size=(ns, 1))
        blocks.append((x, y, xs))
    dblocks = [tuple(dev.to_device(a, tdt, "cuda") for a in blk) for blk in blocks]

    def buffers():
        return dict(kbuf=dev.alloc_matrix(n, n, tdt, "cuda"), wbuf=dev.alloc_matrix(ns + q, n, tdt, "cuda"),
                    ws=dev.potrf_workspace(n, tdt, "cuda"), info=torch.zeros(1, dtype=torch.int32, device="cuda"),
                    alpha=torch.zeros((n, q), dtype=tdt, device="cuda"), z=torch.zeros((n, q), dtype=tdt, device="cuda"),
                    scratch=torch.empty(2 * q * n, dtype=tdt, device="cuda"))
    # earlier tests of this process may have used up the device's eight look-ahead contexts (a ninth caller stream
    # shares one by hash and has no solve queue of its own): start from none
    torch.cuda.synchronize()
    from cimrgp_amd import _lib
    _lib.check(_lib.load().cimrgp_shutdown(), "cimrgp_shutdown")
    pairs = []
    for _ in range(2):
        st = torch.cuda.Stream()
        sq = dev.solve_queue(st)
        assert sq.cuda_stream != st.cuda_stream
        pairs.append((st, sq, [buffers(), buffers()]))
    assert pairs[0][1].cuda_stream != pairs[1][1].cuda_stream
    means = [torch.zeros((ns, q), dtype=tdt, device="cuda") for _ in blocks]
    vars_ = [torch.zeros(ns, dtype=tdt, device="cuda") for _ in blocks]
    alphas = [None] * len(blocks)
    infos = [None] * len(blocks)
    torch.cuda.synchronize()
    for i in range(2 * nper):
        st, sq, sets = pairs[i % 2]
        b = sets[(i // 2) % 2]
        xd, yd, xsd = dblocks[i]
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"],
                            means[i], vars_[i], scratch=b["scratch"], streams=(st, st, sq))
        if i == 2 * nper - 2:
            continue                                 # its set is reused by the plain call below with NOTHING in between
        with torch.cuda.stream(sq):                  # behind this call's solve stage, ahead of the set's next use
            alphas[i] = b["alpha"].clone()
            infos[i] = b["info"].clone()
    # a PLAIN call on pair 0's most recently used set, on a third stream, right behind the staged call that used it
    i = 2 * nper
    st3 = torch.cuda.Stream()
    b = pairs[0][2][((2 * nper - 2) // 2) % 2]
    xd, yd, xsd = dblocks[i]
    with torch.cuda.stream(st3):
        dev.block_posterior(xd, yd, xsd, ell, sf2, noise, b["kbuf"], b["wbuf"], b["ws"], b["info"], b["alpha"], b["z"],
                            means[i], vars_[i], scratch=b["scratch"])
        alphas[i] = b["alpha"].clone()
        infos[i] = b["info"].clone()
    torch.cuda.synchronize()
    for i, (x, y, xs) in enumerate(blocks):
        fit = oracle.block_fit(x, y, ell, sf2, noise)
        om, ov = oracle.block_predict(x, fit, xs, ell, sf2, True)
        if alphas[i] is not None:
            assert int(infos[i].item()) == 0
            assert _relerr(alphas[i].cpu().numpy(), fit["alpha"]) < 1e-8, i
        assert _relerr(means[i].cpu().numpy(), om) < 1e-8, i
        assert float(np.max(np.abs(vars_[i].cpu().numpy() - ov))) / sf2 < 1e-9, i


@pytest.mark.parametrize("n,ns,q,d", [(300, 70, 2, 2), (1100, 130, 3, 1), (5377, 70, 2, 1), (6000, 260, 1, 2)])
def test_block_posterior_one_call_matches_the_separate_calls_and_the_oracle(dev, n, ns, q, d):
    """cimrgp_block_posterior: Gram, factorisation with the cross-Gram rows and the targets carried, backward solve,
    predictive mean and variance in ONE call -- bit for bit what the separate calls give, and the oracle's
    posterior; one-queue sizes and the look-ahead schedule with ragged last panels."""
    rng = np.random.default_rng(n + ns)
    x = rng.uniform(-2.0, 2.0, size=(n, d))
    x = x[np.argsort(x[:, 0])]
    y = np.stack([np.sin(2 * x[:, 0] + c) for c in range(q)], axis=1) + 0.1 * rng.normal(size=(n, q))
    xs = rng.uniform(-2.0, 2.0, size=(ns, d))
    ell, sf2, noise = 0.3 if d == 2 else 0.05, 1.2, 0.02
    tdt = torch.float64
    xd, yd, xsd = (dev.to_device(a, tdt, "cuda") for a in (x, y, xs))
    kbuf = dev.alloc_matrix(n, n, tdt, "cuda")
    wbuf = dev.alloc_matrix(ns + q, n, tdt, "cuda")
    ws = dev.potrf_workspace(n, tdt, "cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    alpha = torch.zeros((n, q), dtype=tdt, device="cuda")
    z = torch.zeros((n, q), dtype=tdt, device="cuda")
    mean = torch.zeros((ns, q), dtype=tdt, device="cuda")
    var = torch.zeros(ns, dtype=tdt, device="cuda")
    dev.block_posterior(xd, yd, xsd, ell, sf2, noise, kbuf, wbuf, ws, info, alpha, z, mean, var)
    assert int(info.item()) == 0
    # the separate calls
    k2 = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    w2 = dev.alloc_matrix(ns + q, n, tdt, "cuda")
    dev.rbf_cross(xsd, xd, ell, sf2, out=w2)
    w2[ns:ns + q, :n] = yd.t()
    ws2, info2 = dev.potrf_rows(k2, n, w2, ns + q)
    z2 = w2[ns:ns + q, :n].t().contiguous()
    a2 = dev.solve_lt(k2, n, ws2, z2.clone())
    m2 = torch.zeros_like(mean)
    v2 = torch.zeros_like(var)
    dev.predict_from_w(w2, ns, n, z2, sf2, 0.0, None, m2, v2, accumulate=False)
    assert torch.equal(torch.tril(kbuf[:n, :n]), torch.tril(k2[:n, :n]))
    assert torch.equal(wbuf[:ns + q, :n], w2[:ns + q, :n])
    assert torch.equal(z, z2) and torch.equal(alpha, a2) and torch.equal(mean, m2) and torch.equal(var, v2)
    # the oracle
    fit = oracle.block_fit(x, y, ell, sf2, noise)
    om, ov = oracle.block_predict(x, fit, xs, ell, sf2, True)
    assert _relerr(alpha.cpu().numpy(), fit["alpha"]) < 1e-8
    assert _relerr(mean.cpu().numpy(), om) < 1e-8
    assert float(np.max(np.abs(var.cpu().numpy() - ov))) / sf2 < 1e-9
