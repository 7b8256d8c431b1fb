import sys, numpy as np, torch
sys.path.insert(0, '.')
from cimrgp_amd import device as dev
dev.require_gpu()
rng = np.random.default_rng(0)
worst = 0
for n in [513, 769, 1280, 4863, 4864, 4865, 5119, 5120, 5121, 5376, 5633, 6000, 7000, 8448, 8960, 9217, 12000]:
    x = torch.as_tensor(np.sort(rng.uniform(-2, 2, size=(n, 1)), axis=0)).cuda()
    k = dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=False)
    kf = k[:n, :n].clone()
    ws, info = dev.potrf(k, n)
    l = torch.tril(k[:n, :n])
    v = torch.randn(n, 2, dtype=torch.float64, device='cuda')
    e = float((l @ (l.t() @ v) - kf @ v).abs().max() / (kf @ v).abs().max())
    # rows variant
    k2 = dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=True)
    w = dev.alloc_matrix(37, n, torch.float64, 'cuda'); w[:37, :n] = torch.randn(37, n, dtype=torch.float64, device='cuda'); w0 = w[:37, :n].clone()
    _, info2 = dev.potrf_rows(k2, n, w, 37)
    e2 = float((torch.tril(k2[:n, :n]) - l).abs().max())
    e3 = float((w[:37, :n] @ l.t() - w0).abs().max() / w0.abs().max())
    print(n, int(info.item()), int(info2.item()), '%.1e %.1e %.1e' % (e, e2, e3))
    worst = max(worst, e, e3); assert int(info.item()) == 0 and e < 1e-11 and e2 < 1e-11 and e3 < 1e-9
print('ok worst', worst)
