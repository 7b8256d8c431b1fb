"""FP32 look-ahead debugging: where does the factor go wrong?  python tools/lab/f32_debug.py n m [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cimrgp_amd import device as dev

n, m = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev.require_gpu()
rng = np.random.default_rng(n + m)
x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
ell, sf2, noise = 0.05, 1.0, 0.1
x64 = dev.to_device(x, torch.float64, "cuda")
k64 = dev.rbf_gram(x64, ell, sf2, noise, lower_only=True)
_, i64 = dev.potrf(k64, n)
l64 = torch.tril(k64[:n, :n])
x32 = dev.to_device(x, torch.float32, "cuda")
for rep in range(reps):
    k32 = dev.rbf_gram(x32, ell, sf2, noise, lower_only=True)
    if m:
        b = dev.alloc_matrix(m, n, torch.float32, "cuda")
        b.normal_()
        _, info = dev.potrf_rows(k32, n, b, m)
    else:
        _, info = dev.potrf(k32, n)
    torch.cuda.synchronize()
    l32 = torch.tril(k32[:n, :n]).double()
    bad = []
    nb = n // 64
    d = (l32 - l64).abs()
    colmax = d.reshape(n, nb, 64).amax(dim=2)            # per row, per 64-column block
    blk = colmax.reshape(nb, 64, nb).amax(dim=1)          # per 64-row block x 64-col block
    blk = torch.nan_to_num(blk, nan=1e30)
    idx = (blk > 1e-2).nonzero()
    first = idx[:6].tolist()
    print("rep %d info64 %d info32 %d bad 64-blocks (row-block, col-block) first: %s count %d" % (rep, int(i64.item()), int(info.item()), first, idx.shape[0]), flush=True)
    if idx.shape[0]:
        rb, cb = first[0]
        sub = d[rb * 64:rb * 64 + 64, cb * 64:cb * 64 + 64].cpu().numpy()
        sub = np.nan_to_num(sub, nan=9e9)
        print("  first bad block (%d, %d): max |dL| per 4-column group (lower part):" % (rb, cb))
        print("  ", ["%.1e" % sub[:, c:c + 4].max() for c in range(0, 64, 4)])
        print("   per 16-row tile x 16-col tile:")
        for rt in range(4):
            print("    ", ["%.1e" % sub[rt * 16:rt * 16 + 16, ct * 16:ct * 16 + 16].max() for ct in range(4)])
        # the block to its left in the same rows (the rows' solved columns of the previous sub-block / panel)
        if cb > 0:
            left = np.nan_to_num(d[rb * 64:rb * 64 + 64, (cb - 1) * 64:cb * 64].cpu().numpy(), nan=9e9)
            print("   block to the left (%d, %d) max |dL| %.1e; block above-left diagonal (%d, %d) max %.1e" % (rb, cb - 1, left.max(), rb - 1, cb - 1, float(np.nan_to_num(d[(rb - 1) * 64:rb * 64, (cb - 1) * 64:cb * 64].cpu().numpy(), nan=9e9).max())))
