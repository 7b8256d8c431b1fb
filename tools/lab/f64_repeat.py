"""FP64 factorisations repeated: every repetition must give the same L as the first.  python tools/lab/f64_repeat.py n reps [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cimrgp_amd import device as dev
n, reps = int(sys.argv[1]), int(sys.argv[2])
tdt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
dev.require_gpu()
rng = np.random.default_rng(n)
x = dev.to_device(np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0), tdt, "cuda")
ref = None
bad = 0
for rep in range(reps):
    k = dev.rbf_gram(x, 0.05, 1.0, 0.1, lower_only=True)
    _, info = dev.potrf(k, n)
    torch.cuda.synchronize()
    l = torch.tril(k[:n, :n]).clone()
    if ref is None:
        ref = l
    elif not torch.equal(l, ref):
        bad += 1
        d = torch.nan_to_num((l - ref).abs(), nan=1e30)
        rows, cols = (d > 0).nonzero()[0].tolist()
        print("rep %d differs: first differing entry (%d, %d), max |d| %.2e, info %d" % (rep, rows, cols, float(d.max()), int(info.item())), flush=True)
print("%s n=%d: %d of %d repetitions differ from the first" % (str(tdt), n, bad, reps - 1))
