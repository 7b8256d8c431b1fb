"""Write-only rates on this GPU: torch fill of an n x ld f64 matrix (the ceiling the Gram builder's stores can reach)."""
import json, sys
import numpy as np, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k = torch.empty((n, n + 16), dtype=torch.float64, device="cuda")
half = torch.empty((n // 2, n + 16), dtype=torch.float64, device="cuda")
for name, t in (("full", k), ("half", half)):
    ts = []
    for it in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.fill_(1.0); e1.record(); torch.cuda.synchronize()
        if it >= 5: ts.append(e0.elapsed_time(e1))
    us = float(np.median(ts)) * 1e3
    print(json.dumps(dict(what="fill_" + name, bytes=t.numel() * 8, us=round(us, 1), TBps=round(t.numel() * 8 / us / 1e6, 2))))
