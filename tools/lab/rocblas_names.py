"""Which rocBLAS (Tensile) kernels serve FP64 GEMMs of the update's shapes (lab: run under rocprofv3 --kernel-trace)."""
import torch
c = torch.randn(7936, 7936, dtype=torch.float64, device="cuda")
for kk in (256, 1024):
    a = torch.randn(7936, kk, dtype=torch.float64, device="cuda")
    for _ in range(3):
        torch.addmm(c, a, a.t(), alpha=-1.0, out=c)
b = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
for _ in range(3):
    torch.mm(b, b)
torch.cuda.synchronize()
