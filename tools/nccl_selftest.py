"""RCCL self-test on one GPU: torch.distributed backend "nccl" with a world of one, bound to the device
as bench.py binds it; the fused [mean | var] buffer of a step is reduced THROUGH the backend
(cimrgp_amd.dist.allreduce_sum_ with force=True: at world size 1 the plain call is a no-op).  Two ranks
cannot share a GPU under RCCL, so the N > 1 rehearsal on a one-GPU box uses gloo.
   python tools/nccl_selftest.py"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
import torch, torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cimrgp_amd import dist
fused = torch.arange(3 * 2048, dtype=torch.float64, device="cuda").reshape(3, 2048)      # [mean (q = 2) | var] x N/4
want = float(fused.sum())
dist.allreduce_sum_(fused, force=True)                     # RCCL all-reduce (SUM) of the step's buffer
td.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); td.all_reduce(t, op=td.ReduceOp.MAX)
assert float(fused.sum()) == want
print("nccl world 1 ok", float(fused[0, :12].sum()), float(t.item()), td.get_backend())
td.destroy_process_group()
