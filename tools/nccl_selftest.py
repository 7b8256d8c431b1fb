"""RCCL self-test on one GPU: torch.distributed backend "nccl" with a world of one, bound to the device
as bench.py binds it, through cimrgp_amd.dist (two ranks cannot share a GPU under RCCL, so the N > 1
rehearsal on a one-GPU box uses gloo; this checks that the backend itself initialises and reduces).
   python tools/nccl_selftest.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
import torch, torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cimrgp_amd import dist
x = torch.arange(12, dtype=torch.float64, device="cuda").reshape(3, 4)
dist.allreduce_sum_(x)
td.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); td.all_reduce(t, op=td.ReduceOp.MAX)
print("nccl world 1 ok", float(x.sum()), float(t.item()), dist.world_size() if hasattr(dist, "world_size") else "")
td.destroy_process_group()
