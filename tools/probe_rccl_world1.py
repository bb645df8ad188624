"""RCCL reachable from torch.distributed on this box?  One rank, backend "nccl": the calls bench.py makes for N > 1
(barrier, all_reduce on a device tensor, all_gather_object).  python3 tools/probe_rccl_world1.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.0, 2.0], device="cuda")
dist.all_reduce(t)
g = [None]
dist.all_gather_object(g, (0, 0, "x"))
print("nccl world=1 ok", t.tolist(), g)
dist.destroy_process_group()
