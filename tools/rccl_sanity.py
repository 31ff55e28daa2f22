#!/usr/bin/env python3
"""One-rank RCCL sanity check of the exact collective calls bench.py makes for N > 1 (uint8 all_gather_into_tensor of the
144-byte partial, barrier, float64 all_reduce MAX) — what can be rehearsed on a one-GPU box."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
mine = torch.arange(144, dtype=torch.uint8, device="cuda")
gather = torch.empty(144, dtype=torch.uint8, device="cuda")
host = torch.empty(144, dtype=torch.uint8).pin_memory()
dist.all_gather_into_tensor(gather, mine)
host.copy_(gather, non_blocking=True)
torch.cuda.current_stream().synchronize()
assert host.numpy().tobytes() == bytes(range(144))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
buf = [torch.empty(96, dtype=torch.uint8, device="cuda")]
dist.all_gather(buf, torch.zeros(96, dtype=torch.uint8, device="cuda"))
dist.destroy_process_group()
print("rccl sanity OK")
