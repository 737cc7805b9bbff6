"""Can RCCL run two ranks on ONE device?  (python -m torch.distributed.run --nproc-per-node 2 this.py)"""
import os, sys
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(8, device=dev) * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over nccl on a shared device ok -> {t[0].item()}", flush=True)
    dist.destroy_process_group()
except Exception as e:   # noqa: BLE001
    print(f"rank {rank}: RCCL refused two ranks on one device: {type(e).__name__}: {str(e)[:600]}", flush=True)
    sys.exit(0)
