"""Can two torch.distributed ranks share ONE GPU on this box (nccl = RCCL)?  Run under torch.distributed.run with 2 ranks."""
import os
import sys

import torch
import torch.distributed as dist

backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend, rank=rank, world_size=world)
    t = torch.full((4,), float(rank + 1), device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    dist.barrier()
    print("rank %d/%d backend %s all_reduce -> %s OK" % (rank, world, backend, t.tolist()), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank %d backend %s FAILED: %r" % (rank, backend, e), flush=True)
    sys.exit(1)
