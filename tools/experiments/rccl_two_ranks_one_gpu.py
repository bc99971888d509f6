"""Can RCCL run two ranks on ONE device (so that the N > 1 exchange could be tested on a 1-GPU box)?
Launched as: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 <this>"""
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    x = torch.full((4,), float(rank), device="cuda:0")
    out = [torch.empty_like(x) for _ in range(2)]
    dist.all_gather(out, x)
    torch.cuda.synchronize()
    print("rank", rank, "all_gather ok", [o.tolist() for o in out])
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", type(e).__name__, str(e).splitlines()[0][:300])
    sys.exit(0)
