"""Sanity check of ops.pg_stream_shares_queue on a one-rank RCCL group: over torch's pool streams the answer must follow the 1-in-4 pattern of
the hardware-queue map (tools/stream_queue_probe.py), not be constant.  GPU box only."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dense2sparse-vit_amd"))
import torch, torch.distributed as dist
from d2s import ops
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cands = [torch.cuda.Stream() for _ in range(10)]
print("default stream shares the RCCL queue:", ops.pg_stream_shares_queue(torch.cuda.current_stream()))
print("pool streams:", [int(ops.pg_stream_shares_queue(s)) for s in cands])
print("again:       ", [int(ops.pg_stream_shares_queue(s)) for s in cands])
dist.destroy_process_group()
