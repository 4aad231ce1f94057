"""Which torch pool streams run beside the default stream?  Ratio = (start of spin on A -> end of spin on B) / (one spin): 1 = concurrent,
2 = serialised (same hardware queue).  With D2S_FORCE_DIST=1 an RCCL process group (1 rank) is created and used first.  GPU box only."""
import os, sys
import torch
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if os.environ.get("D2S_FORCE_DIST") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.ones(1 << 20, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        w = dist.all_reduce(t, async_op=True); w.wait()
    torch.cuda.synchronize()
main = torch.cuda.current_stream()
cands = [torch.cuda.Stream() for _ in range(10)]
x = torch.zeros(8, device=dev)
for s in cands:                      # first use of a stream: lazy set-up, keep it out of the measurement
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
def ratio(a, b, cycles=4000000):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        ev[0].record(); torch.cuda._sleep(cycles); ev[1].record()
    with torch.cuda.stream(b):
        ev[2].record(); torch.cuda._sleep(cycles); ev[3].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[3]) / ev[0].elapsed_time(ev[1])
print("mode:", "RCCL group alive" if os.environ.get("D2S_FORCE_DIST") == "1" else "no process group")
for rep in range(2):
    print("  vs default stream:", " ".join(f"{ratio(main, s):.2f}" for s in cands))
print("  candidate 0 vs others:", " ".join(f"{ratio(cands[0], s):.2f}" for s in cands[1:]))
