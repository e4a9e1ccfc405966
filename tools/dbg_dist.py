"""One-rank RCCL rehearsal: (1) raw all_to_all_single corrupts per-peer messages above 1 GiB (RCCL 2.26 in this
torch build); (2) reflexiv_amd.dist._alltoallv (512 MiB rounds) returns them intact."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflexiv_amd import dist as rd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.Generator(device="cuda"); g.manual_seed(1)
src = torch.randint(-2**62, 2**62, ((1 << 28) + 12345,), dtype=torch.int64, device="cuda", generator=g)
for n in (1 << 27, (1 << 27) + 8, (1 << 28) + 12345):
    x = src[:n]
    r = torch.full_like(x, 7)
    dist.all_to_all_single(r, x, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    r2, _ = rd._alltoallv(x, [n])
    r3, works = rd._alltoallv(x, [n], async_op=True)
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    print("n=%d (%.3f GiB): raw intact %s | capped intact %s | capped async intact %s" %
          (n, n * 8 / 2**30, bool(torch.equal(r, x)), bool(torch.equal(r2, x)), bool(torch.equal(r3, x))), flush=True)
dist.destroy_process_group()
