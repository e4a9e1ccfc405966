"""Timing rehearsal of dist.sharded_count (chunked) on a one-rank RCCL group."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reflexiv_amd
from reflexiv_amd import dist as rd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rfx = reflexiv_amd.Reflexiv(0)
L,k,wpr,G=150,31,5,4_640_000
n_reads=int(5e9/L)//2*2
dg=torch.empty((G+31)//32,dtype=torch.int64,device="cuda"); dw=torch.empty(n_reads*wpr,dtype=torch.int64,device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(1,G,dg.data_ptr()); rfx.synth_reads_dev(1,dg.data_ptr(),G,0,n_reads,L,wpr,dw.data_ptr()); rfx.sync()
rfx.use_stream(torch.cuda.current_stream().cuda_stream)
reads=dict(words=dw,n_reads=n_reads,wpr=wpr,read_len=L,k=k)
eng=rd.HipEngine(rfx); eng.force_exchange=True
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    subs=eng.split_reads(reads,21)
    tb=te=0
    for sub in subs:
        torch.cuda.synchronize(); a=time.perf_counter()
        km,off=eng.bucket_by_owner(sub,1)
        torch.cuda.synchronize(); b=time.perf_counter()
        r,w=rd.exchange_by_owner_async(km,off,None,2)
        for x in w: x.wait()
        torch.cuda.synchronize(); c=time.perf_counter()
        tb+=b-a; te+=c-b
    t1=time.perf_counter()
    keys,counts,tot=rd.sharded_count(eng,reads,30,10_000_000,0,chunks=21)
    torch.cuda.synchronize(); t2=time.perf_counter()
    print("iter",it,"bucket %.1f ms exchange %.1f ms loop %.1f ms | sharded_count %.1f ms"%(tb*1e3,te*1e3,(t1-t0)*1e3,(t2-t1)*1e3), tot, flush=True)
dist.destroy_process_group()
