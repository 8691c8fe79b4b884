import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
buf=torch.zeros(4_700_000, device=dev)
comm=torch.cuda.Stream()
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); works=[]
    for _ in range(10):
        with torch.cuda.stream(comm):
            works.append(dist.all_reduce(buf, async_op=True))
    t1=time.perf_counter()
    for w in works: w.wait()
    torch.cuda.synchronize(); t2=time.perf_counter()
    print("host issue %.1f us per all_reduce, total with drain %.1f us per"%((t1-t0)/10*1e6,(t2-t0)/10*1e6))
dist.destroy_process_group()
