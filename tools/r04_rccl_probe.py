import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(4,device="cuda",dtype=torch.float64)*3
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(device_ids=[0])
g=[torch.zeros(4,device="cuda",dtype=torch.float64)]
dist.gather(t, g, dst=0)
print("rccl one-rank group ok", t.tolist(), g[0].tolist())
dist.destroy_process_group()
