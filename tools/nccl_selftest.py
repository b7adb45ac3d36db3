import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(4,device="cuda"); dist.all_reduce(t); dist.barrier()
g=[torch.zeros(3,4,dtype=torch.int32,device="cuda")]
dist.gather(torch.ones(3,4,dtype=torch.int32,device="cuda"), g, dst=0)
print("nccl ok", t.tolist(), g[0].sum().item(), torch.cuda.nccl.version())
dist.destroy_process_group()
