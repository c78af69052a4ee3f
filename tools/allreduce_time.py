import os, time, torch, torch.distributed as dist
import sys; sys.path.insert(0, os.getcwd())
import bench
from phoenix_amd import parallel
rank=int(os.environ.get("RANK","0")); lr=int(os.environ.get("LOCAL_RANK","0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dev=torch.device("cuda", lr)
wl=bench.WORKLOADS["breast"]
net,y0,t=bench.make_problem(wl, dev, 0)
for p in net.parameters(): p.grad=torch.randn_like(p)
for i in range(5): parallel.allreduce_grads(net)
torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter()
    for i in range(50): parallel.allreduce_grads(net)
    torch.cuda.synchronize()
    print("allreduce_grads world=1: %.1f us per call"%((time.perf_counter()-t0)/50*1e6))
# host-side issue time only
t0=time.perf_counter()
for i in range(50): parallel.allreduce_grads(net)
print("issue: %.1f us"%((time.perf_counter()-t0)/50*1e6)); torch.cuda.synchronize()
dist.destroy_process_group()
