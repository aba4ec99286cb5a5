import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, '.')
from molecular_dynamics_neural_operator_amd.rollout import gather_trajectories
torch.cuda.set_device(0); dev = torch.device('cuda', 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("init", time.perf_counter() - t0)
K, M, N = 20, 8, 504
x = torch.randn(K, M, N, 3, device=dev)
def tm(f, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); out.append((time.perf_counter() - t) * 1e3)
    return [round(v, 3) for v in out]
print("gather_trajectories", tm(lambda: gather_trajectories(x, M)))
recv = torch.empty((1, K, M, N, 3), device=dev)
print("all_gather_into_tensor", tm(lambda: dist.all_gather_into_tensor(recv.view(K, M, N, 3), x)))
print("barrier", tm(lambda: dist.barrier()))
t = torch.tensor([1.0], dtype=torch.float64, device=dev)
print("all_reduce", tm(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX)))
parts = [torch.randn(55, 4, N, 3, device=dev) for _ in range(2)]
print("cat traj", tm(lambda: torch.cat(parts, dim=1)))
print("empty", tm(lambda: torch.empty((K, M, N, 3), device=dev)))
dist.destroy_process_group()
