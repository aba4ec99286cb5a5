import ctypes as C, sys, torch
sys.path.insert(0, '.')
from molecular_dynamics_neural_operator_amd import ops, _lib
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
dev = torch.device('cuda:0')
lib = _lib.load()
lib.mdno_debug_read.argtypes = [C.c_void_p]; lib.mdno_debug_read.restype = C.c_int
torch.manual_seed(0)
model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4).to(dev)
pack = model.param_pack(dev, conv_mode="materialized")
B, N, W = 128, 28, 10
frames = torch.randn(W, B, N, 3, device=dev)
aa = torch.randint(0, 20, (B * N,), device=dev)
x0 = ops.node_prologue(pack, frames, aa)
g0 = torch.randn_like(x0)
for _ in range(3): ops.node_prologue_bwd(pack, frames, aa, x0, g0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): ops.node_prologue_bwd(pack, frames, aa, x0, g0)
b.record(); torch.cuda.synchronize()
print("node_prologue_bwd (kernel + 4 reduces)", a.elapsed_time(b) / 10 * 1e3, "us")
out = (C.c_longlong * 16)()
lib.mdno_debug_read(out)
st = list(out)[:7]
print("stamps (cycles, 100 MHz counter?):", [st[i + 1] - st[i] for i in range(6)], "total", st[6] - st[0])
