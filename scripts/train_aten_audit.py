"""Which ATen operators (and device copies) still run inside one cfg4 training batch, and from which line.

  python scripts/train_aten_audit.py [--precision bf16] [--batches 3]

torch.profiler over a few train_epoch batches at the cfg4 shape (N = 28, batch 128, k = 1024, depth 6) with Python
stacks: per operator the calls per batch and the innermost frame inside this repository.  A measurement aid for
keeping the training step on libmdno's kernels (Weak 8) — nothing in the product imports it.
"""
import argparse
import collections
import sys
import tempfile
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from molecular_dynamics_neural_operator_amd import synthetic as syn  # noqa: E402
from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz  # noqa: E402
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss  # noqa: E402
from molecular_dynamics_neural_operator_amd.training import DeviceTrajectory, train_epoch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", choices=["fp32", "bf16"], default="bf16")
ap.add_argument("--batches", type=int, default=3)
ap.add_argument("--batch-size", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, W, B = 28, 10, a.batch_size
traj = syn.ou_trajectory(syn.chain_frame(N, seed=0), 1200, sigma=0.3, theta=0.1, seed=2)
cms = [syn.contact_map(f, 8.0) for f in traj]
with tempfile.TemporaryDirectory() as td:
    path = Path(td) / "t.npz"
    write_trajectory_npz(path, traj, cms, syn.amino_acids(N, seed=0))
    dset = ContactMapDataset(str(path), window_size=W, horizon=1)
dtraj = DeviceTrajectory(dset, dev)
torch.manual_seed(0)
model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
with torch.no_grad():
    for p_ in model.conv1.net.layers[4].parameters():
        p_.mul_(0.05)
model.to(dev)
model.train_precision = a.precision
try:
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, fused=True)
except (RuntimeError, TypeError):
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)
loss_fn = LpLoss(size_average=False)
idx = [list(range(s, s + B)) for s in range(0, (a.batches + 2) * B, B)]
train_epoch(model, (dtraj.batch(i) for i in idx[:2]), opt, loss_fn)          # warm-up
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_epoch(model, (dtraj.batch(i) for i in idx[2:2 + a.batches]), opt, loss_fn)
    torch.cuda.synchronize()

ops = collections.Counter()
where = collections.defaultdict(collections.Counter)
kernels = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or str(ev.device_type).endswith("CUDA"):
        kernels[ev.name[:90]] += 1
        continue
    if not ev.name.startswith("aten::"):
        continue
    ops[ev.name] += 1
    frame = "?"
    for fr in (ev.stack or []):
        if "molecular_dynamics_neural_operator_amd/" in fr:
            frame = fr[fr.index("molecular_dynamics_neural_operator_amd/"):]
            break
    where[ev.name][frame] += 1
print(f"precision {a.precision}, {a.batches} batches of {B}: ATen operators per batch (innermost repo frame)")
for name, n in ops.most_common():
    print(f"  {n / a.batches:7.1f}  {name}")
    for fr, c in where[name].most_common(6):
        print(f"           {c / a.batches:6.1f}  {fr}")
print("device activities per batch:")
for name, n in kernels.most_common():
    print(f"  {n / a.batches:7.1f}  {name}")
