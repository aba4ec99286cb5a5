import sys, copy, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_training import _replica_loss, rel_err
from molecular_dynamics_neural_operator_amd import synthetic as syn
from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
from molecular_dynamics_neural_operator_amd.training import collate
from oracle import graph_kernel_oracle as O
dev = torch.device('cuda:0')
base = syn.chain_frame(28, seed=0)
traj = syn.ou_trajectory(base, 60, sigma=0.15, theta=0.2, seed=2)
write_trajectory_npz('/tmp/t.npz', traj, [syn.contact_map(f, 8.0) for f in traj], syn.amino_acids(28, seed=0))
dset = ContactMapDataset('/tmp/t.npz', window_size=10, horizon=1)
for (k, depth, B) in ((128, 2, 3), (1024, 2, 4), (1024, 6, 4), (1024, 6, 16)):
    sub = [dset[i] for i in range(B)]
    torch.manual_seed(3)
    model = KernelNN(64, k, depth, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters(): p_.mul_(0.2)
    model.to(dev).train()
    y = torch.cat([s.y for s in sub]).to(dev)
    res = {}
    for prec in ("fp32", "bf16"):
        model.train_precision = prec
        model.zero_grad()
        out = model(sub)
        loss = LpLoss(size_average=False)(out.view(B, -1), y.view(B, -1)); loss.backward()
        res[prec] = {n: p_.grad.clone() for n, p_ in model.named_parameters()}
    want_loss, want_out, want = _replica_loss(model, O, collate(sub), B)
    e32 = {n: rel_err(res["fp32"][n], want[n]) for n in want}
    e16 = {n: rel_err(res["bf16"][n], want[n]) for n in want}
    print(f"k={k} depth={depth} B={B}: fp32 max {max(e32.values()):.1e}; bf16:", {n.replace('conv1.net.layers','L').replace('.weight','.w').replace('.bias','.b'): f"{v:.1e}" for n, v in e16.items()})
