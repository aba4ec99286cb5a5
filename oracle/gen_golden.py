#!/usr/bin/env python3
"""Golden-vector generator — TEST INFRASTRUCTURE, runs only in the build container.

Imports the reference's own ``graph_kernel.py`` / ``dataset.py`` from ``/root/reference``
*unchanged* (third-party packages it needs that are absent from the image come from
``oracle/_stubs``, see its README), runs the reference code on seeded synthetic inputs on the
CPU and writes inputs + expected outputs as small ``.npz`` fixtures under ``tests/golden/``.

Two things are patched at run time, nothing in the reference is edited:
  * ``graph_kernel.args`` — the module-global the reference's ``KernelNN.forward`` reads
    (graph_kernel.py:279-281) only exists under ``__main__``; we set it to a Namespace.
  * ``torch.Tensor.cuda`` — ``forward`` hard-codes ``.cuda()`` for the LSTM state
    (graph_kernel.py:281-282); on this GPU-less container it is made a no-op.

The reference itself never travels: only the data written here is committed.

Usage:  python oracle/gen_golden.py [--out tests/golden] [--skip-large] [--only NAME[,NAME]]
        (sections: base, live28, live504, checkpoint, train; every section seeds itself, so any subset
        reproduces the same files)
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REFERENCE = Path(os.environ.get("MDNO_REFERENCE", "/root/reference"))


def _import_reference():
    if not (REFERENCE / "graph_kernel.py").exists():
        raise SystemExit(f"reference not found at {REFERENCE} (this script only runs in the build container)")
    sys.path.insert(0, str(Path(__file__).resolve().parent / "_stubs"))
    sys.path.insert(0, str(REFERENCE))
    sys.path.insert(0, str(REPO))
    import matplotlib
    matplotlib.use("Agg")
    import graph_kernel as ref_gk  # noqa: the reference, unchanged
    import dataset as ref_ds       # noqa
    torch.Tensor.cuda = lambda self, *a, **k: self  # CPU container: neutralise the hard-coded .cuda()
    return ref_gk, ref_ds


def _sd_np(sd, prefix="p."):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def _checksums(sd):
    names = sorted(sd.keys())
    sums = np.array([float(sd[k].double().sum()) for k in names], dtype=np.float64)
    asums = np.array([float(sd[k].double().abs().sum()) for k in names], dtype=np.float64)
    return np.array(names), sums, asums


def _flat_contact_map(pd):
    return np.concatenate([pd.edge_index[0].numpy(), pd.edge_index[1].numpy()]).astype(np.int64)


def _cm_checksum(cm):
    """Order-sensitive 64-bit checksum of a flat [rows..., cols...] contact map (wraps mod 2^64)."""
    c = np.asarray(cm, dtype=np.uint64)
    w = (np.arange(c.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return np.uint64((c * w).sum())


def gen_live504(gk, ds, out: Path):
    """Width-64 / k=1024 / depth-6 model at N=504 with bounded, live activations (the near-identity
    weight set with a kernel integral strong enough to move atoms ~0.4 A per step): 5 teacher-forced
    forwards on the reference's own ContactMapDataset samples and 5 free-running steps through the
    reference's recursive_propagation (graph_kernel.py:396-413).  ~10 reference forwards at 7.6 TFLOP."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    THR, N, W, NTF, NFREE = 8.0, 504, 10, 5, 5
    T = W + NTF + 1
    base = syn.box_frame(N, seed=1)
    traj = syn.ou_trajectory(base, T, sigma=0.15, theta=0.2, seed=4)          # [T,N,3]
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    gk.args = Namespace(window_size=W, num_residues=N, batch_size=1)
    cms = np.empty(T, dtype=object)
    for t in range(T):
        cms[t] = _flat_contact_map(gk.construct_pairdata(traj[t:t + 1], aa, threshold=THR))
    gains = dict(seed=0, kernel_gain=0.02, feature_gain=0.1, kernel_to_coords=1.0)
    sd = near_identity_state_dict(64, 1024, **gains)
    model = gk.KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    print("live504 load:", model.load_state_dict(sd))
    model.eval()
    names, sums, asums = _checksums(model.state_dict())
    with tempfile.TemporaryDirectory() as td:
        h5 = Path(td) / "synthetic504.h5"
        with open(h5, "wb") as fh:
            np.savez(fh, contact_map=cms, point_cloud=np.transpose(traj, (0, 2, 1)).copy(),
                     rmsd=np.zeros(T, np.float32), amino_acids=aa.numpy())
        dset = ds.ContactMapDataset(str(h5), window_size=W, horizon=1, node_feature_dset_path=str(h5))
        tf_out, tf_lat0 = [], None
        with torch.no_grad():
            for i in range(NTF):
                o, lat = model(dset[i], return_latent=True)
                tf_out.append(o.numpy())
                if i == 0:
                    tf_lat0 = lat.numpy()
                print(f"  live504 teacher-forced {i}: |out-y| mean {float((o - dset[i].y).norm(dim=1).mean()):.3f}", flush=True)
        holder = Namespace(module=model, eval=lambda: None)
        fc = gk.recursive_propagation(holder, dset, "cpu", num_steps=NFREE, starting_points=[0], threshold=THR)
    free = np.stack([f.x_position[-1].numpy() for f in fc])
    free_E = np.array([f.edge_index.shape[1] for f in fc])
    free_gap = np.array([syn.min_threshold_gap(f.x_position[-1].numpy(), THR) for f in fc])
    np.savez_compressed(
        out / "kernelnn_live504.npz",
        ctor=np.array([64, 1024, 6, 6, 7, 3, 20, 4]), threshold=THR, window=W,
        weight_gains=np.array([gains["seed"], gains["kernel_gain"], gains["feature_gain"], gains["kernel_to_coords"]]),
        param_names=names, param_sum=sums, param_abs_sum=asums,
        frames=traj, amino_acids=aa.numpy(),
        contact_map_len=np.array([c.size for c in cms]), contact_map_checksum=np.array([_cm_checksum(c) for c in cms]),
        teacher_forced_out=np.stack(tf_out), teacher_forced_latent0=tf_lat0,
        free_frames=free, free_num_edges=free_E, free_min_gap=free_gap,
        free_edge_checksum_last=_cm_checksum(_flat_contact_map(fc[-1])),
    )
    zf = float((tf_lat0 == 0).mean())
    print(f"kernelnn_live504: ok; E0 {cms[0].size // 2}, free-run E {free_E}, min gap {free_gap}, "
          f"latent zeros {zf:.2f}, |latent| max {np.abs(tf_lat0).max():.2f}")


def gen_live28(gk, ds, out: Path):
    """The reference's own BBA shape (N=28 C-alpha chain, nb:1034; the size of its one published number,
    nb:370) at the CLI-default model size (width 64 / k=1024 / depth 6, graph_kernel.py:528-537) with the
    live near-identity weight set of `gen_live504`: 20 teacher-forced forwards on the reference's own
    ContactMapDataset samples (latent of the first) and 20 free-running steps through the reference's
    recursive_propagation (graph_kernel.py:396-413), with the edge count and the smallest distance-to-cutoff
    gap of every produced frame.  ~40 reference forwards of 66 ms."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    THR, N, W, NTF, NFREE = 8.0, 28, 10, 20, 20
    T = W + NTF + 1
    base = syn.chain_frame(N, seed=0)
    traj = syn.ou_trajectory(base, T, sigma=0.15, theta=0.2, seed=6)          # [T,N,3]
    aa = torch.from_numpy(syn.amino_acids(N, seed=0))
    gk.args = Namespace(window_size=W, num_residues=N, batch_size=1)
    cms = np.empty(T, dtype=object)
    for t in range(T):
        cms[t] = _flat_contact_map(gk.construct_pairdata(traj[t:t + 1], aa, threshold=THR))
    gains = dict(seed=0, kernel_gain=0.02, feature_gain=0.1, kernel_to_coords=1.0)
    sd = near_identity_state_dict(64, 1024, **gains)
    model = gk.KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    print("live28 load:", model.load_state_dict(sd))
    model.eval()
    names, sums, asums = _checksums(model.state_dict())
    with tempfile.TemporaryDirectory() as td:
        h5 = Path(td) / "synthetic28.h5"
        with open(h5, "wb") as fh:
            np.savez(fh, contact_map=cms, point_cloud=np.transpose(traj, (0, 2, 1)).copy(),
                     rmsd=np.zeros(T, np.float32), amino_acids=aa.numpy())
        dset = ds.ContactMapDataset(str(h5), window_size=W, horizon=1, node_feature_dset_path=str(h5))
        tf_out, tf_lat0 = [], None
        with torch.no_grad():
            for i in range(NTF):
                o, lat = model(dset[i], return_latent=True)
                tf_out.append(o.numpy())
                if i == 0:
                    tf_lat0 = lat.numpy()
        holder = Namespace(module=model, eval=lambda: None)
        fc = gk.recursive_propagation(holder, dset, "cpu", num_steps=NFREE, starting_points=[0], threshold=THR)
    free = np.stack([f.x_position[-1].numpy() for f in fc])
    free_E = np.array([f.edge_index.shape[1] for f in fc])
    free_gap = np.array([syn.min_threshold_gap(f.x_position[-1].numpy(), THR) for f in fc])
    free_cm = np.array([_cm_checksum(_flat_contact_map(f)) for f in fc])
    np.savez_compressed(
        out / "kernelnn_live28.npz",
        ctor=np.array([64, 1024, 6, 6, 7, 3, 20, 4]), threshold=THR, window=W,
        weight_gains=np.array([gains["seed"], gains["kernel_gain"], gains["feature_gain"], gains["kernel_to_coords"]]),
        param_names=names, param_sum=sums, param_abs_sum=asums,
        frames=traj, amino_acids=aa.numpy(),
        contact_map_len=np.array([c.size for c in cms]), contact_map_checksum=np.array([_cm_checksum(c) for c in cms]),
        teacher_forced_out=np.stack(tf_out), teacher_forced_latent0=tf_lat0,
        free_frames=free, free_num_edges=free_E, free_min_gap=free_gap, free_edge_checksum=free_cm,
    )
    zf = float((tf_lat0 == 0).mean())
    step = float(np.linalg.norm(free[1:] - free[:-1], axis=-1).mean())
    print(f"kernelnn_live28: ok; E0 {cms[0].size // 2}, free-run E {free_E}, min gap {free_gap.min():.2e}, "
          f"latent zeros {zf:.2f}, |latent| max {np.abs(tf_lat0).max():.2f}, mean |step| {step:.3f} A")


def gen_checkpoint(gk, out: Path):
    """A `best.pt`-shaped checkpoint (graph_kernel.py:630-639) written from the reference's own
    KernelNN wrapped as `main` wraps it (DataParallel -> `module.` key prefix, :528), stored as
    plain arrays, plus the reference forward it must reproduce once loaded."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    THR, N, W = 8.0, 28, 10
    win = syn.jitter_window(syn.chain_frame(N, seed=0), W, seed=0)
    aa = torch.from_numpy(syn.amino_acids(N, seed=0))
    gk.args = Namespace(window_size=W, num_residues=N, batch_size=1)
    pd = gk.construct_pairdata(win, aa, threshold=THR)
    torch.manual_seed(77)
    model = gk.KernelNN(64, 32, 2, 6, 7, 3, 20, 4)
    wrapped = torch.nn.DataParallel(model)               # CPU: only the `module.` prefix matters here
    optimizer = torch.optim.Adam(wrapped.parameters(), lr=0.01, weight_decay=5e-4)      # :541-543
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=50, gamma=0.8)     # :544-546
    ckpt = {"epoch": 3, "model_state_dict": wrapped.state_dict(),
            "optimizer_state_dict": optimizer.state_dict(), "scheduler_state_dict": scheduler.state_dict()}
    assert all(k.startswith("module.") for k in ckpt["model_state_dict"])
    model.eval()
    with torch.no_grad():
        out_ref = model(pd).numpy()
    arrays = {"msd/" + k: v.detach().numpy() for k, v in ckpt["model_state_dict"].items()}
    np.savez_compressed(
        out / "checkpoint_best_pt.npz", epoch=ckpt["epoch"], ctor=np.array([64, 32, 2, 6, 7, 3, 20, 4]),
        scheduler_step_size=50, scheduler_gamma=0.8, optimizer_lr=0.01, optimizer_weight_decay=5e-4,
        x_position=win, x_aminoacid=aa.numpy(), edge_index=pd.edge_index.numpy(), edge_attr=pd.edge_attr.numpy(),
        out=out_ref, **arrays)
    print("checkpoint_best_pt: ok;", len(arrays), "tensors, |out| max", float(np.abs(out_ref).max()))


def gen_train_step(gk, ds, out: Path):
    """One iteration of the reference's own `train()` (graph_kernel.py:445-474) at batch size 1 — the only
    batch size at which its forward is well defined without torch_geometric's DataParallel collation
    (SURVEY.md §3.3): zero_grad, forward, `LpLoss(size_average=False)` (:547), backward, Adam step
    (lr 0.01, weight_decay 5e-4, :541-543).  Stored: the sample, the loss `train()` returns, every
    parameter's gradient as left in `.grad`, and the parameters after the step.  Two sizes: width 8 /
    k 16 (weights stored) and width 64 / k 128 (the HIP training kernels' width; weights regenerated from
    the seed by the test, checksums stored)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    THR, N, W = 8.0, 28, 10
    base = syn.chain_frame(N, seed=0)
    T = 40
    traj = syn.ou_trajectory(base, T, sigma=0.15, theta=0.2, seed=2)
    aa = torch.from_numpy(syn.amino_acids(N, seed=0))
    gk.args = Namespace(window_size=W, num_residues=N, batch_size=1)
    cms = np.empty(T, dtype=object)
    for t in range(T):
        cms[t] = _flat_contact_map(gk.construct_pairdata(traj[t:t + 1], aa, threshold=THR))

    class OneSampleBatch(torch.nn.Module):      # stands in for DataParallel's collation of a 1-element list
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, batch):
            assert len(batch) == 1
            return self.module(batch[0])

    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        h5 = Path(td) / "synthetic.h5"
        with open(h5, "wb") as fh:
            np.savez(fh, contact_map=cms, point_cloud=np.transpose(traj, (0, 2, 1)).copy(),
                     rmsd=np.zeros(T, np.float32), amino_acids=aa.numpy())
        dset = ds.ContactMapDataset(str(h5), window_size=W, horizon=1, node_feature_dset_path=str(h5))
        idx = 5
        sample = dset[idx]
        for tag, ctor, seed in (("s8", (8, 16, 2, 6, 7, 3, 20, 4), 21), ("w64", (64, 128, 2, 6, 7, 3, 20, 4), 31)):
            torch.manual_seed(seed)
            model = gk.KernelNN(*ctor)
            with torch.no_grad():                  # keep activations O(1) through the random-init layers
                for p_ in model.conv1.net.layers[4].parameters():
                    p_.mul_(0.2)
            before = {k: v.detach().clone() for k, v in model.state_dict().items()}
            wrapped = OneSampleBatch(model)
            optimizer = torch.optim.Adam(wrapped.parameters(), lr=0.01, weight_decay=5e-4)
            avg_loss, avg_mse = gk.train(wrapped, [[sample]], optimizer, gk.LpLoss(size_average=False), "cpu")
            grads = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters()}
            after = {k: v.detach().clone() for k, v in model.state_dict().items()}
            names, sums, asums = _checksums(before)
            arrays.update({f"{tag}.ctor": np.array(ctor), f"{tag}.seed": seed, f"{tag}.last_layer_scale": 0.2,
                           f"{tag}.loss": avg_loss, f"{tag}.mse": avg_mse,
                           f"{tag}.param_names": names, f"{tag}.param_sum": sums, f"{tag}.param_abs_sum": asums})
            arrays.update(_sd_np(grads, f"{tag}.g."))
            if tag == "s8":
                arrays.update(_sd_np(before, f"{tag}.p."))
                arrays.update(_sd_np(after, f"{tag}.a."))
            else:       # after-step values of the small tensors only (the wide ones follow from their gradients)
                arrays.update(_sd_np({k: v for k, v in after.items() if v.numel() <= 4096}, f"{tag}.a."))
            print(f"train_step {tag}: loss {avg_loss:.6f} mse {avg_mse:.6f}; |grad| max "
                  f"{max(float(g.abs().max()) for g in grads.values()):.3e}")
    np.savez_compressed(
        out / "train_step_b1.npz", sample_index=idx, window=W, threshold=THR, lr=0.01, weight_decay=5e-4,
        point_cloud=np.transpose(traj, (0, 2, 1)).copy(), contact_map=cms, amino_acids=aa.numpy(),
        rmsd=np.zeros(T, np.float32),
        x_position=sample.x_position.numpy(), x_aminoacid=sample.x_aminoacid.numpy(), y=sample.y.numpy(),
        edge_index=sample.edge_index.numpy(), edge_attr=sample.edge_attr.numpy(), **arrays)
    print("train_step_b1: ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", type=Path, default=REPO / "tests" / "golden")
    ap.add_argument("--skip-large", action="store_true", help="skip the N=504 full-size forwards (minutes of CPU)")
    ap.add_argument("--only", default="", help="comma-separated sections: base, live28, live504, checkpoint, train (default: all)")
    a = ap.parse_args()
    a.out.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    only = set(filter(None, a.only.split(",")))

    gk, ds = _import_reference()
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    if not only or "checkpoint" in only:
        gen_checkpoint(gk, a.out)
    if not only or "train" in only:
        gen_train_step(gk, ds, a.out)
    if not only or "live28" in only:
        gen_live28(gk, ds, a.out)
    if (not only and not a.skip_large) or "live504" in only:
        gen_live504(gk, ds, a.out)
    if only and "base" not in only:
        return

    THR = 8.0
    N, W = 28, 10
    base = syn.chain_frame(N, seed=0)
    window = syn.jitter_window(base, W, seed=0)           # [W,N,3]
    aa = torch.from_numpy(syn.amino_acids(N, seed=0))
    gap = syn.min_threshold_gap(window[-1], THR)
    assert gap > 1e-4, f"fixture frame has a pair within {gap} A of the threshold"

    # ---- (4) pairdata_graph: positions -> edge_index / edge_attr  (graph_kernel.py:362-393)
    gk.args = Namespace(window_size=W, num_residues=N, batch_size=1)
    pd_ref = gk.construct_pairdata(window, aa, threshold=THR)
    box = syn.box_frame(120, seed=11)[None]               # a denser, unordered cloud as a second case
    assert syn.min_threshold_gap(box[-1], THR) > 1e-5
    pd_box = gk.construct_pairdata(box, torch.zeros(120, dtype=torch.long), threshold=THR)
    np.savez_compressed(
        a.out / "pairdata_graph.npz",
        threshold=THR,
        x_position=window, edge_index=pd_ref.edge_index.numpy(), edge_attr=pd_ref.edge_attr.numpy(),
        box_position=box, box_edge_index=pd_box.edge_index.numpy(), box_edge_attr=pd_box.edge_attr.numpy(),
        min_gap=gap,
    )
    print("pairdata_graph: E =", pd_ref.edge_index.shape[1], " box E =", pd_box.edge_index.shape[1])

    # ---- (1) nnconv_small: one NNConv_old application, Cin=Cout=8, k=16  (graph_kernel.py:125-214)
    torch.manual_seed(1234)
    cin = cout = 8
    kw = 16
    for aggr in ("mean", "add"):
        net = gk.DenseNet([6, kw, kw, cin * cout], torch.nn.ReLU)
        conv = gk.NNConv_old(cin, cout, net, aggr=aggr)
        x = torch.randn(N, cin)
        with torch.no_grad():
            y = conv(x, pd_ref.edge_index, pd_ref.edge_attr)
            w_e = net(pd_ref.edge_attr)
        np.savez_compressed(
            a.out / f"nnconv_small_{aggr}.npz",
            x=x.numpy(), edge_index=pd_ref.edge_index.numpy(), edge_attr=pd_ref.edge_attr.numpy(),
            y=y.numpy(), w_e=w_e.numpy(), aggr=aggr, **_sd_np(conv.state_dict()),
        )
    print("nnconv_small: ok")

    # ---- (2) kernelnn_small: whole forward at width 8, ker_width 16, depth 2  (graph_kernel.py:245-309)
    torch.manual_seed(4321)
    model_s = gk.KernelNN(8, 16, 2, 6, 7, 3, 20, 4)
    model_s.eval()
    with torch.no_grad():
        out_s, lat_s = model_s(pd_ref, return_latent=True)
    np.savez_compressed(
        a.out / "kernelnn_small.npz",
        ctor=np.array([8, 16, 2, 6, 7, 3, 20, 4]),
        x_position=window, x_aminoacid=aa.numpy(),
        edge_index=pd_ref.edge_index.numpy(), edge_attr=pd_ref.edge_attr.numpy(),
        out=out_s.numpy(), latent=lat_s.numpy(), **_sd_np(model_s.state_dict()),
    )
    print("kernelnn_small: ok", out_s.shape, lat_s.shape)

    # ---- (5) rollout_20: free-running + teacher-forced steps at small dims (graph_kernel.py:396-413)
    # dataset in the reference's on-disk layout (dataset.py:112-127, 159, 189), npz twin of the HDF5 file
    T = 40
    traj = syn.ou_trajectory(base, T, sigma=0.15, theta=0.2, seed=2)          # [T,N,3]
    cms = np.empty(T, dtype=object)
    for t in range(T):
        pd_t = gk.construct_pairdata(traj[t:t + 1], aa, threshold=THR)
        cms[t] = np.concatenate([pd_t.edge_index[0].numpy(), pd_t.edge_index[1].numpy()]).astype(np.int64)
    with tempfile.TemporaryDirectory() as td:
        h5 = Path(td) / "synthetic.h5"          # npz bytes under the name the reference's loader dispatches on
        with open(h5, "wb") as fh:
            np.savez(fh, contact_map=cms, point_cloud=np.transpose(traj, (0, 2, 1)).copy(),
                     rmsd=np.linspace(0, 1, T).astype(np.float32), amino_acids=aa.numpy())
        dset = ds.ContactMapDataset(str(h5), window_size=W, horizon=1, node_feature_dset_path=str(h5))
        assert len(dset) == T - W - 1 + 1
        samp = dset[3]
        # untrained weights collapse every atom onto one point (E = N^2 after one step), which would pin
        # only the trivial graph: the free run uses the repo's near-identity weight set (weights.py),
        # loaded into the reference's own KernelNN, so the graph changes gradually over the 20 steps.
        from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
        model_r = gk.KernelNN(8, 16, 2, 6, 7, 3, 20, 4)
        print("load near-identity:", model_r.load_state_dict(
            near_identity_state_dict(8, 16, seed=3, kernel_gain=1e-2, feature_gain=1e-1, kernel_to_coords=1.0)))
        model_r.eval()
        holder = Namespace(module=model_r, eval=lambda: None)
        fc = gk.recursive_propagation(holder, dset, "cpu", num_steps=20, starting_points=[0], threshold=THR)
        free = np.stack([f.x_position[-1].numpy() for f in fc])               # [20,N,3]
        free_E = np.array([f.edge_index.shape[1] for f in fc])
        free_gap = np.array([syn.min_threshold_gap(f.x_position[-1].numpy(), THR) for f in fc])
        free_ei_last = fc[-1].edge_index.numpy()
        # teacher-forced: forward on dataset[i] for i in 0..19
        tf = []
        with torch.no_grad():
            for i in range(20):
                tf.append(model_s(dset[i]).numpy())
        tf = np.stack(tf)
        tf_y = np.stack([dset[i].y.numpy() for i in range(20)])
    np.savez_compressed(
        a.out / "rollout_20.npz",
        threshold=THR, window=W, horizon=1,
        point_cloud=np.transpose(traj, (0, 2, 1)).copy(), contact_map=cms, amino_acids=aa.numpy(),
        rmsd=np.linspace(0, 1, T).astype(np.float32),
        sample3_x_position=samp.x_position.numpy(), sample3_y=samp.y.numpy(),
        sample3_edge_index=samp.edge_index.numpy(), sample3_edge_attr=samp.edge_attr.numpy(),
        dataset_len=len(dset),
        free_frames=free, free_num_edges=free_E, free_min_gap=free_gap, free_edge_index_last=free_ei_last,
        teacher_forced_out=tf, teacher_forced_y=tf_y,
        **_sd_np(model_s.state_dict(), "tf."), **_sd_np(model_r.state_dict(), "free."),
    )
    print("rollout_20: ok; free-run E per step:", free_E, " min gap", free_gap.min())

    # ---- LpLoss (graph_kernel.py:75-122)
    torch.manual_seed(5)
    lx, ly = torch.randn(4, 84), torch.randn(4, 84)
    np.savez_compressed(
        a.out / "lploss.npz", x=lx.numpy(), y=ly.numpy(),
        rel_sum=gk.LpLoss(size_average=False)(lx, ly).numpy(),
        rel_mean=gk.LpLoss(size_average=True)(lx, ly).numpy(),
        rel_none=gk.LpLoss(reduction=False)(lx, ly).numpy(),
        abs_mean=gk.LpLoss().abs(lx, ly).numpy(),
    )

    # ---- (3) kernelnn_full_seeded: CLI defaults (w=64, k=1024, depth 6), torch.manual_seed(0) init
    torch.manual_seed(0)
    model_f = gk.KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model_f.eval()
    names, sums, asums = _checksums(model_f.state_dict())
    with torch.no_grad():
        out_f, lat_f = model_f(pd_ref, return_latent=True)
        we_f = model_f.conv1.net(pd_ref.edge_attr)
    np.savez_compressed(
        a.out / "kernelnn_full_seeded.npz",
        ctor=np.array([64, 1024, 6, 6, 7, 3, 20, 4]), seed=0,
        x_position=window, x_aminoacid=aa.numpy(),
        edge_index=pd_ref.edge_index.numpy(), edge_attr=pd_ref.edge_attr.numpy(),
        out=out_f.numpy(), latent=lat_f.numpy(),
        w_e_checksum=np.array([float(we_f.double().sum()), float(we_f.double().abs().sum())]),
        w_e_first_edge=we_f[0].numpy(),
        param_names=names, param_sum=sums, param_abs_sum=asums,
    )
    print("kernelnn_full_seeded: ok; |out| max", float(out_f.abs().max()))

    # ---- shape B: N=504 all-atom stand-in, full-size model, reference CPU forward (slow: 7.5 TFLOP)
    if not a.skip_large:
        NB = 504
        base_b = syn.box_frame(NB, seed=1)
        win_b = syn.jitter_window(base_b, W, seed=1)
        aa_b = torch.from_numpy(syn.amino_acids(NB, seed=1))
        gk.args = Namespace(window_size=W, num_residues=NB, batch_size=1)
        pd_b = gk.construct_pairdata(win_b, aa_b, threshold=THR)
        print("shapeB: E =", pd_b.edge_index.shape[1], "gap", syn.min_threshold_gap(win_b[-1], THR))
        with torch.no_grad():
            out_b, lat_b = model_f(pd_b, return_latent=True)
        np.savez_compressed(
            a.out / "kernelnn_shapeB_seeded.npz",
            ctor=np.array([64, 1024, 6, 6, 7, 3, 20, 4]), seed=0, threshold=THR,
            x_position=win_b, x_aminoacid=aa_b.numpy(), num_edges=pd_b.edge_index.shape[1],
            min_gap=syn.min_threshold_gap(win_b[-1], THR),
            out=out_b.numpy(), latent=lat_b.numpy(),
            param_names=names, param_sum=sums, param_abs_sum=asums,
        )
        print("kernelnn_shapeB_seeded: ok; |out| max", float(out_b.abs().max()))


if __name__ == "__main__":
    main()
