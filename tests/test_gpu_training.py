"""Training path (cfg4): every backward op against torch autograd of the oracle's formulas, and the
whole differentiable forward+backward against an fp64 CPU replica of the model built from the
oracle's ops.  Runs the HIP kernels through the C ABI on cuda:0."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, write_golden_trajectory

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    from molecular_dynamics_neural_operator_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import graph_kernel_oracle
    return graph_kernel_oracle


def test_linear_atb_colsum_transpose_relu(dev):
    from molecular_dynamics_neural_operator_amd import ops
    g = torch.Generator().manual_seed(0)
    for rows, n, k in ((300, 128, 64), (517, 256, 1024), (260, 4096, 128), (100, 24, 6), (77, 6, 40)):
        a, w, b = torch.randn(rows, k, generator=g), torch.randn(n, k, generator=g), torch.randn(n, generator=g)
        for mode in ("f32", "split_bf16"):     # split_bf16: bf16 matrix pipe where (n, k) tile, fp32 kernels otherwise
            for relu in (False, True):
                want = F.linear(a.double(), w.double(), b.double())
                want = want.relu() if relu else want
                assert rel_err(ops.linear(a.to(dev), w.to(dev), b.to(dev), relu, gemm_mode=mode), want) < 2e-6
            assert rel_err(ops.linear(a.to(dev), w.to(dev), None, gemm_mode=mode), F.linear(a.double(), w.double())) < 2e-6
    for rows, n1, n2 in ((5000, 128, 256), (333, 1024, 128), (4097, 256, 6), (50, 7, 3)):
        a, b = torch.randn(rows, n1, generator=g), torch.randn(rows, n2, generator=g)
        got = ops.gemm_atb(a.to(dev), b.to(dev))
        assert rel_err(got, a.double().t() @ b.double()) < 2e-6
        assert torch.equal(got, ops.gemm_atb(a.to(dev), b.to(dev)))          # fixed-order partials: reproducible
        assert rel_err(ops.colsum(a.to(dev)), a.double().sum(0)) < 2e-6
    a = torch.randn(70, 45, generator=g)
    assert torch.equal(ops.transpose(a.to(dev)).cpu(), a.t().contiguous())
    gq, y, sc = torch.randn(40, 64, generator=g), torch.randn(40, 64, generator=g), torch.rand(40, generator=g)
    assert torch.equal(ops.relu_bwd(gq.to(dev), y.to(dev)).cpu(), gq * (y > 0))
    torch.testing.assert_close(ops.relu_bwd(gq.to(dev), y.to(dev), sc.to(dev)).cpu(), gq * (y > 0) * sc[:, None])
    gz, gs = torch.empty(40, 64, device=dev), torch.empty(40, 64, device=dev)       # both outputs in one pass
    ops.relu_bwd2(gq.to(dev), y.to(dev), sc.to(dev), gz, gs)
    assert torch.equal(gz, ops.relu_bwd(gq.to(dev), y.to(dev))) and torch.equal(gs, ops.relu_bwd(gq.to(dev), y.to(dev), sc.to(dev)))


def test_nnconv_backward_ops_vs_autograd(dev, O):
    """One conv application on an irregular graph (hub, isolated node, duplicate edges): input, root,
    bias and per-edge-weight gradients of mean(x_j . W_e) + x.root + bias vs torch autograd (fp64)."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(5)
    n, E = 90, 1500
    ei = torch.randint(0, n, (2, E), generator=gen)
    ei[1, :220] = 11
    ei[1, ei[1] == 30] = 31
    x = torch.randn(n, 64, generator=gen).double().requires_grad_()
    w_e = (torch.randn(E, 4096, generator=gen) * 0.1).double().requires_grad_()
    root = (torch.randn(64, 64, generator=gen) * 0.1).double().requires_grad_()
    bias = torch.randn(64, generator=gen).double().requires_grad_()
    y = torch.relu(O.nnconv_apply(x, ei, w_e, root, bias, "mean"))
    gy = torch.randn(n, 64, generator=gen).double()
    y.backward(gy)
    g = ops.coo_to_csr(ei.to(dev), n)
    perm = g.perm[:E].long()
    w_csr = w_e.detach().float().to(dev)[perm].contiguous()
    xd = x.detach().float().to(dev)
    yd = ops.nnconv(xd, g, w_csr, root.detach().float().to(dev), bias.detach().float().to(dev), "mean", relu=True)
    inv = ops.inv_degree(g, "mean")
    gz = ops.relu_bwd(gy.float().to(dev), yd)
    gs = ops.relu_bwd(gy.float().to(dev), yd, inv)
    by_src = ops.source_sorted(g, n)
    gx = ops.nnconv_bwd_x(gz, gs, by_src, w_csr, root.detach().float().to(dev))
    assert rel_err(gx, x.grad) < 1e-5
    d_root, d_bias = ops.nnconv_bwd_root(xd, gz)
    assert rel_err(d_root, root.grad) < 1e-5 and rel_err(d_bias, bias.grad) < 1e-5
    d_we = ops.nnconv_bwd_we(xd.unsqueeze(0), gs.unsqueeze(0), g)
    want = w_e.grad[perm.cpu()]
    assert rel_err(d_we, want) < 1e-5


@pytest.mark.parametrize("bf16", [False, True])
def test_conv_chain_entries_equal_the_single_ops(dev, bf16):
    """mdno_nnconv_chain_*: the 2*depth applications of a step in one call each way.  Forward == 2*depth single conv
    calls, bitwise; backward (every ReLU backward below the top one fused into the input-gradient kernel above it)
    == 2*depth (relu_bwd2, nnconv_bwd_x) pairs, bitwise — on an irregular batch graph (hub, isolated node)."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(11)
    R, E, depth = 300, 4000, 3
    L = 2 * depth
    ei = torch.randint(0, R, (2, E), generator=gen)
    ei[1, :300] = 17
    ei[1, ei[1] == 40] = 41
    g = ops.coo_to_csr(ei.to(dev), R)
    by_src = ops.source_sorted(g, R)
    w_e = (torch.randn(E, 4096, generator=gen) * 0.05).to(dev)
    if bf16:
        w_e = w_e.to(torch.bfloat16)
    r1, r2 = [(torch.randn(64, 64, generator=gen) * 0.2).to(dev) for _ in range(2)]
    b1, b2 = [torch.randn(64, generator=gen).to(dev) for _ in range(2)]
    conv = ops.nnconv_bf16w if bf16 else ops.nnconv
    bwd_x = ops.nnconv_bwd_x_bf16w if bf16 else ops.nnconv_bwd_x
    X = torch.zeros((L + 1, R, 64), device=dev)
    X[0] = torch.randn(R, 64, generator=gen).to(dev)
    want = X.clone()
    for a in range(1, L + 1):
        conv(want[a - 1], g, w_e, r1 if a <= depth else r2, b1 if a <= depth else b2, "mean", relu=True, out=want[a])
    ops.nnconv_chain_fwd(X, g, w_e, r1, b1, r2, b2, depth)
    assert torch.equal(X, want) and float(X[L].abs().max()) > 0
    inv = ops.inv_degree(g, "mean")
    g_out = torch.randn(R, 64, generator=gen).to(dev)
    GZ, GS = torch.empty((L, R, 64), device=dev), torch.empty((L, R, 64), device=dev)
    gg = g_out
    for a in range(L, 0, -1):
        ops.relu_bwd2(gg, X[a], inv, GZ[a - 1], GS[a - 1])
        gg = bwd_x(GZ[a - 1], GS[a - 1], by_src, w_e, r1 if a <= depth else r2)
    gz, gs, g_in = ops.nnconv_chain_bwd(g_out, X, inv, by_src, w_e, r1, r2, depth)
    assert torch.equal(gz, GZ) and torch.equal(gs, GS) and torch.equal(g_in, gg)


def test_colsum_atb_bf16_one_pass(dev):
    """colsum(a) and a^T . b (b = 6 fp32 attribute columns) in one pass over a bf16 [rows, n]: against fp64 of the
    stored values, reproducible, and the odd shapes fall back to the two separate ops."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(2)
    for rows, n, kb in ((43712, 1024, 6), (1000, 512, 8), (77, 128, 6), (500, 256, 5)):
        a = (torch.randn(rows, n, generator=gen) * (torch.rand(rows, 1, generator=gen) > 0.5)).to(torch.bfloat16).to(dev)
        b = (torch.randn(rows, kb, generator=gen) * 10).to(dev)
        cs, atb = ops.colsum_atb_bf16(a, b)
        a64 = a.double().cpu()
        assert rel_err(cs, a64.sum(0)) < 2e-6 and rel_err(atb, a64.t() @ b.double().cpu()) < (2e-6 if kb != 5 else 1e-2)
        cs2, atb2 = ops.colsum_atb_bf16(a, b)
        assert torch.equal(cs, cs2) and torch.equal(atb, atb2)


@pytest.mark.parametrize("bf16", [True, False])
@pytest.mark.parametrize("E,L", [(1, 1), (7, 12), (4001, 12), (2500, 16), (300, 5), (900, 18)])
def test_bwd_we_on_the_matrix_pipe_with_column_sums(dev, E, L, bf16):
    """mdno_nnconv_bwd_we[_bf16]_colsum: dW_e[p] = sum_l x_l[src p] (x) gs_l[dst p] as one MFMA k-step per quadrant (three
    bf16 planes per fp32 operand, six products) and, from the same pass, the column sums of the stored tensor.
    Against fp64 (fp32: accumulation error only; bf16: within half a bf16 ulp of it), against the FMA kernel (bf16: the
    same value except where the exact sum sits on a rounding boundary — a handful of one-ulp flips), the sums against a
    fp64 sum of the tensor the kernel itself stored, twice bitwise the same.  L > 16 takes the FMA kernel + colsum."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(100 + E + L)
    R = 77
    ei = torch.randint(0, R, (2, E), generator=gen)
    g = ops.coo_to_csr(ei.to(dev), R)
    X = torch.randn(L, R, 64, generator=gen).to(dev)
    GS = (torch.randn(L, R, 64, generator=gen) * 0.3).to(dev)
    op = ops.nnconv_bwd_we_bf16 if bf16 else ops.nnconv_bwd_we
    d_we, cs = op(X, GS, g, with_colsum=True)
    old = op(X, GS, g)
    src, dst = g.src[:E].long(), g.dst[:E].long()
    want = torch.einsum("lei,leo->eio", X.double()[:, src], GS.double()[:, dst]).reshape(E, 4096)
    got = d_we.double()
    mag = torch.einsum("lei,leo->eio", X.double()[:, src].abs(), GS.double()[:, dst].abs()).reshape(E, 4096)
    ulp = 2.0 ** -8 if bf16 else 0.0
    assert bool(((got - want).abs() <= want.abs() * ulp + mag * 2.0 ** -20).all())
    if bf16:
        diff = d_we != old
        assert float(diff.float().mean()) < 2e-3
        if bool(diff.any()):       # one bf16 ulp apart where they differ (more only where the sum cancels: fp32 accumulation)
            a, b = d_we[diff].double(), old[diff].double()
            assert bool(((a - b).abs() <= torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + mag[diff] * 2.0 ** -20).all())
    else:
        assert bool(((got - old.double()).abs() <= mag * 2.0 ** -20).all())
    assert float((cs.double() - got.sum(0)).abs().max()) <= 1e-5 * float(got.abs().sum(0).max())
    d2, cs2 = op(X, GS, g, with_colsum=True)
    assert torch.equal(d2, d_we) and torch.equal(cs2, cs)


@pytest.mark.parametrize("R,depth", [(3584, 6), (77, 1), (300, 3)])
def test_root_gradients_of_both_convs_in_one_launch(dev, R, depth):
    """mdno_nnconv_bwd_root_pair == two mdno_nnconv_bwd_root calls on the halves, bitwise (same slices, same order), and
    both within fp32 accumulation of the fp64 sums."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(R + depth)
    X = torch.randn(2 * depth, R, 64, generator=gen).to(dev)
    GZ = torch.randn(2 * depth, R, 64, generator=gen).to(dev)
    r1, b1, r2, b2 = ops.nnconv_bwd_root_pair(X, GZ)
    for got_r, got_b, sl in ((r1, b1, slice(0, depth)), (r2, b2, slice(depth, 2 * depth))):
        x, gz = X[sl].reshape(-1, 64), GZ[sl].reshape(-1, 64)
        wr, wb = ops.nnconv_bwd_root(x, gz)
        assert torch.equal(got_r, wr) and torch.equal(got_b, wb)
        assert rel_err(got_r, x.double().t() @ gz.double()) < 2e-6 and rel_err(got_b, gz.double().sum(0)) < 2e-6


def _as_dicts(samples):
    return [dict(x_position=s.x_position.cpu(), x_aminoacid=s.x_aminoacid.cpu(), y=s.y.cpu(),
                 edge_index=s.edge_index.cpu(), edge_attr=s.edge_attr.cpu()) for s in samples]


def _replica_loss(model, O, samples, bf16=False):
    """The exact answer: the oracle's train step (the reference's forward, LpLoss(size_average=False) and
    backward, sample by sample) in fp64 on the model's current parameters.  bf16=True: the same with the bf16
    path's storage roundings put where the device has them (tests/bf16_replica.py)."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if bf16:
        from bf16_replica import train_step_bf16
        return train_step_bf16(O, sd, _as_dicts(samples), model.depth)
    return O.train_step(sd, _as_dicts(samples), model.depth)


def test_train_step_reference_golden_on_device(dev, tmp_path):
    """ONE iteration of the reference's own train() at batch size 1 (tests/golden/train_step_b1.npz, written
    by oracle/gen_golden.py from graph_kernel.py:445-474 + LpLoss(size_average=False) + Adam(0.01, 5e-4)):
    the HIP training path's loss and every parameter gradient against the REFERENCE's, in each GEMM mode, and
    the parameters after training.train_epoch's optimizer step against the reference's updated parameters.
    Width 64, k = 128, depth 2, N = 28, window 10; weights regenerated from the seed (same RNG-draw order)."""
    from test_oracle_golden import _train_step_case
    from molecular_dynamics_neural_operator_amd.dataset import PairData
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import train_epoch
    z = load_golden("train_step_b1.npz")
    sd, sm, depth = _train_step_case(z, "w64")
    want = {k[len("w64.g."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w64.g.")}
    after = {k[len("w64.a."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w64.a.")}
    sample = PairData(sm["x_aminoacid"], sm["x_position"], sm["y"], sm["edge_attr"], sm["edge_index"])
    for gemm_mode in ("f32", "split_bf16", "split_f16"):
        model = KernelNN(*[int(v) for v in z["w64.ctor"]])
        model.load_state_dict(sd)
        model.to(dev).train()
        model.gemm_mode = gemm_mode
        out = model([sample])
        loss = LpLoss(size_average=False)(out.view(1, -1), sample.y.to(dev).view(1, -1))
        loss.backward()
        assert float(loss) == pytest.approx(float(z["w64.loss"]), rel=1e-5)
        errs = {n: rel_err(p_.grad, want[n]) for n, p_ in model.named_parameters()}
        print(gemm_mode, "gradient rel. L2 vs the reference:", {k: f"{v:.1e}" for k, v in errs.items()})
        assert set(errs) == set(want) and max(errs.values()) < 2e-5, errs
        # the optimizer step of train() (graph_kernel.py:467) through the reference-shaped epoch loop
        model.zero_grad(set_to_none=True)
        opt = torch.optim.Adam(model.parameters(), lr=float(z["lr"]), weight_decay=float(z["weight_decay"]))
        avg_loss, avg_mse = train_epoch(model, [[sample]], opt, LpLoss(size_average=False))
        assert avg_loss == pytest.approx(float(z["w64.loss"]), rel=1e-5)
        assert avg_mse == pytest.approx(float(z["w64.mse"]), rel=1e-5)
        now = model.state_dict()
        # Adam's first step moves every entry by lr * sign(g) (|g| >> eps): a sign flip of a near-zero gradient
        # entry would show as 2 * lr; none is allowed on these small tensors
        for k, v in after.items():
            torch.testing.assert_close(now[k].cpu(), v, rtol=1e-4, atol=2e-6, msg=lambda m, k=k: f"{k}: {m}")


def test_gradients_survive_accumulation_and_clipping(dev, tmp_path):
    """every gradient the HIP backward hands to autograd owns its storage.  Two backward passes
    without zeroing (gradient accumulation) give exactly twice one pass, bias_ih / bias_hh do not share a
    buffer, and clip_grad_norm_ scales every tensor once."""
    from test_oracle_golden import _train_step_case
    from molecular_dynamics_neural_operator_amd.dataset import PairData
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    z = load_golden("train_step_b1.npz")
    sd, sm, depth = _train_step_case(z, "w64")
    sample = PairData(sm["x_aminoacid"], sm["x_position"], sm["y"], sm["edge_attr"], sm["edge_index"])
    model = KernelNN(*[int(v) for v in z["w64.ctor"]])
    model.load_state_dict(sd)
    model.to(dev).train()

    def backward_once():
        out = model([sample])
        LpLoss(size_average=False)(out.view(1, -1), sample.y.to(dev).view(1, -1)).backward()

    backward_once()
    once = {n: p_.grad.clone() for n, p_ in model.named_parameters()}
    ptrs = [p_.grad.data_ptr() for p_ in model.parameters()]
    assert len(set(ptrs)) == len(ptrs)                                 # no two .grad tensors alias
    backward_once()                                                     # accumulate: no zero_grad in between
    for n, p_ in model.named_parameters():
        assert torch.equal(p_.grad, 2 * once[n]), n
    model.zero_grad(set_to_none=False)
    backward_once()
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in once.values()))
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float(total) * 0.5)
    for n, p_ in model.named_parameters():
        torch.testing.assert_close(p_.grad, once[n] * 0.5, rtol=1e-5, atol=0.0, msg=lambda m, n=n: f"{n}: {m}")


@pytest.mark.parametrize("gemm_mode", ["split_bf16", "f32"])
def test_model_gradients_vs_fp64_replica(dev, O, tmp_path, gemm_mode):
    """Batch of 3 dataset samples (N=28, W=10), width 64, k=128, depth 2: loss, outputs and every
    parameter gradient of the HIP training path vs the fp64 replica, in both GEMM modes."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import collate, train_forward
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    samples = [dset[i] for i in (0, 7, 19)]
    B = len(samples)
    torch.manual_seed(3)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():                      # keep activations O(1) through 4 random-init layers
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev).train()
    model.gemm_mode = gemm_mode
    out = model(samples)                       # training mode + autograd -> differentiable HIP path
    assert out.requires_grad and out.shape == (B * 28, 3)
    y = torch.cat([s.y for s in samples]).to(dev)
    loss = LpLoss(size_average=False)(out.view(B, -1), y.view(B, -1))
    loss.backward()
    want_loss, want_out, want_grads = _replica_loss(model, O, samples)
    assert abs(float(loss) - want_loss) < 1e-5 * abs(want_loss)
    assert rel_err(out, want_out) < 1e-5
    for name, p_ in model.named_parameters():
        assert p_.grad is not None, name
        assert rel_err(p_.grad, want_grads[name]) < 1e-4, (name, rel_err(p_.grad, want_grads[name]))
    # a second identical pass gives bitwise identical gradients (no float atomics anywhere)
    g1 = {n: p_.grad.clone() for n, p_ in model.named_parameters()}
    model.zero_grad()
    out2 = train_forward(model, samples)
    LpLoss(size_average=False)(out2.view(B, -1), y.view(B, -1)).backward()
    for n, p_ in model.named_parameters():      # every parameter: the per-atom ends are HIP kernels too
        assert torch.equal(p_.grad, g1[n]), n


def test_training_reduces_loss(dev, tmp_path):
    """A few Adam steps on a tiny synthetic trajectory: the loss goes down (graph_kernel.py:541-547)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import train_epoch
    from oracle import graph_kernel_oracle as O
    base = syn.chain_frame(28, seed=0)
    traj = syn.ou_trajectory(base, 60, sigma=0.15, theta=0.2, seed=2)
    cms = [O.radius_graph_coo(f, 8.0).reshape(-1) for f in traj]
    path = tmp_path / "t.npz"
    write_trajectory_npz(path, traj, cms, syn.amino_acids(28, seed=0))
    dset = ContactMapDataset(str(path), window_size=10, horizon=1)
    torch.manual_seed(0)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    batches = [[dset[i] for i in range(s, s + 8)] for s in range(0, 40, 8)]
    first, _ = train_epoch(model, batches, opt, LpLoss(size_average=False))
    for _ in range(5):
        last, _ = train_epoch(model, batches, opt, LpLoss(size_average=False))
    assert last < 0.8 * first, (first, last)


# ------------------------------------------------------------------------------- bf16 training (cfg4)
def test_bf16_ops_against_fp64_of_the_rounded_operands(dev, O):
    """csrc/train_bf16.hip op by op.  GEMMs are exact products of the bf16-rounded operands with fp32
    accumulation, so against fp64 of the SAME rounded operands they are as accurate as an fp32 GEMM; the
    conv kernels with bf16 W_e equal the fp32 kernels fed the rounded weights (same chains, same FMAs)."""
    from molecular_dynamics_neural_operator_amd import ops
    g = torch.Generator().manual_seed(7)
    bf = lambda t_: t_.to(torch.bfloat16)
    for rows, n, k in ((300, 128, 64), (1000, 256, 1024), (517, 4096, 128)):
        a, w, b = torch.randn(rows, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
        ab = ops.cast_bf16(a.to(dev))
        assert torch.equal(ab.cpu(), bf(a))                                             # round to nearest even
        want = F.linear(bf(a).double(), bf(w).double(), b.double())
        got = ops.linear_bf16(ab, w.to(dev), b.to(dev), relu=False, out_bf16=False)
        assert rel_err(got, want) < 3e-6
        got = ops.linear_bf16(ab, w.to(dev), b.to(dev), relu=True, out_bf16=True)
        assert got.dtype == torch.bfloat16 and rel_err(got.float(), want.relu()) < 4e-3      # one bf16 rounding
        assert rel_err(ops.linear_bf16(ab, w.to(dev), None, out_bf16=False), F.linear(bf(a).double(), bf(w).double())) < 3e-6
    for rows, n1, n2 in ((5000, 128, 256), (333, 1024, 128), (4097, 256, 4096), (31, 128, 128)):
        a, b = torch.randn(rows, n1, generator=g), torch.randn(rows, n2, generator=g)
        got = ops.gemm_atb_bf16(bf(a).to(dev), bf(b).to(dev))
        assert rel_err(got, bf(a).double().t() @ bf(b).double()) < 3e-6
        assert torch.equal(got, ops.gemm_atb_bf16(bf(a).to(dev), bf(b).to(dev)))        # fixed-order slices
        assert rel_err(ops.colsum_bf16(bf(a).to(dev)), bf(a).double().sum(0)) < 3e-6
    # first edge-MLP layer (K = 6 edge attributes, bf16 out in one kernel) == fp32 kernel, then the cast, bit for bit
    for rows, n, k, relu in ((1001, 1024, 6, True), (77, 128, 6, False), (300, 64, 8, True), (5, 8, 1, True)):
        a, w, b = torch.randn(rows, k, generator=g).to(dev), torch.randn(n, k, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
        got = ops.linear_smallk_bf16(a, w, b, relu=relu)
        assert got.dtype == torch.bfloat16 and torch.equal(got, ops.cast_bf16(ops.linear(a, w, b, relu=relu)))
        assert torch.equal(ops.linear_smallk_bf16(a, w, None, relu=relu), ops.cast_bf16(ops.linear(a, w, None, relu=relu)))
    gq, y = torch.randn(40, 64, generator=g), torch.randn(40, 64, generator=g)
    assert torch.equal(ops.relu_bwd_bf16(gq.to(dev), bf(y).to(dev), out_bf16=False).cpu(), gq * (bf(y).float() > 0))
    assert torch.equal(ops.relu_bwd_bf16(gq.to(dev), bf(y).to(dev), out_bf16=True).cpu(), bf(gq * (bf(y).float() > 0)))
    # conv kernels on an irregular graph (hub, isolated node, duplicate edges)
    n, E = 90, 1500
    ei = torch.randint(0, n, (2, E), generator=g)
    ei[1, :220] = 11
    ei[1, ei[1] == 30] = 31
    x = torch.randn(n, 64, generator=g).to(dev)
    w_e = (torch.randn(E, 4096, generator=g) * 0.1)
    root, bias = (torch.randn(64, 64, generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev)
    gr = ops.coo_to_csr(ei.to(dev), n)
    perm = gr.perm[:E].long()
    wb = bf(w_e).to(dev)[perm].contiguous()
    y32 = ops.nnconv(x, gr, wb.float(), root, bias, "mean", relu=True)
    yb = ops.nnconv_bf16w(x, gr, wb, root, bias, "mean", relu=True)
    assert torch.equal(yb, y32)
    gy = torch.randn(n, 64, generator=g).to(dev)
    inv = ops.inv_degree(gr, "mean")
    gz, gs = ops.relu_bwd(gy, yb), ops.relu_bwd(gy, yb, inv)
    by_src = ops.source_sorted(gr, n)
    assert torch.equal(ops.nnconv_bwd_x_bf16w(gz, gs, by_src, wb, root), ops.nnconv_bwd_x(gz, gs, by_src, wb.float(), root))
    d32 = ops.nnconv_bwd_we(x.unsqueeze(0), gs.unsqueeze(0), gr)
    assert torch.equal(ops.nnconv_bwd_we_bf16(x.unsqueeze(0), gs.unsqueeze(0), gr), bf(d32))


def test_bf16_model_gradients_vs_fp64_replica(dev, O, tmp_path):
    """train_precision="bf16" on the batch of test_model_gradients_vs_fp64_replica.  Two references:
    (1) the fp64 replica with the bf16 path's storage roundings put where the device has them
    (tests/bf16_replica.py): what is left is fp32-vs-fp64 accumulation, so loss, outputs and EVERY parameter
    gradient must agree to 1e-3 relative L2 (measured ~1e-5) — an indexing error in any of the backward
    layers shows here;  (2) the un-rounded fp64 replica (the reference's arithmetic): what bf16 storage of h1,
    h2, W_e, dW_e costs — 8 mantissa bits per stored value, ~3e-2 on the block's own parameters at 3 samples,
    up to 0.25 on the parameters upstream of 2*depth backward conv steps (sums with heavy cancellation), all
    pointing the same way (cosine > 0.97).  Bitwise repeatable."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import collate, train_forward
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    samples = [dset[i] for i in (0, 7, 19)]
    B = len(samples)
    torch.manual_seed(3)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev).train()
    model.train_precision = "bf16"
    out = model(samples)
    y = torch.cat([s.y for s in samples]).to(dev)
    loss = LpLoss(size_average=False)(out.view(B, -1), y.view(B, -1))
    loss.backward()
    em_loss, em_out, em_grads = _replica_loss(model, O, samples, bf16=True)
    assert abs(float(loss) - em_loss) < 1e-4 * abs(em_loss)
    assert rel_err(out, em_out) < 1e-4
    em_errs = {n: rel_err(p_.grad, em_grads[n]) for n, p_ in model.named_parameters()}
    print("bf16 gradient rel errors vs the rounding-emulating replica:", {k: f"{v:.1e}" for k, v in em_errs.items()})
    assert max(em_errs.values()) < 1e-3, em_errs
    want_loss, want_out, want_grads = _replica_loss(model, O, samples)
    assert abs(float(loss) - want_loss) < 5e-3 * abs(want_loss)
    assert rel_err(out, want_out) < 5e-3
    errs = {n: rel_err(p_.grad, want_grads[n]) for n, p_ in model.named_parameters()}
    print("bf16 gradient rel errors vs the un-rounded replica:", {k: f"{v:.1e}" for k, v in errs.items()})
    for n, p_ in model.named_parameters():
        upstream = n.startswith(("lstm", "emb", "fc1"))
        assert errs[n] < (0.3 if upstream else 6e-2), (n, errs[n])
        cos = F.cosine_similarity(p_.grad.detach().cpu().double().flatten(), want_grads[n].flatten(), dim=0)
        assert float(cos) > 0.97, (n, float(cos))
    g1 = {n: p_.grad.clone() for n, p_ in model.named_parameters()}
    model.zero_grad()
    out2 = train_forward(model, samples)
    LpLoss(size_average=False)(out2.view(B, -1), y.view(B, -1)).backward()
    for n, p_ in model.named_parameters():
        assert torch.equal(p_.grad, g1[n]), n


def test_bf16_training_reduces_loss(dev, tmp_path):
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import train_epoch
    base = syn.chain_frame(28, seed=0)
    traj = syn.ou_trajectory(base, 60, sigma=0.15, theta=0.2, seed=2)
    path = tmp_path / "t.npz"
    write_trajectory_npz(path, traj, [syn.contact_map(f, 8.0) for f in traj], syn.amino_acids(28, seed=0))
    dset = ContactMapDataset(str(path), window_size=10, horizon=1)
    torch.manual_seed(0)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev)
    model.train_precision = "bf16"
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    batches = [[dset[i] for i in range(s, s + 8)] for s in range(0, 40, 8)]
    first, _ = train_epoch(model, batches, opt, LpLoss(size_average=False))
    for _ in range(5):
        last, _ = train_epoch(model, batches, opt, LpLoss(size_average=False))
    assert last < 0.8 * first, (first, last)


# ------------------------------------------------------------------------------- per-atom ends in HIP
@pytest.mark.parametrize("variant", ["intree", "notebook"])
def test_node_prologue_and_fc2_backward_vs_autograd(dev, variant):
    """csrc/train_nodes.hip against torch autograd (fp64) of the same modules: LSTM(3,3) over the window with
    the atoms as the batch and zero initial state, lstm_fc, Embedding, concat, fc1, ReLU (graph_kernel.py:279-298)
    and fc2 (:305).  300 atoms = two workgroups, one partial; repeated residue types; bitwise repeatable."""
    from molecular_dynamics_neural_operator_amd import ops
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, KernelNNNotebook
    gen = torch.Generator().manual_seed(21)
    R, W = 300, (10 if variant == "intree" else 1)
    torch.manual_seed(5)
    model = (KernelNN if variant == "intree" else KernelNNNotebook)(64, 128, 2, 6, 7, 3, 20, 4).to(dev)
    frames = (torch.randn(W, R, 3, generator=gen) * 2.0).to(dev)
    aa = torch.randint(0, 20, (R,), generator=gen).to(dev)
    g0 = torch.randn(R, 64, generator=gen).to(dev)
    pack = model.param_pack(dev, conv_mode="materialized")
    x0 = ops.node_prologue(pack, frames.unsqueeze(1), aa)
    got = ops.node_prologue_bwd(pack, frames.unsqueeze(1), aa, x0, g0)
    # reference: the torch modules in fp64
    import copy
    ref = copy.deepcopy(model).cpu().double()
    fr = frames.cpu().double()
    if variant == "intree":
        hidden = (torch.zeros(1, R, 3, dtype=torch.double), torch.zeros(1, R, 3, dtype=torch.double))
        out = None
        for t_ in range(W):
            out, hidden = ref.lstm(fr[t_].unsqueeze(0), hidden)
        feat = ref.lstm_fc(out.reshape(R, 3))
    else:
        feat = fr[-1]
    want_x0 = F.relu(ref.fc1(torch.cat((ref.emb(aa.cpu()), feat), dim=1)))
    assert rel_err(x0, want_x0) < 1e-6
    want_x0.backward(g0.cpu().double())
    sd = dict(ref.named_parameters())
    for name, g in got.items():
        assert rel_err(g, sd[name].grad) < 2e-5, (name, rel_err(g, sd[name].grad))
    again = ops.node_prologue_bwd(pack, frames.unsqueeze(1), aa, x0, g0)
    assert all(torch.equal(again[k], got[k]) for k in got)
    # fc2
    x = torch.randn(R, 64, generator=gen).to(dev)
    g = torch.randn(R, 3, generator=gen).to(dev)
    w, b = model.fc2.weight.detach(), model.fc2.bias.detach()
    out = ops.fc_out(x, w, b)
    assert rel_err(out, F.linear(x.double().cpu(), w.double().cpu(), b.double().cpu())) < 1e-6
    dx, d_w, d_b = ops.fc_out_bwd(x, w, g)
    assert rel_err(dx, g.double().cpu() @ w.double().cpu()) < 1e-6
    assert rel_err(d_w, g.double().cpu().t() @ x.double().cpu()) < 1e-6
    assert rel_err(d_b, g.double().cpu().sum(0)) < 1e-6


# ------------------------------------------------------------------------------- device-side batches
def test_device_trajectory_batches_equal_host_collation(dev, tmp_path):
    """DeviceTrajectory.batch(indices) — the batch built on the device from the resident trajectory
    (mdno_collate_samples) — holds bit for bit the tensors `collate([dataset[i] ...])` builds on the host from
    ContactMapDataset.__getitem__ (dataset.py:180-227) and the PairData batching rule (dataset.py:41-45):
    unordered, repeated and boundary indices; and a training step on it gives the same loss and gradients."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import DeviceTrajectory, collate
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    for W, h in ((int(z["window"]), 1), (1, 1), (4, 3)):
        dset = ContactMapDataset(str(path), window_size=W, horizon=h)
        traj = DeviceTrajectory(dset, dev)
        assert len(traj) == len(dset)
        idx = [len(dset) - 1, 0, 7, 7, 3]
        got = traj.batch(idx)
        want = collate([dset[i] for i in idx])
        assert got.num_graphs == len(idx)
        for f in ("x_position", "y", "edge_index", "edge_attr", "x_aminoacid"):
            assert torch.equal(getattr(got, f).cpu(), getattr(want, f)), (W, h, f)
        with pytest.raises(IndexError):
            traj.batch([len(dset)])
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    traj = DeviceTrajectory(dset, dev)
    idx = [0, 7, 19]
    torch.manual_seed(3)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev).train()
    res = []
    for batch in (traj.batch(idx), [dset[i] for i in idx]):
        model.zero_grad(set_to_none=True)
        out = model(batch)
        y = batch.y if not isinstance(batch, list) else torch.cat([s.y for s in batch]).to(dev)
        loss = LpLoss(size_average=False)(out.view(3, -1), y.view(3, -1))
        loss.backward()
        res.append((loss.detach().clone(), {n: p_.grad.clone() for n, p_ in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


@pytest.mark.parametrize("gemm_mode", ["split_f16", "split_bf16", "f32"])
def test_validate_epoch_eval_mode_batched_forward(dev, O, tmp_path, gemm_mode):
    """The reference's validate() (graph_kernel.py:476-493): model.eval(), torch.no_grad(), out = model(batch) on a
    list of samples.  In eval mode a list / a device-collated batch runs as B block-diagonal members of one inference
    forward: every sample's rows are BITWISE `model(sample)` on it alone, equal to the training-mode forward to
    fp32 rounding, the losses are the oracle's validation losses, nothing is kept for a backward (peak memory of a
    validation pass below that of a training step), and an index error is raised once per pass."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, PairData
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import DeviceTrajectory, train_forward, validate_epoch
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    traj = DeviceTrajectory(dset, dev)
    torch.manual_seed(11)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev)
    model.gemm_mode = gemm_mode
    model.eval()
    idx = [0, 7, 19, 7, 12]
    N = dset[0].x_aminoacid.shape[0]
    with torch.no_grad():
        singles = [model(dset[i].to(dev)) for i in idx]
        out_list = model([dset[i] for i in idx])                        # CPU samples, as a DataListLoader yields them
        out_dev = model(traj.batch(idx))
        out_lat, lat = model([dset[i].to(dev) for i in idx], return_latent=True)
    assert out_list.shape == (len(idx) * N, 3) and lat.shape == (len(idx) * N, 64)
    assert torch.equal(out_list, torch.cat(singles)) and torch.equal(out_dev, out_list) and torch.equal(out_lat, out_list)
    assert not dset[0].x_position.is_cuda and not out_list.requires_grad
    # eval mode under enable_grad still takes the inference path (no autograd state)
    with torch.enable_grad():
        assert not model([dset[0], dset[1]]).requires_grad
    # == the training-mode forward on the same batch, to fp32 rounding (different kernels: fp32 Linear ops there)
    model.train()
    with torch.enable_grad():
        tr = train_forward(model, traj.batch(idx))
    assert rel_err(out_list, tr) < 2e-6
    # validate_epoch: the reference's averages (loss per batch via LpLoss(size_average=False), MSE per batch)
    loss_fn = LpLoss(size_average=False)
    vb = [[3, 4], [10, 2], [19, 0]]
    model.eval()
    got_loss, got_mse = validate_epoch(model, ([dset[i] for i in b] for b in vb), loss_fn)
    got_loss_d, got_mse_d = validate_epoch(model, (traj.batch(b) for b in vb), loss_fn)
    assert (got_loss, got_mse) == (got_loss_d, got_mse_d) and not model.training
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want_loss = want_mse = 0.0
    for b in vb:
        outs, ys = [], []
        for i in b:
            s = dset[i]
            outs.append(O.kernelnn_forward(sd, s.x_position, s.x_aminoacid, s.edge_index, s.edge_attr, model.depth, hoist=True))
            ys.append(s.y)
        o, y = torch.cat(outs), torch.cat(ys)
        want_loss += float(O.lp_loss_rel(o.view(2, -1), y.view(2, -1), size_average=False))
        want_mse += float(F.mse_loss(o, y))
    assert got_loss == pytest.approx(want_loss / 3, rel=1e-5) and got_mse == pytest.approx(want_mse / 3, rel=1e-5)
    model.train()
    validate_epoch(model, [[dset[0]]], loss_fn)
    assert model.training                                               # the mode it was called in is restored
    # memory: a validation pass on a batch of 16 stays below a training step on the same batch
    big = list(range(16))
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    from molecular_dynamics_neural_operator_amd.training import train_epoch
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    validate_epoch(model, [traj.batch(big)], loss_fn)
    torch.cuda.synchronize()
    peak_val = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    train_epoch(model, [traj.batch(big)], opt, loss_fn)
    torch.cuda.synchronize()
    peak_train = torch.cuda.max_memory_allocated() - base
    print(f"peak memory above the resident set: validation {peak_val / 2**20:.1f} MiB, training {peak_train / 2**20:.1f} MiB")
    assert peak_val < peak_train
    # an amino-acid id outside the table: raised once, at the end of the pass; cleared afterwards
    bad = dset[0]
    bad.x_aminoacid = bad.x_aminoacid.clone()
    bad.x_aminoacid[3] = 20
    with pytest.raises(IndexError):
        validate_epoch(model, [[dset[1], bad]], loss_fn)
    validate_epoch(model, [[dset[1], dset[2]]], loss_fn)
    with pytest.raises(IndexError):
        model.eval()
        with torch.no_grad():
            model([dset[1], bad])                                       # a direct call checks at once



def test_training_index_errors_are_deferred_not_lost(dev, tmp_path):
    """The training forward does not wait for the device; an amino-acid id outside the embedding table (the
    reference: IndexError from nn.Embedding inside that forward) is raised by check_train_status / at the end
    of train_epoch, and the status word is cleared afterwards."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import check_train_status, train_epoch
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    bad = dset[0]
    bad.x_aminoacid = bad.x_aminoacid.clone()
    bad.x_aminoacid[3] = 20
    with pytest.raises(IndexError):
        train_epoch(model, [[dset[1], bad]], opt, LpLoss(size_average=False))
    check_train_status(model)                          # cleared: nothing left to raise
    train_epoch(model, [[dset[1], dset[2]]], opt, LpLoss(size_average=False))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,n,k", [(700, 256, 1024), (4500, 1024, 256)])      # few-row kernel / 256 x 128 tiles
def test_linear_split_f16_rows_of_any_magnitude(dev, rows, n, k):
    """ops.linear(gemm_mode="split_f16") — the fp32 training path's GEMMs: two fp16 planes per operand with every
    row of A and of W scaled by its own power of two.  Rows spanning 1e-12 .. 1e8 (gradients next to activations)
    must each come out at fp32 level — worst output row's relative L2 error vs fp64 < 1e-6, and no worse than the
    bf16 3-way split's worst row."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(rows)
    a = torch.randn(rows, k, generator=gen) * (10.0 ** (torch.rand(rows, 1, generator=gen) * 20 - 12))
    w = torch.randn(n, k, generator=gen) * (10.0 ** (torch.rand(n, 1, generator=gen) * 8 - 4))
    b = torch.randn(n, generator=gen) * 1e-3
    ref = a.double() @ w.double().t()
    err = {}
    for mode in ("split_f16", "split_bf16"):
        got = ops.linear(a.to(dev), w.to(dev), None, gemm_mode=mode).cpu().double()
        err[mode] = float(((got - ref).norm(dim=1) / ref.norm(dim=1)).max())
    print("worst row, rel L2 vs fp64:", {m: f"{e:.2e}" for m, e in err.items()})
    assert err["split_f16"] < 1e-6 and err["split_f16"] < 2 * err["split_bf16"], err
    # bias + ReLU epilogue
    got = ops.linear(a.to(dev), w.to(dev), b.to(dev), relu=True, gemm_mode="split_f16").cpu().double()
    want = torch.relu(ref + b.double())
    assert float((got - want).norm() / want.norm()) < 1e-6
    # a non-finite input row stays that row's problem
    a2 = a.clone()
    a2[3, 5] = float("inf")
    got = ops.linear(a2.to(dev), w.to(dev), None, gemm_mode="split_f16").cpu()
    assert not torch.isfinite(got[3]).all() and torch.isfinite(got[4:]).all() and torch.isfinite(got[:3]).all()


@pytest.mark.gpu
def test_gemm_atb_split_f16_columns_of_any_magnitude(dev):
    """ops.gemm_atb(gemm_mode="split_f16") — the weight gradients of the fp32 training path on two fp16 planes per
    operand with every column scaled by its own power of two.  Columns spanning 1e-12 .. 1e6: every entry of the
    product, measured against the norms of its two columns, is as close to fp64 as the exact-fp32-MFMA product."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(9)
    rows, n1, n2 = 5000, 512, 256
    a = torch.randn(rows, n1, generator=gen) * (10.0 ** (torch.rand(1, n1, generator=gen) * 18 - 12))
    b = torch.randn(rows, n2, generator=gen) * (10.0 ** (torch.rand(1, n2, generator=gen) * 12 - 6))
    a[:, 7] = 0.0                                                      # an all-zero column: scale 1, result 0
    ref = a.double().t() @ b.double()
    scale = a.double().norm(dim=0).clamp_min(1e-300)[:, None] * b.double().norm(dim=0)[None, :]
    err = {}
    for mode in ("split_f16", "f32"):
        got = ops.gemm_atb(a.to(dev), b.to(dev), gemm_mode=mode).cpu().double()
        assert torch.isfinite(got).all()
        err[mode] = float(((got - ref) / scale).abs().max())
    print("max |C - fp64| / (|a_i| |b_j|):", {m: f"{e:.2e}" for m, e in err.items()})
    assert err["split_f16"] < 3e-7 and err["split_f16"] < 3 * err["f32"] + 1e-8, err
    assert float(ops.gemm_atb(a.to(dev), b.to(dev), gemm_mode="split_f16")[7].abs().max()) == 0.0
    # two runs, same bits (slabs are added in a fixed order)
    assert torch.equal(ops.gemm_atb(a.to(dev), b.to(dev), gemm_mode="split_f16"), ops.gemm_atb(a.to(dev), b.to(dev), gemm_mode="split_f16"))


# ------------------------------------------------------------------------------- optimiser trajectory (cfg4)
def _oracle_adam_trajectory(step_fn, sd0, batches, depth, lr, weight_decay, step_size, gamma):
    """The reference's epoch loop (graph_kernel.py:583-622: train() over the loader, then scheduler.step())
    with one batch per epoch, on the HOST in fp64: `step_fn` (the oracle's train step, or its bf16-storage
    replica) gives loss and gradients, torch.optim.Adam + StepLR (:541-546) move fp64 copies of the
    parameters.  conv1.net and conv2.net are one module in the reference (:271-273): one set of tensors."""
    p = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd0.items() if not k.startswith("conv2.net.")}
    opt = torch.optim.Adam(list(p.values()), lr=lr, weight_decay=weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=step_size, gamma=gamma)
    losses, lrs = [], []
    for samples in batches:
        sd = {k: v.detach() for k, v in p.items()}
        loss, _, g = step_fn(sd, samples)
        opt.zero_grad()
        for k, v in p.items():
            v.grad = g[k].detach().double().clone()
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
        losses.append(loss)
    final = {k: v.detach() for k, v in p.items()}
    for k in list(final):
        if k.startswith("conv1.net."):
            final["conv2.net." + k[len("conv1.net."):]] = final[k]
    return losses, final, lrs


@pytest.mark.parametrize("precision,lr,tol_loss,tol_param", [("fp32", 3e-4, 1e-5, 1e-4), ("fp32", 1e-2, 1e-3, 1e-2),
                                                             ("bf16", 3e-4, 1e-3, 2e-2)])
@pytest.mark.parametrize("optimizer", ["torch", "mdno"])
def test_adam_trajectory_vs_oracle(dev, O, tmp_path, precision, lr, tol_loss, tol_param, optimizer):
    """SURVEY.md §8(d) cfg4, "loss ... vs the CPU restatement" over a multi-step optimiser trajectory: six epochs
    of one batch of 4 dataset samples through training.train_epoch with the reference's optimiser set-up —
    Adam(lr, weight_decay 5e-4) + StepLR(step_size 2, gamma 0.8), scheduler.step() after every epoch
    (graph_kernel.py:445-474, 541-546, 622), so two decays land inside — at width 64, k = 128, depth 2.  The same
    loop runs on the host in fp64 over the oracle's train step.
      fp32 (GEMM mode split_f16), lr 3e-4: a smooth descent (4.02 -> 2.33); every epoch's loss within 1e-5 of the
           oracle's, the final parameters within 1e-4 (relative L2 per tensor).
      fp32, lr 0.01 (the reference's CLI default, graph_kernel.py:319): at this model size the first Adam step
           moves every weight by 0.01 and the second epoch's loss is 5,641 — the map from parameters to the next
           loss amplifies a 1e-7 perturbation of the start 50x (measured on the oracle itself), so the bar is 1e-3.
      bf16, lr 3e-4: against the replica with the device's bf16 storage roundings (tests/bf16_replica.py), 1e-3."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import train_epoch
    z = load_golden("rollout_20.npz")
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=int(z["window"]), horizon=1)
    epochs, B, depth = 6, 4, 2
    wd, step_size, gamma = 5e-4, 2, 0.8
    idx = [[(4 * e + 3 * j) % len(dset) for j in range(B)] for e in range(epochs)]
    torch.manual_seed(11)
    model = KernelNN(64, 128, depth, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)
    model.gemm_mode, model.train_precision = "split_f16", precision
    # torch.optim.Adam as the reference sets it up (graph_kernel.py:541-543), or training.Adam: the same update as one
    # libmdno launch over all parameter tensors
    from molecular_dynamics_neural_operator_amd.training import Adam as MdnoAdam
    opt = (MdnoAdam if optimizer == "mdno" else torch.optim.Adam)(model.parameters(), lr=lr, weight_decay=wd)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=step_size, gamma=gamma)
    got, got_lr = [], []
    for e in range(epochs):
        loss, _ = train_epoch(model, [[dset[i] for i in idx[e]]], opt, LpLoss(size_average=False))
        got_lr.append(opt.param_groups[0]["lr"])
        sched.step()
        got.append(loss)
    if precision == "bf16":
        from bf16_replica import train_step_bf16
        step_fn = lambda sd, samples: train_step_bf16(O, sd, _as_dicts(samples), depth)        # noqa: E731
    else:
        step_fn = lambda sd, samples: O.train_step(sd, _as_dicts(samples), depth)              # noqa: E731
    want, final, want_lr = _oracle_adam_trajectory(step_fn, sd0, [[dset[i] for i in idx[e]] for e in range(epochs)],
                                                    depth, lr, wd, step_size, gamma)
    assert got_lr == pytest.approx(want_lr) and got_lr[0] == lr and got_lr[-1] == pytest.approx(lr * gamma ** 2)
    rel = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    print(precision, "loss per epoch", [f"{v:.6f}" for v in got], "rel. to the oracle", [f"{v:.1e}" for v in rel])
    assert abs(want[-1] - want[0]) > 0.05 * want[0]          # the trajectory is a real one: the loss moves
    assert max(rel) < tol_loss, rel
    now = model.state_dict()
    errs = {k: rel_err(now[k], final[k]) for k in final}
    print(precision, "final parameters, rel. L2 vs the oracle:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < tol_param, errs


def test_mdno_adam_matches_torch_adam(dev):
    """training.Adam (mdno_adam_step: every parameter tensor of a group in ONE launch) against torch.optim.Adam on the
    host in fp64 and on the device in fp32, step by step: tensors of awkward sizes (1, 3, 108 floats cut into unaligned
    slices like the LSTM gradients, 4,097, 1.05 M elements), weight decay, a learning rate a scheduler changes between
    steps, a parameter that gets no gradient in one step; state_dicts interchange with torch.optim.Adam both ways."""
    from molecular_dynamics_neural_operator_amd.training import Adam as MdnoAdam
    gen = torch.Generator().manual_seed(5)
    shapes = [(1,), (3,), (12, 3), (4097,), (1024, 1025), (20, 4)]
    init = [torch.randn(*sh, generator=gen) for sh in shapes]
    buf = torch.randn(108, generator=gen)                       # gradients handed out as slices of one buffer
    def make(dtype, device):
        return [torch.nn.Parameter(t.detach().clone().to(device=device, dtype=dtype)) for t in init]
    ours, theirs, exact = make(torch.float32, dev), make(torch.float32, dev), make(torch.float64, "cpu")
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    o1, o2, o3 = MdnoAdam(ours, **kw), torch.optim.Adam(theirs, **kw), torch.optim.Adam(exact, **kw)
    scheds = [torch.optim.lr_scheduler.StepLR(o, step_size=2, gamma=0.5) for o in (o1, o2, o3)]
    for step in range(6):
        grads = [torch.randn(*sh, generator=gen) * (10.0 ** (step - 3)) for sh in shapes]
        grads[2] = buf[5:41].reshape(12, 3) * (step + 1)        # an unaligned contiguous slice
        for plist, dt, dv in ((ours, torch.float32, dev), (theirs, torch.float32, dev), (exact, torch.float64, "cpu")):
            for i, (p_, g_) in enumerate(zip(plist, grads)):
                p_.grad = None if (i == 1 and step == 2) else g_.to(device=dv, dtype=dt)
        for o, sc in zip((o1, o2, o3), scheds):
            o.step()
            sc.step()
        for a_, b_, c_ in zip(ours, theirs, exact):
            ref = c_.detach()
            err = float((a_.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            err_t = float((b_.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            assert err < 3e-6 and err <= 4 * err_t + 1e-7, (step, tuple(a_.shape), err, err_t)
    assert o1.param_groups[0]["lr"] == pytest.approx(3e-3 * 0.5 ** 3)
    # state_dict layout == torch.optim.Adam's: each loads the other's and carries on identically to fp32 rounding
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sd1["state"].keys() == sd2["state"].keys() and set(sd1["state"][0]) == set(sd2["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd1["state"][0]["step"]) == float(sd2["state"][0]["step"]) == 6.0 and float(sd1["state"][1]["step"]) == 5.0
    n1, n2 = MdnoAdam(make(torch.float32, dev), **kw), torch.optim.Adam(make(torch.float32, dev), **kw)
    n1.load_state_dict(sd2)
    n2.load_state_dict(sd1)
    for o in (n1, n2):
        for p_, src_ in zip(o.param_groups[0]["params"], ours):
            p_.data.copy_(src_.data)
            p_.grad = torch.ones_like(p_) * 0.25
        o.step()
    for a_, b_ in zip(n1.param_groups[0]["params"], n2.param_groups[0]["params"]):
        torch.testing.assert_close(a_.detach(), b_.detach(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("B,D,size_average", [(1, 84, False), (4, 84, True), (128, 84, False), (37, 1512, False), (300, 3, True)])
def test_lploss_rel_on_device_vs_oracle(dev, O, B, D, size_average):
    """LpLoss.rel (p = 2) on device tensors runs in libmdno (csrc/loss.hip): loss, gradient and the batch MSE against
    the oracle's lp_loss_rel / torch autograd in fp64 (graph_kernel.py:105-119, 462-465); a sample that is hit
    exactly gets a zero gradient (torch.norm's subgradient); an upstream gradient scales the result; bitwise
    reproducible."""
    from molecular_dynamics_neural_operator_amd.graph_kernel import LpLoss
    g = torch.Generator().manual_seed(B * 1000 + D)
    y = torch.randn(B, D, generator=g) * 5.0
    x = y + 0.3 * torch.randn(B, D, generator=g)
    if B > 2:
        x[1] = y[1]                                                  # an exact hit
    xd = x.to(dev).requires_grad_(True)
    fn = LpLoss(size_average=size_average)
    loss, mse = fn.rel_with_mse(xd, y.to(dev))
    assert loss.dim() == 0 and not mse.requires_grad
    (loss * 3.0).backward()
    x64 = x.double().requires_grad_(True)
    want = O.lp_loss_rel(x64, y.double(), size_average=size_average)
    (want * 3.0).backward()
    assert float(loss) == pytest.approx(float(want), rel=2e-6)
    assert float(mse) == pytest.approx(float(((x.double() - y.double()) ** 2).mean()), rel=2e-6)
    assert rel_err(xd.grad, x64.grad) < 2e-6
    if B > 2:
        assert float(xd.grad[1].abs().max()) == 0.0
    assert float(fn(xd.detach(), y.to(dev))) == float(loss)          # __call__ == rel, same bits on a second run
    # the other forms stay the reference's torch expressions
    per_sample = LpLoss(reduction=False)(xd.detach(), y.to(dev))
    assert per_sample.shape == (B,)
    torch.testing.assert_close(per_sample.cpu().double(), O.lp_loss_rel(x.double(), y.double(), reduction=False), rtol=1e-5, atol=0)


def test_lploss_reference_golden_on_device(dev):
    """tests/golden/lploss.npz (the reference's own LpLoss, oracle/gen_golden.py) through the device op."""
    from molecular_dynamics_neural_operator_amd.graph_kernel import LpLoss
    z = load_golden("lploss.npz")
    x, y = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["y"]).to(dev)
    torch.testing.assert_close(LpLoss(size_average=False)(x, y).cpu(), torch.from_numpy(z["rel_sum"]), rtol=2e-6, atol=0)
    torch.testing.assert_close(LpLoss(size_average=True)(x, y).cpu(), torch.from_numpy(z["rel_mean"]), rtol=2e-6, atol=0)


def test_csr_by_source_and_permute_rows(dev):
    """mdno_csr_by_source == mdno_coo_to_csr on the swapped (dst, src) arrays of the destination-sorted graph (what the
    training forward did through torch.stack(...).to(long)), bitwise, on a graph with a hub row above the big-row
    threshold, duplicates and empty rows; mdno_permute_rows == x[perm]; the edge count is written by the sort itself."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(9)
    R, E = 700, 30000
    ei = torch.randint(0, R, (2, E), generator=gen)
    ei[0, :2600] = 3                                  # a hub SOURCE: a by-source row beyond kBigRow = 2,048
    ei[1, 2600:5300] = 11                             # and a hub target
    ei[:, 6000:6100] = ei[:, 5900:6000]               # duplicates
    ei[0, ei[0] == 50] = 51                           # node 50: no out-edge
    g = ops.coo_to_csr(ei.to(dev), R)
    assert int(g.num_edges.item()) == E == g.edge_count()
    want = ops.coo_to_csr(torch.stack([g.dst[:E], g.src[:E]]).to(torch.long), R)
    got = ops.source_sorted(g, R)
    for f in ("row_ptr", "src", "dst", "perm"):
        assert torch.equal(getattr(got, f)[:E + 1 if f == "row_ptr" else E], getattr(want, f)[:E + 1 if f == "row_ptr" else E]), f
    assert int(got.row_ptr[51]) == int(got.row_ptr[50])        # the empty row
    x = torch.randn(E, 6, generator=gen).to(dev)
    assert torch.equal(ops.permute_rows(x, g.perm, E), x[g.perm[:E].long()])
    x1 = torch.randn(E, generator=gen).to(dev)
    assert torch.equal(ops.permute_rows(x1, g.perm, E), x1[g.perm[:E].long()])
