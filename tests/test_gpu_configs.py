"""GPU tests at the BASELINE.json configuration sizes (cfg2 live fixture, cfg3 ensemble share, cfg4
training step, cfg5 50k-atom box) plus the reference-shaped entry points that had no coverage
(propogate, checkpoint loading on the device, index validation).  Everything runs through the C ABI.

Floating point: rtol 1e-4, atol 1e-4*max|y| per element AND relative L2 <= 1e-5 (`close`, shared
with test_gpu_parity.py); graphs, CSR and edge counts bit-exact.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, write_golden_trajectory
from test_gpu_parity import close, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from molecular_dynamics_neural_operator_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import graph_kernel_oracle
    return graph_kernel_oracle


def cm_checksum(cm):
    c = np.asarray(cm, dtype=np.uint64)
    w = (np.arange(c.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return np.uint64((c * w).sum())


def near_threshold_atoms(frame, thr, eps):
    """Atoms that are an endpoint of a pair whose f64 distance lies within eps of the cutoff."""
    p = frame.astype(np.float64)
    d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
    i, j = np.nonzero(np.abs(d - thr) < eps)
    return np.unique(i), len(i) // 2


# ------------------------------------------------------------------------------- cfg2, live fixture
@pytest.fixture(scope="module")
def live504(tmp_path_factory):
    """The reference's own KernelNN (width 64, k=1024, depth 6) at N=504 with bounded, live
    activations: 5 teacher-forced forwards + 5 free-running steps (oracle/gen_golden.py gen_live504)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    z = load_golden("kernelnn_live504.npz")
    thr, W = float(z["threshold"]), int(z["window"])
    frames = z["frames"]
    cms = [syn.contact_map(f, thr) for f in frames]
    assert [c.size for c in cms] == list(z["contact_map_len"])
    assert [cm_checksum(c) for c in cms] == list(z["contact_map_checksum"])      # the reference's own contact maps
    path = tmp_path_factory.mktemp("live504") / "traj.npz"
    write_trajectory_npz(path, frames, cms, z["amino_acids"])
    dset = ContactMapDataset(str(path), window_size=W, horizon=1)
    seed, kg, fg, kc = z["weight_gains"]
    sd = near_identity_state_dict(64, 1024, seed=int(seed), kernel_gain=float(kg), feature_gain=float(fg),
                                  kernel_to_coords=float(kc))
    for n, s_, a_ in zip([str(x) for x in z["param_names"]], z["param_sum"], z["param_abs_sum"]):
        assert float(sd[n].double().sum()) == pytest.approx(float(s_), rel=1e-12, abs=1e-12), n
        assert float(sd[n].double().abs().sum()) == pytest.approx(float(a_), rel=1e-12), n
    return z, dset, sd


@pytest.mark.parametrize("gemm_mode", ["split_bf16", "split_f16", "f32"])
def test_live504_teacher_forced_reference_golden(dev, live504, gemm_mode):
    """The 64x64 fast kernels against the REFERENCE on live activations: materialized (the sample's own
    edge list) and factored (the same graph rebuilt on the device from the sample's first frame)."""
    from molecular_dynamics_neural_operator_amd import ops
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    z, dset, sd = live504
    model = KernelNN(*[int(v) for v in z["ctor"]])
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode = gemm_mode
    lat0 = z["teacher_forced_latent0"]
    assert 0.3 < float((lat0 == 0).mean()) < 0.7 and float(np.abs(lat0).max()) < 50      # live, bounded
    thr = float(z["threshold"])
    for i in range(z["teacher_forced_out"].shape[0]):
        s = dset[i].to(dev)
        # model(sample) on the sample's own edge list: the reference's formulation, and what "auto" picks for this
        # graph (120 neighbours per atom: the factored form, which takes any edge list in every GEMM mode)
        for conv_mode in ("materialized", "auto"):
            model.conv_mode = conv_mode
            with torch.no_grad():
                out, lat = model(s, return_latent=True)
            close(out, z["teacher_forced_out"][i], name=f"tf{i} model(sample) {conv_mode} {gemm_mode}")
            if i == 0:
                close(lat, lat0, name=f"latent0 model(sample) {conv_mode}")
        picked = model._conv_mode_for_edges(dev, 1, s.x_aminoacid.shape[0], s.edge_index.shape[1])
        assert picked == "factored"
        # factored: the sample's graph is the radius graph of its FIRST window frame (dataset.py:189-201)
        first = s.x_position[0].contiguous()
        g = ops.radius_graph(first, first.shape[0], thr)
        assert torch.equal(g.to_edge_index(), s.edge_index)
        counts = {}
        of, lf = ops.kernelnn_forward(model.param_pack(dev, conv_mode="factored"), s.x_position.unsqueeze(1),
                                      s.x_aminoacid, g, edge_pos=first, return_latent=True, fallback_counts=counts)
        close(of, z["teacher_forced_out"][i], name=f"tf{i} factored {gemm_mode}")
        # a live model inside the fp16 planes' ranges: every product of the forward took the fast path (and the
        # counters exist, and stay zero, in the other GEMM modes)
        assert set(counts) == set(ops.FALLBACK_KEYS) and all(v == 0 for v in counts.values()), counts
        if i == 0:
            close(lf, lat0, name="latent0 factored")


@pytest.mark.parametrize("conv_mode,gemm_mode", [("factored", "split_bf16"), ("materialized", "split_bf16"),
                                                 ("factored", "split_f16"), ("materialized", "split_f16"),
                                                 ("factored", "f32"), ("materialized", "f32")])
def test_live504_free_run_reference_golden(dev, live504, conv_mode, gemm_mode):
    """5 free-running steps through recursive_propagation (on-device loop) against the reference's own
    loop (graph_kernel.py:396-413).  A pair whose distance sits within 2e-4 A of the cutoff in the
    reference's frame may fall on the other side for fp32-rounded positions (the reference's frames
    here have pairs as close as 2e-6 A): its two atoms are then compared loosely from that step on."""
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, recursive_propagation
    z, dset, sd = live504
    model = KernelNN(*[int(v) for v in z["ctor"]])
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode, model.conv_mode = gemm_mode, conv_mode
    thr = float(z["threshold"])
    want = z["free_frames"]
    steps = want.shape[0]
    fc = recursive_propagation(model, dset, dev, num_steps=steps, starting_points=[0], threshold=thr)
    got = np.stack([f.x_position[-1].numpy() for f in fc])
    loose = np.zeros(want.shape[1], dtype=bool)
    risky_pairs = 0
    for s in range(steps):
        strict = ~loose
        close(got[s][strict], want[s][strict], name=f"free step {s} {conv_mode}/{gemm_mode} ({int(strict.sum())} atoms)")
        if loose.any():
            assert float(np.abs(got[s][loose] - want[s][loose]).max()) < 5e-2
        # the graph the NEXT step runs on is built from this frame
        e_got, e_want = fc[s].edge_index.shape[1], int(z["free_num_edges"][s])
        atoms, pairs = near_threshold_atoms(want[s], thr, 2e-4)
        risky_pairs += pairs
        assert abs(e_got - e_want) <= 2 * risky_pairs, (s, e_got, e_want)
        loose[atoms] = True
    assert int(loose.sum()) < 0.2 * loose.size


def test_split_f16_hidden_gemm_accuracy_and_range_fallback(dev, live504):
    """gemm_mode "split_f16": the factored path's hidden GEMM and its K1 / K2 (csrc/moment.hip) on two fp16 planes.
    (1) its latent is as close to the reference's as the bf16-split and exact-fp32 kernels'; (2) out of fp16
    range — an activation (coordinates scaled up), a weight, node features far above 65504 or below 2^-10 —
    the hidden GEMM's device-side flags send it through the bf16 kernels inside the same forward, a K1
    workgroup whose own operands are out of range reruns its destination on the bf16 planes, and K2 scales
    every row of S and every column of W3R by its own power of two: results stay finite and as close to the
    exact-fp32 kernels as the bf16 planes' are, up to latents of 1e34."""
    from molecular_dynamics_neural_operator_amd import ops
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    z, dset, sd = live504
    model = KernelNN(*[int(v) for v in z["ctor"]])
    model.load_state_dict(sd)
    model.eval().to(dev)
    s = dset[0].to(dev)
    first = s.x_position[0].contiguous()
    g = ops.radius_graph(first, first.shape[0], float(z["threshold"]))

    def latent(mode, frames=s.x_position, pos=first, graph=g):
        model.gemm_mode = mode
        return ops.kernelnn_forward(model.param_pack(dev, conv_mode="factored"), frames.unsqueeze(1), s.x_aminoacid, graph,
                                    edge_pos=pos, return_latent=True)[1]

    want = torch.from_numpy(z["teacher_forced_latent0"]).double()
    err = {m: float((latent(m).cpu().double() - want).norm() / want.norm()) for m in ("split_f16", "split_bf16", "f32")}
    print("latent rel-L2 vs reference:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["split_f16"] < 3 * max(err["split_bf16"], err["f32"]) and err["split_f16"] < 1e-6
    assert not torch.equal(latent("split_f16"), latent("split_bf16"))            # (it IS a different kernel)
    def rel(u, v):
        return float((u.double() - v.double()).norm() / v.double().norm())

    def held(a, b, c, floor=3e-6):      # a (fp16 planes) as close to c (fp32 MFMA) as b (bf16 planes) is, and finite
        assert bool(torch.isfinite(a).all()) and rel(a, c) < 3 * max(rel(b, c), floor), (rel(a, c), rel(b, c))

    # (2a) edge-MLP activations out of range: the same cloud, coordinates x 3e4 (layer-0 activations reach ~1e5,
    # the latent 1e34) -> the hidden GEMM of this forward runs on the bf16 kernels, every K1 workgroup reruns on the
    # bf16 planes, K2's row scales span a hundred binades.  (x 3e5 overflows fp32 itself in every mode.)
    big_frames, big_pos = s.x_position * 3.0e4, first * 3.0e4
    gb = ops.radius_graph(big_pos, first.shape[0], float(z["threshold"]) * 3.0e4)
    a, b, c = (latent(m, big_frames, big_pos, gb) for m in ("split_f16", "split_bf16", "f32"))
    assert float(c.abs().max()) > 1e30
    held(a, b, c)
    # (2b) one hidden-layer weight above fp16's range: its ROW is scaled down by a power of two before the
    # split (split_layout.h) and the product's column scaled back — still the fp16 planes, still fp32-accurate
    with torch.no_grad():
        model.conv1.net.layers[2].weight[5, 7] = 1.0e5
    a, b = latent("split_f16"), latent("split_bf16")
    assert bool(torch.isfinite(a).all()) and rel(a, b) < 1e-6 and not torch.equal(a, b)
    # (2c) node features out of range (fc1 scaled up: |x| ~ 1e5-1e6 from the first conv application on): every K1
    # workgroup reruns on the bf16 planes
    with torch.no_grad():
        model.fc1.weight.mul_(3.0e4)
        model.fc1.bias.mul_(3.0e4)
    a, b, c = latent("split_f16"), latent("split_bf16"), latent("f32")
    assert float(b.abs().max()) > 65504.0
    held(a, b, c)
    # activations of (2a) AND node features above 65504 in every application (fc1 x 20 on the saturated LSTM
    # output ~7.6e3; the kernel integral damped so that 12 layers stay finite)
    model.load_state_dict(sd)
    model.to(dev)
    with torch.no_grad():
        model.fc1.weight.mul_(20.0)
        model.fc1.bias.mul_(20.0)
        model.conv1.net.layers[4].weight.mul_(1.0e-4)
        model.conv1.net.layers[4].bias.mul_(1.0e-4)
    a, b, c = (latent(m, big_frames, big_pos, gb) for m in ("split_f16", "split_bf16", "f32"))
    assert float(b.abs().max()) > 65504.0
    held(a, b, c)
    # (2d) node features BELOW what two fp16 planes resolve (every |x| < 2^-10; here ~1e-7, where fp16's grid is
    # 6e-8): K1 must take the bf16 planes — on the fp16 planes the latent would be off by percents
    model.load_state_dict(sd)
    model.to(dev)
    with torch.no_grad():
        model.fc1.weight.mul_(1.0e-8)
        model.fc1.bias.mul_(1.0e-8)
        model.conv1.bias.mul_(1.0e-8)
        model.conv2.bias.mul_(1.0e-8)
    a, b = latent("split_f16"), latent("f32")
    assert 0.0 < float(b.abs().max()) < 2.0 ** -10 and rel(a, b) < 1e-6
    # (3) the materialized path (model(data) with the sample's own edge list): both edge-MLP GEMMs run on fp16
    # planes; activations out of range send all of it through the bf16 kernels -> bit-identical to split_bf16;
    # a last-layer weight above 65504 only changes that row's power-of-two scale
    model.load_state_dict(sd)
    model.to(dev)
    model.conv_mode = "materialized"      # ("auto" takes the factored form on this graph)

    def forward(mode, sample):
        model.gemm_mode = mode
        with torch.no_grad():
            return model(sample)

    assert rel(forward("split_f16", s), forward("split_bf16", s)) < 1e-6
    assert not torch.equal(forward("split_f16", s), forward("split_bf16", s))
    big = dset[0].to(dev)
    big.x_position, big.edge_attr = big.x_position * 3.0e5, big.edge_attr * 3.0e5
    # (at this scale the 12 applications overflow: the outputs are NaN — which every ReLU now passes on, as torch's does —
    # in both modes alike; the fallback being the bf16 kernels themselves, the two agree bit for bit, NaNs included)
    fa, fb = forward("split_f16", big), forward("split_bf16", big)
    assert torch.equal(fa.view(torch.int32), fb.view(torch.int32))
    with torch.no_grad():
        model.conv1.net.layers[4].weight[3, 9] = 1.0e5
    # (the 1e5 entry makes the forward ill-conditioned — |out| ~ 1e6 from cancelling terms —, so the three
    # modes are held against each other: the fp16 planes are as close to exact-fp32 MFMA as the bf16 planes are)
    a, b, c = forward("split_f16", s), forward("split_bf16", s), forward("f32", s)
    assert bool(torch.isfinite(a).all()) and not torch.equal(a, b)
    assert rel(a, c) < 3 * max(rel(b, c), 1e-6) and rel(a, c) < 1e-5, (rel(a, c), rel(b, c))


# ------------------------------------------------------------------------------- cfg2 at its stated length
def test_cfg2_1000_step_rollout_n504(dev, live504):
    """BASELINE configs[1] as worded: a 1000-step free-running rollout at N=504, full model, one GPU.
    (a) the benchmark's stationary weights: 1000 steps in one call — finite, every `edges_per_step` written, the
    neighbour density stays put — equal BITWISE to the same rollout stepped in uneven pieces (single steps, short
    and long graph-replay runs mixed), and its first 40 frames bitwise to 40 single plain launches;
    (b) the live weights of the `kernelnn_live504` golden through the reference's call sites (first step on the
    sample's own graph, then the on-device loop): the first 5 of 1000 frames are the frames `recursive_propagation`
    returns — which are held to the REFERENCE's free run — and the run stays finite while the cloud contracts from
    60k edges to the complete graph (E = N^2), i.e. the edge count sweeps its whole range inside one captured step;
    what the model does after that (it diverges around step 300) is passed on as it is."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, recursive_propagation
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, steps = 504, 10, 1000
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=1e-3, feature_gain=0.1))
    model.eval().to(dev)
    win = torch.from_numpy(syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1))
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    cap = default_edge_cap(1, N, 8.0)
    eng = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, edge_cap=cap, device=dev)
    one = eng.run(win, aa, steps).clone()
    e_one = eng.edges_per_step.clone()
    assert eng.conv_mode == "factored" and one.shape == (steps, 1, N, 3) and bool(torch.isfinite(one).all())
    e = e_one.cpu().numpy()
    assert e.min() > 0.97 * e[0] and e.max() < 1.03 * e[0] and 55000 < e[0] < 66000, (e[0], e.min(), e.max())
    assert float((one[-1] - one[0]).abs().max()) > 1e-2                                   # the frames do move
    eng.reset(win, aa)
    done = 0
    for piece in (1, 7, 1, 64, 3, 500, 8, 1, steps):
        n = min(piece, steps - done)
        if n:
            eng.step(n)
            done += n
    eng.synchronize()
    assert torch.equal(eng.frames(), one) and torch.equal(eng.edges_per_step, e_one)
    eng.close()
    plain = RolloutEngine(model, 1, N, W, 8.0, max_steps=40, edge_cap=cap, device=dev, use_graph=False)
    plain.reset(win, aa)
    for _ in range(40):
        plain.step(1)
    plain.synchronize()
    assert torch.equal(plain.frames(), one[:40]) and torch.equal(plain.edges_per_step, e_one[:40])
    plain.close()

    # (b) live weights, the reference's call sites
    z, dset, sd = live504
    live = KernelNN(*[int(v) for v in z["ctor"]])
    live.load_state_dict(sd)
    live.eval().to(dev)
    live.conv_mode = "factored"
    thr = float(z["threshold"])
    fc = recursive_propagation(live, dset, dev, num_steps=5, starting_points=[0], threshold=thr)
    api = torch.stack([f.x_position[-1] for f in fc])                                      # [5,N,3], golden-checked
    close(api[0], z["free_frames"][0], name="cfg2 live first frame vs the reference")
    s0 = dset[0]
    eng = RolloutEngine(live, 1, N, W, thr, max_steps=steps, edge_cap=N * N, device=dev)
    eng.reset(s0.x_position.unsqueeze(1), s0.x_aminoacid)
    eng.first_step_from_sample(s0.edge_index, s0.edge_attr)
    for piece in (2, 9, 88, steps):
        n = min(piece, steps - eng.steps_done)
        if n:
            eng.step(n)
    eng.synchronize()
    fr = eng.frames()
    assert fr.shape[0] == steps and torch.equal(fr[:5, 0].cpu(), api)
    e = eng.edges_per_step.cpu().numpy()
    assert e[0] == s0.edge_index.shape[1] and [int(v) for v in e[1:5]] == [f.edge_index.shape[1] for f in fc[:4]]
    # These weights contract the cloud to the complete graph (E = N^2 from step ~50 on) and then, around step 300, the
    # model itself blows up: |x| 389 -> 1,010 -> 3.6e5 -> non-finite within three steps, in gemm_mode "f32" (plain fp32
    # arithmetic) at the same step, and the oracle's next frame from the last finite window is 1e36 (checked by hand,
    # round 6).  A non-finite value is PASSED ON, as torch's relu passes it on in the reference (up to round 5 every
    # ReLU was fmaxf(v, 0), which turns a NaN into 0, and this run "stayed finite" — frames of no meaning).
    fin = torch.isfinite(fr).reshape(steps, -1).all(1).cpu().numpy()
    first_bad = int(np.argmin(fin)) if not fin.all() else steps
    assert 250 <= first_bad, first_bad
    assert (e[:first_bad] > 0).all() and e[49] > 1.5 * e[0] and e[:first_bad].max() == N * N, (e[:6], e[49], first_bad)
    if first_bad < steps:
        assert float(fr[first_bad - 1].abs().max()) > 1e4 and not fin[first_bad:].any()      # a blow-up, and it stays one
    eng.close()


# ------------------------------------------------------------------------------- a fitted capacity that is outgrown
def test_recursive_propagation_outgrows_fitted_capacity(dev, O):
    """A chain of N > 256 atoms gets an edge capacity fitted to its start window (4x its edges, not N^2).  An untrained
    model pulls the atoms together — the graph becomes complete within a step or two — and the reference simply
    builds the denser graph (graph_kernel.py:363-368).  So must `recursive_propagation` (which has no edge_cap
    argument): capacity grown, steps from the first truncated one re-run, results as the oracle's host loop."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import PairData
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, recursive_propagation
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd._lib import MdnoError
    N, W, steps, thr = 260, 4, 4, 8.0
    torch.manual_seed(5)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    with torch.no_grad():
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.eval().to(dev)
    win = syn.jitter_window(syn.chain_frame(N, seed=7), W, seed=7)
    aa = torch.from_numpy(syn.amino_acids(N, seed=7))
    s0 = O.construct_pairdata(win[:1], aa, thr)                       # a dataset sample: graph of the FIRST frame
    sample = PairData(x_aminoacid=aa, x_position=torch.from_numpy(win), y=None,
                      edge_attr=s0["edge_attr"], edge_index=s0["edge_index"])
    start = dict(x_position=torch.from_numpy(win), x_aminoacid=aa, edge_index=s0["edge_index"], edge_attr=s0["edge_attr"])
    ref = O.recursive_propagation(sd, 2, start, steps, thr, hoist=True)
    e_ref = [f["edge_index"].shape[1] for f in ref]
    e0 = s0["edge_index"].shape[1]
    assert e_ref[-1] > 6 * e0, (e0, e_ref)                            # the cloud does collapse: far beyond 4x
    fc = recursive_propagation(model, [sample], dev, num_steps=steps, starting_points=[0], threshold=thr)
    for k in range(steps):
        close(fc[k].x_position[-1], ref[k]["x_position"][-1], name=f"outgrown capacity, step {k}")
        assert fc[k].edge_index.shape[1] == e_ref[k]
    # the engine itself: same frames, capacity grown, edge counts of the re-run steps recorded
    eng = RolloutEngine(model, 1, N, W, thr, max_steps=steps, device=dev)
    eng.reset(torch.from_numpy(win), aa)
    cap0 = eng.edge_cap
    assert cap0 < N * N // 2
    eng.first_step_from_sample(sample.edge_index, sample.edge_attr)
    eng.step(steps - 1)
    eng.synchronize()
    assert eng.edge_cap > cap0 and eng.edges_per_step.cpu().tolist() == [e0] + e_ref[:-1]
    assert torch.equal(eng.frames()[:, 0].cpu(), torch.stack([f.x_position[-1] for f in fc]))
    # the growth is on record (first re-run step, old and new capacity) ...
    assert eng.regrown and eng.regrown[0][1] == cap0 and eng.regrown[-1][2] == eng.edge_cap and eng.regrown[0][0] >= 1
    eng.close()
    # ... and a per-kernel timer attached before it survives the plan that was rebuilt (it was lost and
    # read_timer failed): its records restart at the re-run
    eng = RolloutEngine(model, 1, N, W, thr, max_steps=steps, device=dev)
    eng.reset(torch.from_numpy(win), aa)
    eng.attach_timer(4096)
    eng.step(steps)
    eng.synchronize()
    assert eng.regrown
    tm = eng.read_timer()
    assert tm["nnconv"][1] > 0 and tm["nnconv"][0] > 0.0
    eng.detach_timer()
    eng.close()
    # a capacity the CALLER chose is a contract: overflow raises
    eng = RolloutEngine(model, 1, N, W, thr, max_steps=steps, edge_cap=cap0, device=dev)
    eng.reset(torch.from_numpy(win), aa)
    eng.step(steps)
    with pytest.raises(MdnoError):
        eng.synchronize()
    eng.close()


# ------------------------------------------------------------------------------- shape A, live fixture (N = 28)
@pytest.fixture(scope="module")
def live28(tmp_path_factory):
    """The reference's own KernelNN at ITS OWN BBA size (N=28, nb:1034) and the CLI model size (width 64, k=1024,
    depth 6) with live weights: 20 teacher-forced forwards + 20 free-running steps of the reference's
    recursive_propagation (oracle/gen_golden.py gen_live28).  No produced frame has a pair within 1e-3 A of the
    cutoff, so graphs must agree exactly."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    z = load_golden("kernelnn_live28.npz")
    thr, W = float(z["threshold"]), int(z["window"])
    frames = z["frames"]
    cms = [syn.contact_map(f, thr) for f in frames]
    assert [c.size for c in cms] == list(z["contact_map_len"])
    assert [cm_checksum(c) for c in cms] == list(z["contact_map_checksum"])      # the reference's own contact maps
    path = tmp_path_factory.mktemp("live28") / "traj.npz"
    write_trajectory_npz(path, frames, cms, z["amino_acids"])
    dset = ContactMapDataset(str(path), window_size=W, horizon=1)
    seed, kg, fg, kc = z["weight_gains"]
    sd = near_identity_state_dict(64, 1024, seed=int(seed), kernel_gain=float(kg), feature_gain=float(fg),
                                  kernel_to_coords=float(kc))
    for n, s_, a_ in zip([str(x) for x in z["param_names"]], z["param_sum"], z["param_abs_sum"]):
        assert float(sd[n].double().sum()) == pytest.approx(float(s_), rel=1e-12, abs=1e-12), n
        assert float(sd[n].double().abs().sum()) == pytest.approx(float(a_), rel=1e-12), n
    assert float(z["free_min_gap"].min()) > 1e-4
    return z, dset, sd


def _live28_model(z, sd, dev, gemm_mode):
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    model = KernelNN(*[int(v) for v in z["ctor"]])
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode = gemm_mode
    return model


@pytest.mark.parametrize("gemm_mode", ["split_f16", "split_bf16", "f32"])
def test_live28_teacher_forced_reference_golden(dev, live28, gemm_mode):
    """model(sample) on the reference's 20 dataset samples (the sample's own edge list: the graph of the window's
    FIRST frame, dataset.py:189-201) against the reference's forwards; latent of the first to rel-L2 <= 1e-5."""
    z, dset, sd = live28
    model = _live28_model(z, sd, dev, gemm_mode)
    lat0 = z["teacher_forced_latent0"]
    assert 0.3 < float((lat0 == 0).mean()) < 0.7 and float(np.abs(lat0).max()) < 50      # live, bounded
    for i in range(z["teacher_forced_out"].shape[0]):
        s = dset[i].to(dev)
        with torch.no_grad():
            out, lat = model(s, return_latent=True)
        close(out, z["teacher_forced_out"][i], name=f"live28 tf{i} {gemm_mode}")
        if i == 0:
            close(lat, lat0, name=f"live28 latent0 {gemm_mode}")


@pytest.mark.parametrize("members", [1, 3, 10])
@pytest.mark.parametrize("gemm_mode", ["split_f16", "split_bf16", "f32"])
def test_live28_free_run_short_chain_kernels_reference_golden(dev, live28, members, gemm_mode):
    """The short-chain kernels of DESIGN 4.8 held to the REFERENCE first-hand, at full model size.  The reference's
    free run (graph_kernel.py:396-413) from its second iteration on IS the engine's loop (graph of the newest frame):
    the engine starts from the window the reference's second iteration sees — frames 1..W-1 of the data + the
    reference's own first produced frame — and must reproduce its remaining 19 frames and their edge counts.
    1 member: `step_head_small_kernel`, `gemm_split_f16_small_kernel`, `nnconv64_colsplit_kernel<4>`, `FcTail`,
    eight steps per graph launch; 3 members: `colsplit<2>`; 10 members: the row kernels of the large shapes (the
    golden window is ONE member of the batch, the others are perturbed copies)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    z, dset, sd = live28
    model = _live28_model(z, sd, dev, gemm_mode)
    thr, W = float(z["threshold"]), int(z["window"])
    want, want_E = z["free_frames"], z["free_num_edges"]
    N = want.shape[1]
    steps = want.shape[0] - 1
    win = np.concatenate([z["frames"][1:W], want[:1]], axis=0).astype(np.float32)            # [W,N,3]
    slot = {1: 0, 3: 1, 10: 6}[members]
    wins = syn.ensemble_windows(win, members, sigma=0.1, seed0=300)                            # [M,W,N,3]
    wins[slot] = win
    tm = torch.from_numpy(np.ascontiguousarray(wins.transpose(1, 0, 2, 3)))                    # [W,M,N,3]
    aa = torch.from_numpy(z["amino_acids"])
    for use_graph in (True, False):
        eng = RolloutEngine(model, members, N, W, thr, max_steps=steps, device=dev, use_graph=use_graph)
        got = eng.run(tm, aa, steps).cpu().numpy()[:, slot]
        assert eng.conv_mode == "materialized"
        for s in range(steps):
            close(got[s], want[s + 1], name=f"live28 free step {s + 1} M={members} {gemm_mode} graph={use_graph}")
        if members == 1:
            assert eng.edges_per_step.cpu().tolist() == [int(e) for e in want_E[:steps]]
        eng.close()


@pytest.mark.parametrize("gemm_mode", ["split_f16", "split_bf16", "f32"])
def test_live28_recursive_propagation_api_reference_golden(dev, live28, gemm_mode):
    """The reference's own call, `recursive_propagation(model, dataset, device, 20, [0])`, at its own BBA size:
    all 20 frames, every returned PairData's edge list (bit-exact: checksum of the reference's) and window."""
    from molecular_dynamics_neural_operator_amd.graph_kernel import recursive_propagation
    z, dset, sd = live28
    model = _live28_model(z, sd, dev, gemm_mode)
    want = z["free_frames"]
    fc = recursive_propagation(model, dset, dev, num_steps=want.shape[0], starting_points=[0],
                               threshold=float(z["threshold"]))
    assert len(fc) == want.shape[0]
    W = int(z["window"])
    for s, f in enumerate(fc):
        assert not f.x_position.is_cuda and tuple(f.x_position.shape) == (W, want.shape[1], 3)
        close(f.x_position[-1], want[s], name=f"live28 API step {s} {gemm_mode}")
        if s > 0:
            assert torch.equal(f.x_position[:-1], fc[s - 1].x_position[1:])                    # the window slides
        assert f.edge_index.shape[1] == int(z["free_num_edges"][s])
        flat = np.concatenate([f.edge_index[0].numpy(), f.edge_index[1].numpy()])
        assert cm_checksum(flat) == z["free_edge_checksum"][s], s
        assert torch.equal(f.edge_attr, torch.cat([f.x_position[-1][f.edge_index[0]], f.x_position[-1][f.edge_index[1]]], 1))


# ------------------------------------------------------------------------------- propogate (nb:336-358)
def test_propogate_notebook_loop_and_mse(dev, O, tmp_path):
    """The notebook's rollout: window 1, start at dataset[0], per-step MSE against dataset[i+1]
    (bba_analysis.ipynb:336-358), against the oracle's loop with the same bookkeeping."""
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNNNotebook, propogate
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    z = load_golden("rollout_20.npz")
    thr = float(z["threshold"])
    path = tmp_path / "traj.npz"
    write_golden_trajectory(path, z)
    dset = ContactMapDataset(str(path), window_size=1, horizon=1)
    sd = {k: v for k, v in near_identity_state_dict(64, 128, seed=2, kernel_gain=2e-2, feature_gain=0.1,
                                                    kernel_to_coords=1.0).items()
          if not k.startswith(("lstm", "conv2"))}
    model = KernelNNNotebook(64, 128, 4, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.to(dev)
    steps = 6
    fc, metrics = propogate(model, dset, dev, steps, threshold=thr)
    assert len(fc) == steps and list(metrics) == ["mse"] and len(metrics["mse"]) == steps
    cur = O.dataset_sample(z["point_cloud"], z["contact_map"], z["amino_acids"], 0, 1, 1)
    aa = cur["x_aminoacid"]
    for i in range(steps):
        out = O.kernelnn_notebook_forward(sd, cur["x_position"], aa, cur["edge_index"], cur["edge_attr"], 4)
        truth = O.dataset_sample(z["point_cloud"], z["contact_map"], z["amino_acids"], i + 1, 1, 1)["x_position"]
        want_mse = float(((out.numpy() - truth.numpy()) ** 2).mean())
        close(fc[i].x_position[-1], out, name=f"propogate step {i}")
        assert metrics["mse"][i] == pytest.approx(want_mse, rel=1e-4)
        cur = O.construct_pairdata(out.numpy(), aa, thr)
        assert fc[i].edge_index.shape[1] == cur["edge_index"].shape[1]
        assert not fc[i].x_position.is_cuda


# ------------------------------------------------------------------------------- shape A: short chains
def test_shape_a_short_chain_paths_do_not_change_a_bit(dev):
    """N = 28, full model (k = 1024, depth 6): one member runs the short-chain kernels (graph + prologue in one
    launch, few-row GEMMs with 32- and 128-row tiles, conv split four ways by output columns, fc2 and the step
    advance inside the last conv application, eight steps per graph launch), three members the two-way split, ten
    members the kernels of the large shapes.  A member's trajectory must be the same bits in all three, and under
    graph replay as under plain launches."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, steps = 28, 10, 13
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=0.02, feature_gain=0.1, kernel_to_coords=1.0))
    model.eval().to(dev)
    base = syn.jitter_window(syn.chain_frame(N, seed=1), W, seed=1)
    wins = syn.ensemble_windows(base, 10, sigma=0.1, seed0=100)                                   # [10,W,N,3]
    tm = torch.from_numpy(np.ascontiguousarray(wins.transpose(1, 0, 2, 3)))                        # [W,10,N,3]
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    e10 = RolloutEngine(model, 10, N, W, 8.0, max_steps=steps, device=dev)
    assert e10.conv_mode == "materialized"
    big = e10.run(tm, aa, steps).clone()
    assert bool(torch.isfinite(big).all()) and float((big[-1] - big[0]).abs().max()) > 1e-3       # the frames move
    e3 = RolloutEngine(model, 3, N, W, 8.0, max_steps=steps, device=dev)
    three = e3.run(tm[:, 4:7].contiguous(), aa, steps).clone()
    assert torch.equal(three, big[:, 4:7])
    e1g = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, device=dev, use_graph=True)
    e1e = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, device=dev, use_graph=False)
    for m in (0, 5, 9):
        solo = e1g.run(tm[:, m:m + 1].contiguous(), aa, steps).clone()
        assert torch.equal(solo[:, 0], big[:, m]), m
        assert torch.equal(e1e.run(tm[:, m:m + 1].contiguous(), aa, steps), solo)
        assert torch.equal(e1g.edges_per_step, e1e.edges_per_step) and int(e1g.edges_per_step.min()) > 0
    # stepping in pieces (8-step graph launches and single steps mixed) == one call
    e1g.reset(tm[:, 9:10].contiguous(), aa)
    e1g.step(3)
    e1g.step(9)
    e1g.step(1)
    e1g.synchronize()
    assert torch.equal(e1g.frames()[:, 0], big[:, 9])


# ------------------------------------------------------------------------------- cfg3: 8 members x 504
def test_cfg3_eight_members_n504_full_model(dev):
    """One GPU's share of the 64-member ensemble at 8 GPUs (BASELINE configs[2]): 8 x N=504, full model,
    3 steps.  factored == materialized within tolerance; a member inside the batch of 8 == the member
    alone, bitwise, in BOTH formulations (the materialized kernel changes its launch shape at 4,096 rows)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, M, steps = 504, 10, 8, 3
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=0.02, feature_gain=0.1, kernel_to_coords=1.0))
    model.eval().to(dev)
    base = syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1)
    wins = syn.ensemble_windows(base, M, sigma=0.1, seed0=100)                        # cfg3 seeds 100..
    tm = torch.from_numpy(np.ascontiguousarray(wins.transpose(1, 0, 2, 3)))          # [W,M,N,3]
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    res = {}
    for conv in ("factored", "materialized"):
        model.conv_mode = conv
        eng = RolloutEngine(model, M, N, W, 8.0, max_steps=steps, edge_cap=default_edge_cap(M, N, 8.0), device=dev)
        assert eng.conv_mode == conv
        res[conv] = (eng.run(tm, aa, steps).clone(), eng.edges_per_step[:steps].clone())
        eng.close()
        solo = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, edge_cap=default_edge_cap(1, N, 8.0), device=dev)
        for m in (0, 5):
            alone = solo.run(tm[:, m:m + 1].contiguous(), aa, steps)
            assert torch.equal(alone[:, 0], res[conv][0][:, m]), (conv, m)
        solo.close()
        del eng, solo
        torch.cuda.empty_cache()
    assert torch.equal(res["factored"][1], res["materialized"][1])
    assert int(res["factored"][1].min()) > 8 * 55_000
    close(res["factored"][0], res["materialized"][0], name="cfg3 factored vs materialized")
    # "auto" takes the factored path for this share, as it does for one member
    model.conv_mode = "auto"
    eng = RolloutEngine(model, M, N, W, 8.0, max_steps=1, edge_cap=default_edge_cap(M, N, 8.0), device=dev)
    assert eng.conv_mode == "factored"
    eng.close()


def test_materialized_conv_launch_shapes_agree_bitwise(dev):
    """nnconv64_row_kernel<16> (fewer than 4,096 rows) and <4> (more) add a row's edges in the same 16
    chains: the same rows give the same bits in either launch (batch invariance)."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(11)
    n, E = 700, 30_000
    ei = torch.randint(0, n, (2, E), generator=gen)
    ei[1, :900] = 3                                   # a hub
    x = torch.randn(n, 64, generator=gen).to(dev)
    w_e = (torch.randn(E, 4096, generator=gen) * 0.05).to(dev)
    root = (torch.randn(64, 64, generator=gen) * 0.1).to(dev)
    bias = torch.randn(64, generator=gen).to(dev)
    g = ops.coo_to_csr(ei.to(dev), n)
    w_csr = w_e[g.perm[:E].long()].contiguous()
    y16 = ops.nnconv(x, g, w_csr, root, bias, "mean", relu=True)
    # the same graph padded with 4,000 isolated rows -> the 4-wave launch
    big = 4700
    g2 = ops.coo_to_csr(ei.to(dev), big)
    x2 = torch.cat([x, torch.zeros(big - n, 64, device=dev)])
    y4 = ops.nnconv(x2, g2, w_csr, root, bias, "mean", relu=True)
    assert torch.equal(y4[:n], y16)


# ------------------------------------------------------------------------------- cfg5: 50k atoms
def test_cfg5_50k_atoms_one_factored_step(dev):
    """BASELINE configs[4]: N=50,000 uniform box, r=10 A, full model, ONE factored step (W_e would be
    298 GB).  Graph: bit-exact against f64 numpy on sampled rows, symmetric, E and degrees in the expected
    range.  Conv: the factored forward of the whole box against the materialized formulation evaluated
    on the first ~4k rows (edge-MLP + conv kernel on that row slice, same x), layer by layer."""
    from molecular_dynamics_neural_operator_amd import _lib, ops, synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, thr = 50_000, 2, 10.0
    frame = syn.box_frame(N, seed=3)
    pos = torch.from_numpy(frame).to(dev)
    g = ops.radius_graph(pos, N, thr, edge_cap=N * 420)
    E = g.edge_count()
    assert int(g.status.item()) == 0
    rp = g.row_ptr.cpu().numpy().astype(np.int64)
    deg = np.diff(rp)
    assert 17_500_000 < E < 18_800_000 and 330 < E / N < 380              # E ~ 18.2M, mean degree ~364
    assert 30 <= deg.min() and deg.max() < 768
    src = g.src[:E].cpu().numpy()
    rng = np.random.default_rng(0)
    p64 = frame.astype(np.float64)
    for r in np.concatenate([[0, N - 1], rng.integers(0, N, 300)]):
        d = np.sqrt(((p64 - p64[r]) ** 2).sum(1))
        assert np.array_equal(src[rp[r]:rp[r + 1]], np.nonzero(d < thr)[0]), r
    # full model forward, factored (explicitly: nothing else fits)
    # (coordinates reach +-40 A here, 4x the N=504 box: a 10x smaller kernel gain keeps the step ~1 A)
    sd = near_identity_state_dict(64, 1024, seed=0, kernel_gain=2e-3, feature_gain=0.1, kernel_to_coords=1.0)
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.eval().to(dev)
    win = torch.from_numpy(syn.jitter_window(frame, W, sigma=0.01, seed=3)).to(dev)
    win[-1] = pos                                                          # the graph's frame is the last one
    aa = torch.from_numpy(syn.amino_acids(N, seed=3)).to(dev)
    out, lat = ops.kernelnn_forward(model.param_pack(dev, conv_mode="factored"), win.unsqueeze(1), aa, g, edge_pos=pos,
                                    return_latent=True)
    assert bool(torch.isfinite(out).all())
    disp = (out - pos).norm(dim=1)
    print(f"cfg5: E={E} mean degree {E / N:.1f} max {deg.max()}  mean displacement {float(disp.mean()):.3f} A")
    assert 1e-3 < float(disp.mean()) < 5.0                                 # the kernel integral moves atoms, boundedly
    del out, lat
    # ONE conv application both ways: the notebook-era model at depth 1 is  fc1 -> relu(conv1) -> fc2,
    # its latent is the conv's output and its conv input x0 comes from the node prologue.  Factored over
    # the whole box vs the materialized formulation (edge-MLP -> W_e -> conv kernel) on the first rows
    # holding ~1.4M edges (23 GB of fp32 W_e).
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNNNotebook
    nb = KernelNNNotebook(64, 1024, 1, 6, 7, 3, 20, 4)
    nb.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("lstm", "conv2"))})
    nb.eval().to(dev)
    one = pos.reshape(1, 1, N, 3)
    _, y_fac = ops.kernelnn_forward(nb.param_pack(dev, conv_mode="factored"), one, aa, g, edge_pos=pos,
                                    return_latent=True)
    x0 = ops.node_prologue(nb.param_pack(dev, conv_mode="materialized"), one, aa)
    rows = int(np.searchsorted(rp, 1_400_000))
    Es = int(rp[rows])
    sl = ops.CSRGraph(g.row_ptr[:rows + 1].contiguous(), g.src[:Es].contiguous(), g.dst[:Es].contiguous(),
                      torch.tensor([Es], dtype=torch.int32, device=dev), Es, None, None)
    w_e = ops.edge_mlp(nb.conv1.net.hip_weights(), 6, 1024, 4096, sl, edge_pos=pos)
    y_mat = torch.empty(rows, 64, device=dev)
    lib = _lib.load()
    _lib.check(lib.mdno_nnconv_fwd(x0.data_ptr(), sl.row_ptr.data_ptr(), sl.src.data_ptr(), rows, w_e.data_ptr(),
                                   nb.conv1.root.data_ptr(), nb.conv1.bias.data_ptr(), 64, 64, 1, 1, y_mat.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "mdno_nnconv_fwd")
    torch.cuda.synchronize()
    close(y_fac[:rows], y_mat, name=f"cfg5 factored vs materialized conv, first {rows} rows / {Es} edges")
    # ... and against the CPU ORACLE (the reference's formulas: per-edge MLP -> W_e -> x_j . W_e -> mean -> root,
    # bias, ReLU) on 300 destination rows sampled over the whole box (~109k edges, 1.1 TFLOP on the host): the
    # two device paths above share their fp16-plane weight images, the oracle shares nothing with them
    from conftest import host_cores
    from oracle import graph_kernel_oracle as O
    threads = torch.get_num_threads()
    torch.set_num_threads(host_cores())
    try:
        sdc = {k: v.detach().cpu() for k, v in nb.state_dict().items()}
        posc, aac = torch.from_numpy(frame), aa.cpu()
        x0c = torch.relu(torch.nn.functional.linear(torch.cat((sdc["emb.weight"][aac], posc), dim=1),
                                                    sdc["fc1.weight"], sdc["fc1.bias"]))
        close(x0, x0c, name="cfg5 node prologue vs oracle formulas")
        sampled = np.unique(np.concatenate([[0, N - 1, int(deg.argmax()), int(deg.argmin())], rng.integers(0, N, 300)]))
        y_fac_c = y_fac.cpu()
        for lo in range(0, len(sampled), 60):
            part = sampled[lo:lo + 60]
            s_idx = np.concatenate([src[rp[r]:rp[r + 1]] for r in part]).astype(np.int64)
            d_idx = np.concatenate([np.full(rp[r + 1] - rp[r], r, dtype=np.int64) for r in part])
            ei = torch.from_numpy(np.stack([s_idx, d_idx]))
            attr = torch.cat([posc[ei[0]], posc[ei[1]]], dim=1)                    # [pos[src], pos[dst]] (graph_kernel.py:372-379)
            w_e_c = O.edge_mlp(attr, sdc, "conv1.net.")
            y_c = torch.relu(O.nnconv_apply(x0c, ei, w_e_c, sdc["conv1.root"], sdc["conv1.bias"], "mean"))
            close(y_fac_c[part], y_c[part], name=f"cfg5 factored conv vs CPU oracle, rows {lo}..{lo + len(part) - 1} of {len(sampled)} sampled")
    finally:
        torch.set_num_threads(threads)


# ------------------------------------------------------------------------------- cfg4: training step
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cfg4_training_step_full_size(dev, O, tmp_path, precision):
    """BASELINE configs[3] at the reference's CLI sizes: k=1024, depth 6, batch 128 of N=28 samples, in the
    fp32 path and in bf16 (what BASELINE names).  The whole batch runs (finite loss, every gradient present);
    because samples are independent problems, the first 4 outputs of the batch equal the 4-sample sub-batch's,
    and the sub-batch's loss and gradients are checked against the oracle's train step in fp64 — for bf16 with
    the storage roundings of h1, h2, W_e, dW_e emulated at the points where the device rounds
    (tests/bf16_replica.py), so that every parameter is held to 1e-3 (fp32: 3e-3 against the un-rounded
    replica, measured 1e-6)."""
    from test_gpu_training import _replica_loss, rel_err
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import collate
    base = syn.chain_frame(28, seed=0)
    traj = syn.ou_trajectory(base, 150, sigma=0.15, theta=0.2, seed=2)
    cms = [syn.contact_map(f, 8.0) for f in traj]
    path = tmp_path / "t.npz"
    write_trajectory_npz(path, traj, cms, syn.amino_acids(28, seed=0))
    dset = ContactMapDataset(str(path), window_size=10, horizon=1)
    B = 128
    batch = [dset[i] for i in range(B)]
    torch.manual_seed(3)
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    with torch.no_grad():                      # keep activations O(1) through 12 random-init layers
        for p_ in model.conv1.net.layers[4].parameters():
            p_.mul_(0.2)
    model.to(dev).train()
    model.train_precision = precision
    out = model(batch)
    assert out.shape == (B * 28, 3)
    y = torch.cat([s.y for s in batch]).to(dev)
    loss = LpLoss(size_average=False)(out.view(B, -1), y.view(B, -1))
    loss.backward()
    assert bool(torch.isfinite(loss)) and all(p_.grad is not None and bool(torch.isfinite(p_.grad).all())
                                               for p_ in model.parameters())
    full_out = out.detach().clone()
    model.zero_grad()
    sub = batch[:4]
    out4 = model(sub)
    close(out4, full_out[:4 * 28], name="cfg4 sub-batch of 4 vs the same samples inside the batch of 128")
    loss4 = LpLoss(size_average=False)(out4.view(4, -1), y[:4 * 28].view(4, -1))
    loss4.backward()
    want_loss, want_out, want_grads = _replica_loss(model, O, sub, bf16=precision == "bf16")
    assert abs(float(loss4) - want_loss) < 1e-4 * abs(want_loss)
    assert rel_err(out4, want_out) < 1e-4
    worst = {}
    for name, p_ in model.named_parameters():
        worst[name] = rel_err(p_.grad, want_grads[name])
    print(f"cfg4 {precision} gradient rel errors:", {k: f"{v:.1e}" for k, v in worst.items()})
    for name, e in worst.items():
        assert e < (1e-3 if precision == "bf16" else 3e-3), (name, e)


# ------------------------------------------------------------------------------- loader / validation
def test_reference_checkpoint_forward_on_device(dev, tmp_path):
    """A best.pt-shaped checkpoint written by the reference's own (DataParallel-wrapped) KernelNN loads
    through load_reference_checkpoint and reproduces the reference's forward."""
    from molecular_dynamics_neural_operator_amd import load_reference_checkpoint
    from molecular_dynamics_neural_operator_amd.dataset import PairData
    z = load_golden("checkpoint_best_pt.npz")
    msd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("msd/")}
    ckpt = {"epoch": int(z["epoch"]), "model_state_dict": msd, "optimizer_state_dict": {}, "scheduler_state_dict": {}}
    p = tmp_path / "best.pt"
    torch.save(ckpt, p)
    model, meta = load_reference_checkpoint(p, depth=int(z["ctor"][2]))
    assert meta["epoch"] == 3 and meta["variant"] == "intree"
    model.eval().to(dev)
    pd = PairData(t(z["x_aminoacid"]), t(z["x_position"]), None, t(z["edge_attr"]), t(z["edge_index"])).to(dev)
    with torch.no_grad():
        close(model(pd), z["out"], name="checkpoint forward")


def test_bad_indices_raise_like_the_reference(dev):
    """An amino-acid id == num_embeddings (1-indexed labels) and an edge to a node that does not exist:
    the reference's nn.Embedding / index_select raise IndexError; so does the HIP path."""
    from molecular_dynamics_neural_operator_amd import MdnoError, ops
    from molecular_dynamics_neural_operator_amd.dataset import PairData
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    z = load_golden("kernelnn_small.npz")
    model = KernelNN(*[int(v) for v in z["ctor"]]).eval().to(dev)
    aa = t(z["x_aminoacid"]).clone()
    aa[5] = 20                                                     # == num_embeddings
    pd = PairData(aa, t(z["x_position"]), None, t(z["edge_attr"]), t(z["edge_index"])).to(dev)
    with pytest.raises(IndexError):
        with torch.no_grad():
            model(pd)
    ei = t(z["edge_index"]).clone()
    ei[0, 7] = 28                                                  # source == num_nodes
    with pytest.raises(IndexError):
        ops.coo_to_csr(ei.to(dev), 28)
    ei = t(z["edge_index"]).clone()
    ei[1, 3] = -1
    with pytest.raises(MdnoError):
        ops.coo_to_csr(ei.to(dev), 28)
    g = ops.coo_to_csr(ei.to(dev), 28, validate=False)             # deferred check: flagged, in bounds
    assert int(g.status.item()) != 0 and int(g.src.max()) < 28 and int(g.row_ptr[-1]) == ei.shape[1]
    pd = PairData(t(z["x_aminoacid"]), t(z["x_position"]), None, t(z["edge_attr"]), ei).to(dev)
    with pytest.raises(IndexError):
        with torch.no_grad():
            model(pd)


def test_coo_to_csr_big_rows_and_determinism(dev):
    """Rows above the per-wave limit (2,048 entries) take the workgroup-per-row rank sort; the result
    is the stable order by destination whatever order the atomic slots were handed out in."""
    from molecular_dynamics_neural_operator_amd import ops
    gen = torch.Generator().manual_seed(9)
    n, E = 300, 40_000
    ei = torch.randint(0, n, (2, E), generator=gen)
    ei[1, :9000] = 17                                  # 9,000+ in-edges: big-row kernel
    ei[1, 9000:12000] = 250                            # 3,000+: big-row kernel, one partial tile
    g = ops.coo_to_csr(ei.to(dev), n)
    order = torch.sort(ei[1], stable=True).indices
    assert torch.equal(g.perm[:E].cpu().long(), order)
    assert torch.equal(g.src[:E].cpu().long(), ei[0][order])
    g2 = ops.coo_to_csr(ei.to(dev), n)
    assert torch.equal(g.perm, g2.perm)


# ------------------------------------------------------------------------------- multi-process
def _run_bench(args, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` typed as the driver types it: the parent starts two fresh workers,
    relays rank 0's line and exits 0.  On a one-GPU box the ranks share the card and the collective
    runs over gloo (rehearsal); with two GPUs the same command runs RCCL."""
    line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--total-members", "4", "--skip-roofline",
                       "--skip-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["total_members"] == 4 and line["config"]["members_this_rank"] == 2


def test_bench_multi_rank_line_schema(dev):
    """The N > 1 bench line as the driver's scaling series reads it, rehearsed with FOUR ranks on whatever GPUs the
    box has (a one-GPU box: the ranks share the card, gloo; the box allows six GPU processes at once, this test
    holds one, so four ranks is what fits — the 8-rank rendezvous itself is rehearsed on CPU in test_dist_gloo.py):
    default strong scaling over the ensemble, an uneven shard (7 members over 4 ranks), `multi_gpu_timing` per
    rank, rank 0's `roofline`, and NONE of the single-GPU legs (they belong to the N = 1 line)."""
    line = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--total-members", "7", "--atoms", "28", "--chain",
                       "--kernel-width", "128", "--depth", "2"])
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["value"] > 0 and line["higher_is_better"]
    assert line["metric"] == "rolled-out MD frames/sec" and line["unit"] == "frames/s" and line["dtype"] == "f32"
    assert line["steps"] == 2 and line["warmup"] == 1 and line["vs_baseline"] is None
    cfg = line["config"]
    assert cfg["total_members"] == 7 and cfg["members_this_rank"] == 2 and cfg["members_per_gpu_max"] == 2
    assert "ensemble-sharded x4" in cfg["parallelism"] and ("gloo" in cfg["parallelism"] or "nccl" in cfg["parallelism"])
    mg = line["multi_gpu_timing"]
    assert len(mg["per_rank_steps_ms"]) == 4 and len(mg["per_rank_gather_ms"]) == 4 and all(v > 0 for v in mg["per_rank_steps_ms"])
    assert mg["gathered_bytes_per_rank"] == 2 * 2 * 28 * 3 * 4
    # value = all members' frames / the slowest rank's wall time (within the all-gather's overhead of it)
    assert line["value"] <= 7 * 2 / (max(mg["per_rank_steps_ms"]) * 1e-3) * 1.001
    assert line["roofline"] is not None and line["roofline"]["bound"] in ("hbm", "mfma") and line["roofline"]["frac"] > 0
    for leg in ("ensemble64_single_gpu", "baseline_1gpu_same_workload", "cfg2_1000_steps", "cfg4_training", "cfg5_shape_c",
                "shape_A", "cpu_baseline", "other_conv_mode"):
        assert line[leg] is None, leg
    # default member count for N > 1 is BASELINE configs[2]'s 64 (checked without running it: the parser's rule)
    import bench
    a = bench.parse(["--gpus", "8"])
    assert a.total_members is None and bench.ENSEMBLE_MEMBERS == 64


def _torchrun_bench(args, env_extra, timeout=600):
    """bench.py under the DRIVER's launcher and command line: python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ... (ranks share the one card here,
    so the collective runs over gloo; with one GPU per rank the same command runs RCCL)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    n = args[args.index("--gpus") + 1]
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(REPO / "bench.py")] + args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_under_torch_distributed_run(dev):
    """The scaling series' own command line (one process per rank started by torch.distributed.run, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from its environment): two ranks, the 4-member ensemble split 2 + 2, one JSON line from rank 0
    on stdout; and the same launch with rank 1 failing at start-up ends non-zero in well under a minute instead of
    sitting in the rendezvous."""
    import time
    gloo = {} if torch.cuda.device_count() >= 2 else {"MDNO_BENCH_BACKEND": "gloo"}
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--total-members", "4", "--skip-roofline", "--skip-cpu-baseline"]
    r = _torchrun_bench(args, gloo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["total_members"] == 4
    assert line["config"]["members_this_rank"] == 2 and len(line["multi_gpu_timing"]["per_rank_steps_ms"]) == 2
    t0 = time.time()
    r = _torchrun_bench(args, dict(gloo, MDNO_BENCH_FAIL_RANK="1"), timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0 and dt < 90.0, (r.returncode, dt, r.stderr[-2000:])
    assert "MDNO_BENCH_FAIL_RANK=1" in r.stderr
    assert not [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{") and '"value"' in ln]


def test_bench_rank_failure_is_fast_and_loud(dev):
    """`python bench.py --gpus 2` with rank 1 raising right after set_device (MDNO_BENCH_FAIL_RANK=1) while rank 0
    waits for it in the rendezvous: the launcher polls every rank from the start, stops rank 0 after its grace period
    and returns non-zero with ONE JSON error line — in under a minute, not the process group's timeout (the driver's
    600 s limit would otherwise expire with nothing written)."""
    import time
    env = dict(os.environ, MDNO_BENCH_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--total-members", "4", "--skip-roofline", "--skip-cpu-baseline"], capture_output=True, text=True,
                       env=env, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0 and dt < 60.0, (r.returncode, dt, r.stderr[-2000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert "rank 1" in line["error"] and line["rank_exit_codes"][1] not in (0, None) and line["n_gpus"] == 2
    assert "value" not in line
    assert "MDNO_BENCH_FAIL_RANK=1" in r.stderr


NCCL_CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MDNO_REPO"])
from molecular_dynamics_neural_operator_amd.rollout import gather_trajectories, shard_members
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
total, T, N = 5, 3, 7                                   # uneven: rank 0 holds 3 members, rank 1 holds 2
full = torch.arange(T * total * N * 3, dtype=torch.float32).reshape(T, total, N, 3)
mine = full[:, shard_members(total, rank, world)].contiguous().to(dev)
out = gather_trajectories(mine, total)
torch.cuda.synchronize()
assert out.is_cuda and torch.equal(out.cpu(), full), "gathered trajectories differ"
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL wants one device per rank)")
def test_gather_trajectories_over_rccl_two_gpus(tmp_path):
    """gather_trajectories over the nccl (= RCCL) backend in two spawned children, one GPU each,
    against the single-process result."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(NCCL_CHILD)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MDNO_REPO=str(REPO))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and "ok" in out, err[-2000:]


NCCL_WORLD1_CHILD = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MDNO_REPO"])
from molecular_dynamics_neural_operator_amd.rollout import gather_trajectories
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mine = torch.arange(3 * 5 * 7 * 3, dtype=torch.float32).reshape(3, 5, 7, 3).to(dev)
out = gather_trajectories(mine, 5)
torch.cuda.synchronize()
assert dist.get_backend() == "nccl" and out.is_cuda and out.data_ptr() != mine.data_ptr()
assert torch.equal(out, mine), "gathered trajectories differ"
t = torch.tensor([2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 2.5
dist.barrier()
dist.destroy_process_group()
print("ok world1 nccl %.1fs" % (time.time() - t0))
"""


def test_gather_trajectories_over_rccl_world_size_one(dev, tmp_path):
    """RCCL executes on the one GPU a builder's box has: a fresh child process makes a world-size-1 nccl group
    (communicator init through librccl on gfx950), runs rollout.gather_trajectories — the one collective of the
    N > 1 path — on a [3,5,7,3] device tensor, the all_reduce(MAX) and the barrier bench.py's timed region uses,
    and tears the group down.  (Two ranks on one card are refused by RCCL, so this is as far as one GPU goes; the
    two-GPU test below runs where two exist.)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "child1.py"
    script.write_text(NCCL_WORLD1_CHILD)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               MDNO_REPO=str(REPO))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok world1 nccl" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    print(r.stdout.strip())


def test_bench_forced_world1_group_runs_the_collective_path(dev):
    """`python bench.py --gpus 1` with MDNO_BENCH_FORCE_DIST=1: the timed region's barrier / all-gather / max over
    ranks run through a world-size-1 nccl group; the line says so and carries the collective's timings."""
    env = dict(os.environ, MDNO_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MDNO_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--total-members", "2", "--atoms", "60", "--kernel-width", "128", "--depth", "2",
                        "--skip-roofline", "--skip-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    mg = line["multi_gpu_timing"]
    assert line["n_gpus"] == 1 and line["value"] > 0 and "nccl world 1, forced" in line["config"]["parallelism"]
    assert mg["backend"] == "nccl" and mg["world_size"] == 1 and mg["init_process_group_s"] > 0
    assert mg["gathered_bytes_per_rank"] == 3 * 2 * 60 * 3 * 4 and len(mg["per_rank_gather_ms"]) == 1


def test_rollout_fallback_counters(dev, live504):
    """RolloutEngine.fallback_counts(): zero for the live 504-atom model in both conv formulations (every product of the
    rollout on two fp16 planes), counted per run (a second run starts from zero); non-zero — and the frames still the
    bf16-plane mode's to fp32 rounding — when the start window is scaled so that the hidden activations leave the fp16
    planes' range; grouped engines add their groups' counters."""
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import GroupedRolloutEngine, RolloutEngine
    z, dset, sd = live504
    model = KernelNN(*[int(v) for v in z["ctor"]])
    model.load_state_dict(sd)
    model.eval().to(dev)
    thr, W, N = float(z["threshold"]), int(z["window"]), 504
    s = dset[0]
    for conv_mode in ("factored", "materialized"):
        model.conv_mode = conv_mode
        eng = RolloutEngine(model, 1, N, W, thr, max_steps=4, device=dev)
        assert eng.fallback_counts() == {k: 0 for k in RolloutEngine.FALLBACK_KEYS}          # no plan yet
        eng.run(s.x_position, s.x_aminoacid, 3)
        assert all(v == 0 for v in eng.fallback_counts().values()), (conv_mode, eng.fallback_counts())
        eng.close()
    # the same function with the hidden activations 4096 x larger (layer 0 and layer 1's bias x 2^12, the last layer's
    # weight x 2^-12: relu is positively homogeneous, powers of two are exact): |H| leaves K1's range [2^-7, 2047) and
    # |h1| may pass fp16's 65504 in the edge-MLP — the rollout runs on bf16 planes where it has to, says so, and gives
    # the frames of the unscaled model
    model.conv_mode = "factored"
    want = RolloutEngine(model, 1, N, W, thr, max_steps=2, device=dev)
    ref_frames = want.run(s.x_position, s.x_aminoacid, 2).clone()
    big_sd = {k: v.clone() for k, v in sd.items()}
    for conv in ("conv1", "conv2"):
        big_sd[f"{conv}.net.layers.0.weight"] *= 4096.0
        big_sd[f"{conv}.net.layers.0.bias"] *= 4096.0
        big_sd[f"{conv}.net.layers.2.bias"] *= 4096.0
        big_sd[f"{conv}.net.layers.4.weight"] /= 4096.0
    big_model = KernelNN(*[int(v) for v in z["ctor"]])
    big_model.load_state_dict(big_sd)
    big_model.eval().to(dev)
    big_model.conv_mode = "factored"
    eng = RolloutEngine(big_model, 1, N, W, thr, max_steps=2, device=dev)
    got = eng.run(s.x_position, s.x_aminoacid, 2).clone()
    c = eng.fallback_counts()
    print("H x 4096:", c)
    # (nearly) every K1 workgroup reran: 2 steps x 12 applications x 504 destinations x 4 column blocks (k = 1024)
    assert 0.9 * 2 * 12 * N * 4 <= c["conv_k1_workgroups_rerun_bf16"] <= 2 * 12 * N * 4 and c["conv_destinations_unscaled"] == 0, c
    close(got, ref_frames, name="H out of the fp16 planes' range vs the unscaled model")
    ref = want
    # per run: a run inside the ranges after one outside reads zero again
    eng2 = RolloutEngine(model, 1, N, W, thr, max_steps=2, device=dev)
    eng2.run(s.x_position, s.x_aminoacid, 1)
    assert all(v == 0 for v in eng2.fallback_counts().values())
    grp = GroupedRolloutEngine(big_model, 2, N, W, thr, max_steps=1, device=dev, groups=2)
    grp.run(torch.stack([s.x_position, s.x_position], dim=1), s.x_aminoacid, 1)
    cg = grp.fallback_counts()
    one = RolloutEngine(big_model, 1, N, W, thr, max_steps=1, device=dev)
    one.run(s.x_position, s.x_aminoacid, 1)
    assert cg == {k: 2 * v for k, v in one.fallback_counts().items()} and cg["conv_k1_workgroups_rerun_bf16"] > 0, (cg, one.fallback_counts())
    for e in (eng, ref, eng2, grp, one):
        e.close()


def test_untied_conv2_kernel_is_evaluated_separately(dev, O):
    """The reference ties conv1.net and conv2.net to ONE module (graph_kernel.py:271-273) and the library
    evaluates the edge-MLP once per forward on that ground.  A state_dict whose conv2.net.* differ (loaded
    from elsewhere) must still give the reference's function: the second block then gets its own
    evaluation — forward with explicit edges, and both conv formulations of the on-device rollout, vs the
    oracle (which evaluates `conv + ".net."` per block)."""
    import copy
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, construct_pairdata
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    sd = near_identity_state_dict(64, 128, seed=4, kernel_gain=2e-2, feature_gain=0.1, kernel_to_coords=1.0)
    other = near_identity_state_dict(64, 128, seed=5, kernel_gain=3e-2, feature_gain=0.1, kernel_to_coords=1.0)
    sd = dict(sd)
    for k in list(sd):
        if k.startswith("conv2.net."):
            sd[k] = other[k].clone()
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    model.conv2.net = copy.deepcopy(model.conv1.net)          # untie, then load the two different kernels
    model.load_state_dict(sd)
    model.eval().to(dev)
    assert not model.param_pack(dev).shared_kernel
    N, W, steps = 70, 10, 3
    win = syn.jitter_window(syn.box_frame(N, seed=6), W, seed=6)
    aa = torch.from_numpy(syn.amino_acids(N, seed=6))
    ref_pd = O.construct_pairdata(win, aa, 8.0)
    want = O.kernelnn_forward(sd, ref_pd["x_position"], aa, ref_pd["edge_index"], ref_pd["edge_attr"], 2)
    with torch.no_grad():
        close(model(construct_pairdata(win, aa, 8.0)), want, name="untied conv2, forward")
    fc = O.recursive_propagation(sd, 2, ref_pd, steps, 8.0, hoist=True)
    ref = np.stack([f["x_position"][-1].numpy() for f in fc])
    for conv in ("factored", "materialized"):
        model.conv_mode = conv
        eng = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, device=dev)
        close(eng.run(torch.from_numpy(win), aa, steps)[:, 0], ref, name=f"untied conv2, rollout {conv}")


# ------------------------------------------------------------------------------- ragged sizes
@pytest.mark.parametrize("gemm_mode", ["split_f16", "split_bf16"])
@pytest.mark.parametrize("conv_mode", ["factored", "materialized"])
def test_ragged_member_sizes_free_run_vs_oracle(dev, O, conv_mode, gemm_mode):
    """Member sizes that are no multiple of any tile (1 atom, 2, 31, 65, 129 and 200 atoms — the last with
    sources of more than 128 edges, i.e. a second row tile in the factored conv), one and two members: three
    free-running steps at width 64 against the oracle's host loop, member by member, with the edge counts of
    every step."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    W, steps = 4, 3
    sd = near_identity_state_dict(64, 128, seed=21, kernel_gain=1e-2, feature_gain=0.1, kernel_to_coords=1.0)
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode, model.conv_mode = gemm_mode, conv_mode
    for N, M in ((1, 1), (2, 2), (31, 1), (65, 2), (129, 1), (200, 2)):
        base = syn.jitter_window(syn.box_frame(N, seed=30 + N), W, seed=30 + N)
        wins = syn.ensemble_windows(base, M, sigma=0.2, seed0=500 + N)                 # [M, W, N, 3]
        aa = torch.from_numpy(syn.amino_acids(N, seed=N))
        eng = RolloutEngine(model, M, N, W, 8.0, max_steps=steps, device=dev)
        traj = eng.run(torch.from_numpy(np.ascontiguousarray(wins.transpose(1, 0, 2, 3))), aa, steps).cpu().numpy()
        edges = np.zeros(steps, dtype=np.int64)
        for m in range(M):
            s0 = O.construct_pairdata(wins[m], aa, 8.0)
            fc = O.recursive_propagation(sd, 2, s0, steps, 8.0, hoist=True)
            ref = np.stack([f["x_position"][-1].numpy() for f in fc])
            np.testing.assert_allclose(traj[:, m], ref, rtol=1e-4, atol=1e-4 * max(np.abs(ref).max(), 1.0),
                                       err_msg=f"N={N} member {m}")
            edges += np.array([s0["edge_index"].shape[1]] + [f["edge_index"].shape[1] for f in fc[:-1]])
        assert eng.edges_per_step.cpu().tolist() == edges.tolist(), (N, M)


def test_member_groups_resolve_auto_once_for_the_shard(dev):
    """conv_mode "auto" in a GroupedRolloutEngine is decided on the graph of the WHOLE shard and given to every group
    (ADVICE r4): two members of very different density — a 504-atom box at 0.1 atoms/A^3 (~118 neighbours: factored on
    its own) and the same atoms spread 3x wider (~6 neighbours: materialized on its own) — as two groups of one take
    the formulation one engine holding both picks, and give its frames (to fp32 rounding: see the class docstring)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import GroupedRolloutEngine, RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, steps = 504, 4, 2
    model = KernelNN(64, 256, 2, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 256, seed=2, kernel_gain=2e-2, feature_gain=0.2, kernel_to_coords=1.0))
    model.eval().to(dev)
    assert model.conv_mode == "auto"
    dense = syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1)
    centre = dense.mean(axis=(0, 1), keepdims=True)
    sparse = (dense - centre) * 3.0 + centre
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    alone = {}
    for name, w in (("dense", dense), ("sparse", sparse)):
        e = RolloutEngine(model, 1, N, W, 8.0, max_steps=steps, device=dev)
        e.reset(torch.from_numpy(w), aa)
        alone[name] = e.conv_mode
        e.close()
    assert alone == {"dense": "factored", "sparse": "materialized"}, alone
    tm = torch.from_numpy(np.ascontiguousarray(np.stack([dense, sparse], axis=1)))              # [W, 2, N, 3]
    one = RolloutEngine(model, 2, N, W, 8.0, max_steps=steps, device=dev)
    want = one.run(tm, aa, steps).clone()
    grp = GroupedRolloutEngine(model, 2, N, W, 8.0, max_steps=steps, device=dev, groups=2)
    got = grp.run(tm, aa, steps)
    assert grp.conv_mode == one.conv_mode and {e.conv_mode for e in grp.engines} == {one.conv_mode}
    for m in range(2):
        d = (got[:, m] - want[:, m]).abs().max().item()
        print(f"member {m}: max |grouped - one engine| = {d:.3e}, edges/step {[int(e.edges_per_step[0]) for e in grp.engines]}")
    # same formulation, same sums; the sparse member alone takes the edge-MLP's bf16 plane products (no activation of
    # its few edges reaches 2^-10: split_layout.h) where the shard as a whole takes the fp16 ones: fp32 rounding apart
    assert torch.equal(got[:, 0], want[:, 0])
    torch.testing.assert_close(got, want, rtol=2e-6, atol=2e-6 * float(want.abs().max()))
    one.close()
    grp.close()


def test_short_chain_plan_replays_eight_steps_per_launch(dev):
    """The N = 28 configuration's plan holds the eight-steps-per-launch graph (a failed capture used to be
    swallowed and would only have shown as a slower shape A); a 504-atom plan replays single steps, a plan without
    graphs reports 0."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    model = KernelNN(64, 128, 2, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 128, seed=2, kernel_gain=2e-2, feature_gain=0.2, kernel_to_coords=1.0))
    model.eval().to(dev)
    for N, use_graph, want in ((28, True, 8), (504, True, 1), (28, False, 0)):
        frame = syn.chain_frame(N, seed=1) if N == 28 else syn.box_frame(N, seed=1)
        eng = RolloutEngine(model, 1, N, 4, 8.0, max_steps=16, device=dev, use_graph=use_graph)
        assert eng.steps_per_launch == 0                        # no plan before reset
        eng.reset(torch.from_numpy(syn.jitter_window(frame, 4, seed=1)), torch.from_numpy(syn.amino_acids(N, seed=1)))
        assert eng.steps_per_launch == want, (N, use_graph, eng.steps_per_launch)
        eng.close()


def test_member_groups_on_concurrent_streams_give_the_same_frames(dev):
    """rollout.GroupedRolloutEngine: 5 members of 504 atoms as 2 groups (3 + 2) and as 5 groups of one, each group an
    engine on its own stream with its own captured step, against ONE engine holding all five: frames bitwise equal
    (members never interact), edge counts equal; per-member residue types; more groups than members are clamped."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import GroupedRolloutEngine, RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, M, steps = 504, 10, 5, 4
    model = KernelNN(64, 256, 2, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(64, 256, seed=2, kernel_gain=2e-2, feature_gain=0.2, kernel_to_coords=1.0))
    model.eval().to(dev)
    base = syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1)
    tm = torch.from_numpy(np.ascontiguousarray(syn.ensemble_windows(base, M, sigma=0.1, seed0=100).transpose(1, 0, 2, 3)))
    aa = torch.cat([torch.from_numpy(syn.amino_acids(N, seed=m)) for m in range(M)])       # a different sequence per member
    one = RolloutEngine(model, M, N, W, 8.0, max_steps=steps, device=dev)
    want = one.run(tm, aa, steps).clone()
    want_edges = one.edges_per_step.clone()
    for g in (2, 5, 9):
        eng = GroupedRolloutEngine(model, M, N, W, 8.0, max_steps=steps, device=dev, groups=g)
        assert len(eng.engines) == min(g, M) and eng.bounds[0][0] == 0 and eng.bounds[-1][1] == M
        got = eng.run(tm, aa, steps)
        assert got.shape == want.shape and torch.equal(got, want), g
        assert torch.equal(eng.edges_per_step, want_edges)
        assert torch.equal(eng.traj[:W].cpu(), tm) and eng.steps_done == steps
        eng.close()
