"""Pins the oracle (oracle/graph_kernel_oracle.py) against golden vectors emitted by the reference's
own code (oracle/gen_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_state_dict, load_golden
from oracle import graph_kernel_oracle as O


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("aggr", ["mean", "add"])
def test_nnconv_small(aggr):
    z = load_golden(f"nnconv_small_{aggr}.npz")
    sd = golden_state_dict(z)
    w_e = O.edge_mlp(t(z["edge_attr"]), sd, "net.")
    assert torch.equal(w_e, t(z["w_e"]))  # same torch CPU ops on the same machine image: bitwise
    y = O.nnconv_forward(t(z["x"]), t(z["edge_index"]), t(z["edge_attr"]), sd, "", aggr)
    torch.testing.assert_close(y, t(z["y"]), rtol=1e-6, atol=1e-6)


def test_kernelnn_small():
    z = load_golden("kernelnn_small.npz")
    sd = golden_state_dict(z)
    depth = int(z["ctor"][2])
    for hoist in (False, True):
        out, lat = O.kernelnn_forward(sd, t(z["x_position"]), t(z["x_aminoacid"]), t(z["edge_index"]),
                                      t(z["edge_attr"]), depth, return_latent=True, hoist=hoist)
        torch.testing.assert_close(out, t(z["out"]), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(lat, t(z["latent"]), rtol=1e-6, atol=1e-6)


def test_pairdata_graph():
    z = load_golden("pairdata_graph.npz")
    thr = float(z["threshold"])
    pd = O.construct_pairdata(z["x_position"], None, thr)
    assert np.array_equal(pd["edge_index"].numpy(), z["edge_index"])
    assert np.array_equal(pd["edge_attr"].numpy(), z["edge_attr"])
    pd = O.construct_pairdata(z["box_position"], None, thr)
    assert np.array_equal(pd["edge_index"].numpy(), z["box_edge_index"])
    assert np.array_equal(pd["edge_attr"].numpy(), z["box_edge_attr"])
    # self-loops are kept and the COO is row-major
    ei = z["edge_index"]
    assert (ei[0] == ei[1]).sum() == z["x_position"].shape[1]
    assert np.all(np.diff(ei[0] * 10_000 + ei[1]) > 0)


def test_dataset_sample_and_len():
    z = load_golden("rollout_20.npz")
    W, h = int(z["window"]), int(z["horizon"])
    assert O.dataset_len(z["point_cloud"].shape[0], W, h) == int(z["dataset_len"])
    s = O.dataset_sample(z["point_cloud"], z["contact_map"], z["amino_acids"], 3, W, h)
    assert np.array_equal(s["x_position"].numpy(), z["sample3_x_position"])
    assert np.array_equal(s["y"].numpy(), z["sample3_y"])
    assert np.array_equal(s["edge_index"].numpy(), z["sample3_edge_index"])
    assert np.array_equal(s["edge_attr"].numpy(), z["sample3_edge_attr"])


def test_rollout_teacher_forced_and_free():
    z = load_golden("rollout_20.npz")
    W, h, thr = int(z["window"]), int(z["horizon"]), float(z["threshold"])
    sd_tf = golden_state_dict(z, "tf.")
    for i in range(20):
        s = O.dataset_sample(z["point_cloud"], z["contact_map"], z["amino_acids"], i, W, h)
        out = O.kernelnn_forward(sd_tf, s["x_position"], s["x_aminoacid"], s["edge_index"], s["edge_attr"], 2)
        torch.testing.assert_close(out, t(z["teacher_forced_out"][i]), rtol=1e-6, atol=1e-6)
        assert np.array_equal(s["y"].numpy(), z["teacher_forced_y"][i])
    sd_fr = golden_state_dict(z, "free.")
    s0 = O.dataset_sample(z["point_cloud"], z["contact_map"], z["amino_acids"], 0, W, h)
    fc = O.recursive_propagation(sd_fr, 2, s0, 20, thr)
    assert [f["edge_index"].shape[1] for f in fc] == list(z["free_num_edges"])
    free = np.stack([f["x_position"][-1].numpy() for f in fc])
    np.testing.assert_allclose(free, z["free_frames"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(fc[-1]["edge_index"].numpy(), z["free_edge_index_last"])


def test_lploss():
    z = load_golden("lploss.npz")
    x, y = t(z["x"]), t(z["y"])
    torch.testing.assert_close(O.lp_loss_rel(x, y, size_average=False), t(z["rel_sum"]))
    torch.testing.assert_close(O.lp_loss_rel(x, y, size_average=True), t(z["rel_mean"]))
    torch.testing.assert_close(O.lp_loss_rel(x, y, reduction=False), t(z["rel_none"]))
    torch.testing.assert_close(O.lp_loss_abs(x, y), t(z["abs_mean"]))


def test_reference_init_order_and_full_forward():
    """Same seed -> same parameters as the reference's KernelNN.__init__ (checksums), and the
    full-size (w=64, k=1024, depth 6) forward on the N=28 chain matches the reference output."""
    z = load_golden("kernelnn_full_seeded.npz")
    c = [int(v) for v in z["ctor"]]
    sd = O.reference_init_state_dict(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], seed=int(z["seed"]))
    names = [str(n) for n in z["param_names"]]
    assert sorted(sd.keys()) == names
    for n, s, a in zip(names, z["param_sum"], z["param_abs_sum"]):
        assert float(sd[n].double().sum()) == pytest.approx(float(s), rel=1e-12, abs=1e-12), n
        assert float(sd[n].double().abs().sum()) == pytest.approx(float(a), rel=1e-12), n
    out, lat = O.kernelnn_forward(sd, t(z["x_position"]), t(z["x_aminoacid"]), t(z["edge_index"]),
                                  t(z["edge_attr"]), c[2], return_latent=True, hoist=True)
    scale = float(np.abs(z["out"]).max())
    torch.testing.assert_close(out, t(z["out"]), rtol=1e-5, atol=1e-5 * scale)
    torch.testing.assert_close(lat, t(z["latent"]), rtol=1e-5, atol=1e-5 * float(np.abs(z["latent"]).max()))


def test_live504_teacher_forced_and_first_free_step():
    """N=504, width 64, k=1024, depth 6, live bounded activations (kernelnn_live504.npz): the oracle
    against the reference's forward on dataset sample 0 (latent + output) and, from there, against the
    first free-running frame and its edge count.  (One edge-MLP evaluation per forward — hoist=True
    gives the same bits as the reference's 12 — keeps this at ~20 s of CPU.)"""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    z = load_golden("kernelnn_live504.npz")
    thr, W = float(z["threshold"]), int(z["window"])
    frames = z["frames"]
    seed, kg, fg, kc = z["weight_gains"]
    sd = near_identity_state_dict(64, 1024, seed=int(seed), kernel_gain=float(kg), feature_gain=float(fg),
                                  kernel_to_coords=float(kc))
    for n, s_, a_ in zip([str(x) for x in z["param_names"]], z["param_sum"], z["param_abs_sum"]):
        assert float(sd[n].double().sum()) == pytest.approx(float(s_), rel=1e-12, abs=1e-12), n
    cm0 = syn.contact_map(frames[0], thr)
    assert cm0.size == int(z["contact_map_len"][0])
    pc = np.transpose(frames, (0, 2, 1))
    s0 = O.dataset_sample(pc, [cm0] * len(frames), z["amino_acids"], 0, W, 1)     # only sample 0 is used
    out, lat = O.kernelnn_forward(sd, s0["x_position"], s0["x_aminoacid"], s0["edge_index"], s0["edge_attr"], 6,
                                  return_latent=True, hoist=True)
    torch.testing.assert_close(lat, t(z["teacher_forced_latent0"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out, t(z["teacher_forced_out"][0]), rtol=1e-5, atol=1e-5)
    # the reference's free run starts from the same sample: its first frame is this forward's output
    torch.testing.assert_close(out, t(z["free_frames"][0]), rtol=1e-5, atol=1e-5)
    nxt = O.construct_pairdata(np.vstack([s0["x_position"].numpy()[1:], z["free_frames"][0][None]]), s0["x_aminoacid"], thr)
    assert nxt["edge_index"].shape[1] == int(z["free_num_edges"][0])


def _cm_checksum(cm):
    c = np.asarray(cm, dtype=np.uint64)
    w = (np.arange(c.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return np.uint64((c * w).sum())


def test_live28_teacher_forced_and_free_run():
    """The reference's own BBA size (N=28, nb:1034) at the CLI model size (width 64, k=1024, depth 6) with live
    weights (kernelnn_live28.npz): the oracle against ALL 20 teacher-forced forwards of the reference on its
    ContactMapDataset samples and ALL 20 free-running steps of its recursive_propagation — frames, edge counts and
    the edge lists themselves (checksums) per step."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    z = load_golden("kernelnn_live28.npz")
    thr, W = float(z["threshold"]), int(z["window"])
    frames = z["frames"]
    seed, kg, fg, kc = z["weight_gains"]
    sd = near_identity_state_dict(64, 1024, seed=int(seed), kernel_gain=float(kg), feature_gain=float(fg),
                                  kernel_to_coords=float(kc))
    for n, s_, a_ in zip([str(x) for x in z["param_names"]], z["param_sum"], z["param_abs_sum"]):
        assert float(sd[n].double().sum()) == pytest.approx(float(s_), rel=1e-12, abs=1e-12), n
    cms = [syn.contact_map(f, thr) for f in frames]
    assert [c.size for c in cms] == list(z["contact_map_len"])
    assert [_cm_checksum(c) for c in cms] == list(z["contact_map_checksum"])
    pc = np.transpose(frames, (0, 2, 1))
    depth = int(z["ctor"][2])
    lat0 = z["teacher_forced_latent0"]
    assert 0.3 < float((lat0 == 0).mean()) < 0.7 and float(np.abs(lat0).max()) < 50          # live, bounded
    for i in range(z["teacher_forced_out"].shape[0]):
        s = O.dataset_sample(pc, cms, z["amino_acids"], i, W, 1)
        out, lat = O.kernelnn_forward(sd, s["x_position"], s["x_aminoacid"], s["edge_index"], s["edge_attr"], depth,
                                      return_latent=True, hoist=True)
        torch.testing.assert_close(out, t(z["teacher_forced_out"][i]), rtol=1e-5, atol=1e-5)
        if i == 0:
            torch.testing.assert_close(lat, t(lat0), rtol=1e-5, atol=1e-5)
    assert float(z["free_min_gap"].min()) > 1e-4       # no pair near the cutoff: the edge lists must be equal
    fc = O.recursive_propagation(sd, depth, O.dataset_sample(pc, cms, z["amino_acids"], 0, W, 1),
                                 z["free_frames"].shape[0], thr, hoist=True)
    for k, f in enumerate(fc):
        torch.testing.assert_close(f["x_position"][-1], t(z["free_frames"][k]), rtol=1e-4, atol=1e-4)
        assert f["edge_index"].shape[1] == int(z["free_num_edges"][k])
        flat = np.concatenate([f["edge_index"][0].numpy(), f["edge_index"][1].numpy()])
        assert _cm_checksum(flat) == z["free_edge_checksum"][k], k


def test_checkpoint_fixture_forward():
    """best.pt-shaped fixture written by the reference's DataParallel-wrapped KernelNN
    (graph_kernel.py:528, :630-639): the oracle strips `module.` and reproduces the forward."""
    z = load_golden("checkpoint_best_pt.npz")
    sd = {k[4:]: t(z[k]) for k in z.files if k.startswith("msd/")}
    assert all(k.startswith("module.") for k in sd)
    out = O.kernelnn_forward(sd, t(z["x_position"]), t(z["x_aminoacid"]), t(z["edge_index"]), t(z["edge_attr"]),
                             int(z["ctor"][2]))
    torch.testing.assert_close(out, t(z["out"]), rtol=1e-6, atol=1e-6)


def _train_step_case(z, tag):
    """(state_dict before the step, sample, depth) of one case of train_step_b1.npz."""
    ctor = [int(v) for v in z[f"{tag}.ctor"]]
    if f"{tag}.p.fc1.weight" in z.files:
        sd = golden_state_dict(z, f"{tag}.p.")
    else:       # regenerated from the seed in the reference's RNG-draw order, then checked against the checksums
        sd = O.reference_init_state_dict(*ctor, seed=int(z[f"{tag}.seed"]))
        for k in list(sd):
            if k.endswith("net.layers.4.weight") or k.endswith("net.layers.4.bias"):
                sd[k] = sd[k] * float(z[f"{tag}.last_layer_scale"])
        for n, s_, a_ in zip([str(x) for x in z[f"{tag}.param_names"]], z[f"{tag}.param_sum"], z[f"{tag}.param_abs_sum"]):
            assert float(sd[n].double().sum()) == pytest.approx(float(s_), rel=1e-12, abs=1e-12), n
            assert float(sd[n].double().abs().sum()) == pytest.approx(float(a_), rel=1e-12), n
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # (dataset.py hands out transposed views; C order here)
    sample = dict(x_position=c(z["x_position"]), x_aminoacid=c(z["x_aminoacid"]), y=c(z["y"]),
                  edge_index=c(z["edge_index"]), edge_attr=c(z["edge_attr"]))
    return sd, sample, ctor[2]


@pytest.mark.parametrize("tag", ["s8", "w64"])
def test_train_step_reference_golden(tag):
    """Loss and every parameter gradient of the oracle's train_step against ONE iteration of the reference's
    own train() at batch size 1 (oracle/gen_golden.py gen_train_step; graph_kernel.py:445-474): in fp32 (the
    reference's arithmetic) and in fp64 (what the GPU tests use as the exact answer): rel. L2 <= 2e-6 per
    parameter either way (measured 1-4e-7: fp32 rounding of the reference's own run).  The Adam update (:541-543) applied to the
    golden gradients reproduces the reference's parameters after the step."""
    z = load_golden("train_step_b1.npz")
    sd, sample, depth = _train_step_case(z, tag)
    want = golden_state_dict(z, f"{tag}.g.")
    assert set(want) == {k for k in sd if not k.startswith("conv2.net.")}      # (named_parameters lists a shared module once)
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 2e-6)):
        loss, out, grads = O.train_step(sd, [sample], depth, dtype=dtype)
        assert loss == pytest.approx(float(z[f"{tag}.loss"]), rel=1e-5)
        worst = {}
        for n, g in want.items():
            worst[n] = float((grads[n].double() - g.double()).norm() / g.double().norm().clamp_min(1e-30))
        print(tag, dtype, {k: f"{v:.1e}" for k, v in worst.items()})
        assert max(worst.values()) < tol, worst
    # one Adam step with the reference's settings on the golden gradients -> the reference's updated parameters
    # (conv1.net / conv2.net are one module there: one parameter, one update)
    after = golden_state_dict(z, f"{tag}.a.")
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if not k.startswith("conv2.net.")}
    opt = torch.optim.Adam(params.values(), lr=float(z["lr"]), weight_decay=float(z["weight_decay"]))
    for k, p_ in params.items():
        p_.grad = want[k].clone()
    opt.step()
    for k, v in after.items():
        src = params[k.replace("conv2.net.", "conv1.net.")]
        torch.testing.assert_close(src.detach(), v, rtol=1e-6, atol=1e-7)
