"""Synthetic weight sets for `KernelNN` under the reference's state_dict key names.

No trained checkpoint exists offline (the notebook's `best.pt` is an absolute path on the
authors' server, bba_analysis.ipynb raw line 80).  A randomly initialised `KernelNN` maps every
atom to almost the same point, so a free-running rollout (graph_kernel.py:396-413: next frame =
model output) collapses to the complete graph after one step and no longer has the BBA
neighbour density the benchmark is quoted on.

`near_identity_state_dict` builds weights for the *unchanged* architecture and the *unchanged*
update rule for which ``model(window) = last frame + small position-dependent displacement``:

  LSTM (graph_kernel.py:264)   W_hh = 0, input gate open, forget gate shut, output gate open,
                               candidate rows = w*I  ->  h_W = tanh(tanh(w * p_last)) ~ w * p_last
  lstm_fc (:265)               (1/w) * I             ->  x ~ p_last
  fc1 (:269)                   channels 0..2 = +x, 3..5 = -x (so ReLU keeps both signs), the
                               remaining channels are small random features of [emb, x]
  conv1/conv2 (:272-273)       root = I on channels 0..5 (+ small random block elsewhere),
                               shared edge-MLP with the last layer scaled by `kernel_gain` and its
                               outputs into channels 0..5 scaled by `kernel_to_coords` (default 0)
  fc2 (:275)                   out = ch[0:3] - ch[3:6] + feature_gain * random(ch[6:])

Every layer is still evaluated at full cost (the arithmetic does not depend on the values), so
frames/s measured with these weights is the cost of the reference architecture at a stationary
neighbour density.  Used by bench.py, the rollout fixtures and the tests; it is weight *data*,
not a code path — the engine has no notion of it.
"""
from __future__ import annotations

from collections import OrderedDict

import torch


def near_identity_state_dict(width: int = 64, ker_width: int = 1024, ker_in: int = 6,
                             num_embeddings: int = 20, embedding_dim: int = 4,
                             seed: int = 0, w_lstm: float = 1e-4,
                             kernel_gain: float = 1e-3, feature_gain: float = 1e-2,
                             kernel_to_coords: float = 0.0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    if width < 7:
        raise ValueError("near-identity weights need width >= 7 (6 coordinate channels + >=1 feature)")
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale

    H = 3
    sd = OrderedDict()
    # --- LSTM(3,3): gate order i, f, g, o (torch.nn.LSTM)
    w_ih = torch.zeros(4 * H, H, dtype=torch.float64)
    w_ih[2 * H:3 * H] = torch.eye(H, dtype=torch.float64) * w_lstm
    b_ih = torch.zeros(4 * H, dtype=torch.float64)
    b_ih[0:H] = 20.0          # input gate  -> 1
    b_ih[H:2 * H] = -20.0     # forget gate -> 0  (cell = candidate of the current frame only)
    b_ih[3 * H:4 * H] = 20.0  # output gate -> 1
    sd["lstm.weight_ih_l0"] = w_ih
    sd["lstm.weight_hh_l0"] = torch.zeros(4 * H, H, dtype=torch.float64)
    sd["lstm.bias_ih_l0"] = b_ih
    sd["lstm.bias_hh_l0"] = torch.zeros(4 * H, dtype=torch.float64)
    sd["lstm_fc.weight"] = torch.eye(H, dtype=torch.float64) / w_lstm
    sd["lstm_fc.bias"] = torch.zeros(H, dtype=torch.float64)
    sd["emb.weight"] = rnd(num_embeddings, embedding_dim)
    # --- fc1: [emb(4), x(3)] -> width
    in_w = embedding_dim + H
    fc1 = torch.zeros(width, in_w, dtype=torch.float64)
    fc1[0:3, embedding_dim:] = torch.eye(H, dtype=torch.float64)
    fc1[3:6, embedding_dim:] = -torch.eye(H, dtype=torch.float64)
    fc1[6:, :embedding_dim] = rnd(width - 6, embedding_dim, scale=0.5)
    fc1[6:, embedding_dim:] = rnd(width - 6, H, scale=0.05)
    sd["fc1.weight"] = fc1
    b1 = torch.zeros(width, dtype=torch.float64)
    b1[6:] = rnd(width - 6, scale=0.1)
    sd["fc1.bias"] = b1
    # --- shared edge-MLP (torch.nn.Linear default-like bounds), last layer scaled down
    def lin(out_f, in_f, gain=1.0):
        bound = 1.0 / (in_f ** 0.5)
        return rnd(out_f, in_f, scale=bound * gain), rnd(out_f, scale=bound * gain)

    k0w, k0b = lin(ker_width, ker_in, gain=0.25)     # inputs are raw coordinates (|p| ~ 10 A)
    k2w, k2b = lin(ker_width, ker_width)
    k4w, k4b = lin(width * width, ker_width, gain=kernel_gain)
    # W_e[:, i, o] for o < 6 (kernel integral INTO the coordinate channels) is scaled by
    # `kernel_to_coords`.  0 (default, the benchmark): coordinates travel through `root` untouched and
    # the per-step displacement is the fc2 read-out of the features alone, so the neighbour density
    # stays put over 1000 steps.  1: neighbours push each other around (the graph changes within a
    # few steps; used by the rollout fixtures and tests).
    coord_out = (torch.arange(width * width) % width) < 6
    k4w[coord_out] *= kernel_to_coords
    k4b[coord_out] *= kernel_to_coords
    for conv in ("conv1", "conv2"):
        root = torch.zeros(width, width, dtype=torch.float64)
        root[0:6, 0:6] = torch.eye(6, dtype=torch.float64)
        root[6:, 6:] = rnd(width - 6, width - 6, scale=1.0 / (width ** 0.5))
        sd[f"{conv}.root"] = root
        b = torch.zeros(width, dtype=torch.float64)
        b[6:] = rnd(width - 6, scale=0.05)
        sd[f"{conv}.bias"] = b
        sd[f"{conv}.net.layers.0.weight"], sd[f"{conv}.net.layers.0.bias"] = k0w, k0b
        sd[f"{conv}.net.layers.2.weight"], sd[f"{conv}.net.layers.2.bias"] = k2w, k2b
        sd[f"{conv}.net.layers.4.weight"], sd[f"{conv}.net.layers.4.bias"] = k4w, k4b
    # --- fc2: coordinates back out + a small readout of the feature channels
    fc2 = torch.zeros(H, width, dtype=torch.float64)
    fc2[:, 0:3] = torch.eye(H, dtype=torch.float64)
    fc2[:, 3:6] = -torch.eye(H, dtype=torch.float64)
    fc2[:, 6:] = rnd(H, width - 6, scale=feature_gain / (width ** 0.5))
    sd["fc2.weight"] = fc2
    sd["fc2.bias"] = torch.zeros(H, dtype=torch.float64)
    return OrderedDict((k, v.to(dtype).contiguous()) for k, v in sd.items())
