"""bf16 training GEMMs alone: correctness against fp64 of the rounded operands on ragged shapes, then time at
the cfg4 batch shapes (E = 43,712 edges, k = 1024, 64*64 = 4096 kernel outputs).  python scripts/micro/bench_train_gemm.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from molecular_dynamics_neural_operator_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
bf = lambda t: t.to(torch.bfloat16)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


ok = True
for rows, n, k in ((300, 256, 64), (1000, 256, 1024), (517, 4096, 128), (257, 512, 96), (5000, 1024, 4096), (1, 256, 64)):
    a, w, b = torch.randn(rows, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    want = torch.nn.functional.linear(bf(a).double(), bf(w).double(), b.double())
    ab = bf(a).to(dev)
    e1 = rel(ops.linear_bf16(ab, w.to(dev), b.to(dev), relu=False, out_bf16=False), want)
    e2 = rel(ops.linear_bf16(ab, w.to(dev), b.to(dev), relu=True, out_bf16=True).float(), want.relu())
    e3 = rel(ops.linear_bf16(ab, w.to(dev), None, relu=False, out_bf16=False), torch.nn.functional.linear(bf(a).double(), bf(w).double()))
    good = e1 < 3e-6 and e2 < 4e-3 and e3 < 3e-6
    ok &= good
    print(f"NT rows={rows} n={n} k={k}: fp32-out {e1:.2e}  bf16-out+relu {e2:.2e}  no-bias {e3:.2e}  {'ok' if good else 'FAIL'}", flush=True)
# masked variant: bf16((y > 0) * (g . w_t^T)) == the two-step path, bit for bit
for rows, n, k in ((300, 256, 64), (1000, 256, 1024), (517, 1024, 4096)):
    gq, w, yv = torch.randn(rows, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(rows, n, generator=g)
    yv = bf(torch.relu(yv)).to(dev)
    two = ops.relu_bwd_bf16(ops.linear_bf16(bf(gq).to(dev), w.to(dev), None, out_bf16=False), yv, out_bf16=True)
    one = ops.linear_bf16_relu_bwd(bf(gq).to(dev), w.to(dev), yv)
    good = torch.equal(one, two)
    ok &= good
    print(f"NT masked rows={rows} n={n} k={k}: equal to GEMM + relu_bwd: {good}", flush=True)
for rows, n1, n2 in ((5000, 128, 256), (333, 1024, 128), (4097, 256, 4096), (31, 256, 256), (70000, 512, 256), (1000, 4096, 1024),
                     (43712, 1024, 1024)):
    a, b = torch.randn(rows, n1, generator=g), torch.randn(rows, n2, generator=g)
    got = ops.gemm_atb_bf16(bf(a).to(dev), bf(b).to(dev))
    e = rel(got, bf(a).double().t() @ bf(b).double())
    same = torch.equal(got, ops.gemm_atb_bf16(bf(a).to(dev), bf(b).to(dev)))
    good = e < 3e-6 and same
    ok &= good
    print(f"TN rows={rows} n1={n1} n2={n2}: {e:.2e} repeatable={same} {'ok' if good else 'FAIL'}", flush=True)
import os
if not ok and not os.environ.get("MDNO_SKIP_CHECK"):
    sys.exit(1)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


E, k = 43712, 1024
h = bf(torch.randn(E, k, generator=g)).to(dev)
big = bf(torch.randn(E, 4096, generator=g)).to(dev)
w1 = (torch.randn(k, k, generator=g) / 32).to(dev)
w2 = (torch.randn(4096, k, generator=g) / 32).to(dev)
w2t = w2.t().contiguous()
b1, b2 = torch.randn(k, generator=g).to(dev), torch.randn(4096, generator=g).to(dev)
for name, fn, fl in (
        ("NT h1.W1^T   [E,1024]x[1024,1024] relu bf16-out", lambda: ops.linear_bf16(h, w1, b1, relu=True, out_bf16=True), 2.0 * E * k * k),
        ("NT h2.W2^T   [E,1024]x[1024,4096]      bf16-out", lambda: ops.linear_bf16(h, w2, b2, relu=False, out_bf16=True), 2.0 * E * k * 4096),
        ("NT dWe.W2    [E,4096]x[4096,1024]      fp32-out", lambda: ops.linear_bf16(big, w2t, None, relu=False, out_bf16=False), 2.0 * E * k * 4096),
        ("NT gz2.W1    [E,1024]x[1024,1024]      fp32-out", lambda: ops.linear_bf16(h, w1, None, relu=False, out_bf16=False), 2.0 * E * k * k),
        ("NT dWe.W2 masked [E,4096]x[4096,1024] bf16-out", lambda: ops.linear_bf16_relu_bwd(big, w2t, h), 2.0 * E * k * 4096),
        ("TN dWe^T.h2  [E,4096]^T x [E,1024]", lambda: ops.gemm_atb_bf16(big, h), 2.0 * E * k * 4096),
        ("TN gz2^T.h1  [E,1024]^T x [E,1024]", lambda: ops.gemm_atb_bf16(h, h), 2.0 * E * k * k)):
    ms = timeit(fn)
    print(f"{name}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 2500:.3f} of 2.5 PF)", flush=True)
