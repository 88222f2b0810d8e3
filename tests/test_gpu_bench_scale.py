"""Parity at the scale the benchmark runs ("What's weak" 1): B = 32 samples of 256^2, i.e.
P = 2 097 152 grid points per layer -- every saved [P,256] tensor is exactly 2^31 bytes, the weight-gradient kernel
folds 256 slabs, the synthesis grid has 8192 tiles per channel block.  The B = 1 path is pinned against the reference
by the golden case ffno2d_cfg3_256 (and B = 8 by ffno2d_cfg3_256_b8); every op of the hot path is per-sample, so

  * per-sample outputs and input gradients of ONE B = 32 call must equal 32 B = 1 calls, and
  * the weight gradients of the B = 32 call must equal the fixed-order sum of the 32 per-sample gradients.

Plus: the fused FeedForward forward / backward chain / streaming weight gradient called directly at
P = 2 097 152 + 7 (a partial last tile behind 65536 full ones) against the per-GEMM leg.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG3 = dict(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True,
            n_ff_layers=3, layer_norm=True, dropout=0.0)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_b32_equals_32_single_sample_calls(gpu_device):
    from models.ffno import FFNO2D
    from utils.synthetic import random_fields
    B, R = 32, 256
    torch.manual_seed(0)
    model = FFNO2D(**CFG3).to(gpu_device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    x = random_fields(B, R, 2, seed=21).to(gpu_device)
    # samples of different magnitude: per-line / per-point scaling must not leak between samples
    x = x * torch.logspace(-2, 2, B, device=gpu_device).view(B, 1, 1, 1)
    # (not zero-mean: the bias-like gradients of the last layer are w_out[c] * sum(cot), which a standardised
    #  cotangent would make pure rounding noise)
    cot = (random_fields(B, R, 2, seed=22) * 0.7 + 0.4).to(gpu_device)

    xb = x.clone().requires_grad_(True)
    out_b = model(xb)
    out_b.backward(cot)
    dx_b = xb.grad.clone()
    gw_b = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    out_b = out_b.detach()
    torch.cuda.synchronize()

    gw_sum = [torch.zeros_like(g, dtype=torch.float64) for g in gw_b]
    worst_out = worst_dx = 0.0
    for i in range(B):
        xi = x[i:i + 1].clone().requires_grad_(True)
        oi = model(xi)
        oi.backward(cot[i:i + 1])
        worst_out = max(worst_out, _rel(out_b[i:i + 1], oi.detach()))
        worst_dx = max(worst_dx, _rel(dx_b[i:i + 1], xi.grad))
        for acc, p in zip(gw_sum, params):
            acc += p.grad.double()
            p.grad = None
    print(f"\n[b32] worst per-sample out {worst_out:.2e}, dx {worst_dx:.2e}")
    assert worst_out < 1e-6, worst_out
    assert worst_dx < 1e-6, worst_dx
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    errs = {n: _rel(g, s) for n, g, s in zip(names, gw_b, gw_sum)}
    worst = max(errs, key=errs.get)
    print(f"[b32] worst weight-gradient rel-L2 vs the sum of per-sample gradients: {worst} {errs[worst]:.2e}")
    bad = {n: e for n, e in errs.items() if not e < 2e-5}
    assert not bad, bad


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_fused_feedforward_at_bench_scale_equals_gemm_path(gpu_device, dropout):
    """rpde_feedforward_fwd / bwd (k_ff3_fwd_h2, k_ff3_bwd_h2, k_wgrad_h2 with 256 slabs) at P = 2^21 + 7"""
    from models.custom_layer import FeedForward
    P = 32 * 256 * 256 + 7
    torch.manual_seed(5)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=dropout).to(gpu_device).train()
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(P, 64, generator=g).to(gpu_device)
    res = torch.randn(P, 64, generator=g).to(gpu_device)
    cot = torch.randn(P, 64, generator=g).to(gpu_device)

    def run():
        xs = x.clone().requires_grad_(True)
        torch.manual_seed(11)                                   # same dropout seed draw in both legs
        out = ff(xs, residual=res)
        out.backward(cot)
        r = [out.detach(), xs.grad] + [p.grad.clone() for p in ff.parameters()]
        for p in ff.parameters():
            p.grad = None
        return r

    from rpde import _lib
    assert _lib.load().rpde_feedforward_is_fused(64, 4, 3, P) == 1
    fused = run()
    with _env(RPDE_FUSED_FF="0", RPDE_WGRAD_H2="0"):
        plain = run()
    names = ["out", "dx"] + [n for n, _ in ff.named_parameters()]
    for n, a, b in zip(names, fused, plain):
        assert torch.isfinite(a).all(), n
        e = _rel(a, b)
        # the tail: the last 7 points sit alone in tile 65536
        if n in ("out", "dx"):
            assert _rel(a[-7:], b[-7:]) < 2e-6, (n, "tail")
        assert e < (2e-6 if n == "out" else 2e-5), (n, e)
