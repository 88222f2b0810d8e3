"""Kernels that wait for prefetched operands with HAND-COUNTED `s_waitcnt vmcnt(N)` (DESIGN.md section 3.1:
k_dft_analysis_sq_h2, k_ff3_bwd_h2, k_conv_syn_h2) reduce in fixed order, so two runs on the same data are bitwise
equal -- unless a count is too small and a value is read before it has landed.  Each case runs twice, the second time
with a copy stream hammering HBM beside the kernels (later arrivals), and must repeat itself exactly; the value itself is
pinned by the parity tests.  profiles/soak_ff_bwd.py and profiles/soak_spectral.py are the long versions
(profiles/r03_soak.txt)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Hammer:
    def __init__(self, dev):
        self.side = torch.cuda.Stream()
        self.junk = torch.empty(32 << 20, device=dev)

    def __call__(self, rounds=20):
        with torch.cuda.stream(self.side):
            for _ in range(rounds):
                self.junk.copy_(self.junk.flip(0))


def test_fused_feedforward_backward_repeats_itself_bitwise(gpu_device):
    from models.custom_layer import FeedForward
    torch.manual_seed(0)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=0.1).to(gpu_device).train()
    hammer = _Hammer(gpu_device)
    for r in range(3):
        x = torch.randn(8, 256, 256, 64, device=gpu_device) * (10.0 ** (r - 1))
        res, g = torch.randn_like(x), torch.randn_like(x)
        runs = []
        for h in (False, True):
            torch.manual_seed(1000 + r)                   # the same dropout masks
            xs = x.clone().requires_grad_(True)
            for p in ff.parameters():
                p.grad = None
            if h:
                hammer()
            ff(xs, residual=res).backward(g)
            torch.cuda.synchronize()
            runs.append([xs.grad.clone()] + [p.grad.clone() for p in ff.parameters()])
        assert all(torch.isfinite(t).all() for t in runs[0])
        assert all(torch.equal(u, v) for u, v in zip(*runs)), r


@pytest.mark.parametrize("B,n", [(8, 256), (4, 192), (5, 160), (8, 96), (16, 64)])
def test_fused_spectral_layer_repeats_itself_bitwise(gpu_device, B, n):
    from rpde import ops
    torch.manual_seed(n)
    K = 20
    x = torch.randn(B, n, n, 64, device=gpu_device)
    wy = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    wx = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    g = torch.randn_like(x)
    hammer = _Hammer(gpu_device)
    runs = []
    for h in (False, True):
        xs, a, b = x.clone().requires_grad_(True), wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
        if h:
            hammer()
        out, skip = ops.fspectral2d(xs, a, b, K, with_skip=True)
        (out * g + skip * (0.5 * g)).sum().backward()
        torch.cuda.synchronize()
        runs.append((out.detach(), xs.grad, a.grad, b.grad))
    assert all(torch.isfinite(t).all() for t in runs[0])
    assert all(torch.equal(u, v) for u, v in zip(*runs))


def test_fno2d_evaluation_forward_repeats_itself_bitwise(gpu_device):
    """config 5's fused evaluation passes (k_conv_syn_h2: LDS-DMA a row ahead, vmcnt(8))"""
    from models.fno import FNO2d
    torch.manual_seed(3)
    model = FNO2d(1, 1, 12, 12, 32).to(gpu_device).eval()
    x = torch.randn(4, 1, 512, 512, device=gpu_device)
    hammer = _Hammer(gpu_device)
    with torch.no_grad():
        a = model(x).clone()
        hammer()
        b = model(x).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b)
