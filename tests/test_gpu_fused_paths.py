"""The fused h2 kernels (two-piece f16 arithmetic with dynamic power-of-two scaling: csrc/h2.h,
fused_spectral.hip, ff_fused.hip) against the per-GEMM split-bf16 path they replace, on the same inputs --
including inputs whose scale is far from 1 or varies by many orders of magnitude inside one tensor, which is what
the scaling has to survive (f16 alone has a 5-bit exponent)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


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


def _spectral(x, wy, wx, K, g):
    from rpde import ops
    xs = x.clone().requires_grad_(True)
    a, b = wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    out = ops.fspectral2d(xs, a, b, K)
    out.backward(g)
    return out.detach(), xs.grad, a.grad, b.grad


@pytest.mark.parametrize("scale", [1.0, 1e-7, 3e5])
@pytest.mark.parametrize("shape", [(2, 64, 64, 20), (1, 96, 32, 12), (3, 32, 64, 4)])
def test_fused_spectral_equals_gemm_path(gpu_device, shape, scale):
    B, M, N, K = shape
    torch.manual_seed(M + N + K)
    x = torch.randn(B, M, N, 64, device=gpu_device) * scale
    # rows whose magnitude differs by 2^+-20 inside one tensor: every line / chunk carries its own scale
    x[:, ::3] *= 1e-6
    x[:, 1::7] *= 1e6
    wy = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    wx = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    g = torch.randn(B, M, N, 64, device=gpu_device) / scale
    fused = _spectral(x, wy, wx, K, g)
    with _env(RPDE_FUSED_SPECTRAL="0"):
        plain = _spectral(x, wy, wx, K, g)
    for name, a, b in zip(("out", "dx", "dWy", "dWx"), fused, plain):
        assert torch.isfinite(a).all(), name
        assert _rel(a, b) < 2e-6, (name, _rel(a, b))


def test_rectangular_grid_clamps_modes_per_axis(gpu_device):
    """[M >= 64, N = 32] with 20 modes: 17 bins survive along y, 20 along x, both pad to 20 -- the fused path's mode mix
    knows ONE clamp for both axes, so this shape must not take it (round-3 advisor finding: x-axis modes 17..19 got zero
    weights and zero gradients).  Checked against torch.fft in float64, forward and the x-axis weight gradient."""
    B, M, N, K = 2, 64, 32, 20
    torch.manual_seed(5)
    x = torch.randn(B, M, N, 64, device=gpu_device)
    wy = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    wx = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    g = torch.randn(B, M, N, 64, device=gpu_device)
    out, dx, dwy, dwx = _spectral(x, wy, wx, K, g)

    xr = x.detach().cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    wyr, wxr = wy.cpu().double().requires_grad_(True), wx.cpu().double().requires_grad_(True)
    ky, kx = min(K, N // 2 + 1), min(K, M // 2 + 1)
    fy = torch.fft.rfft(xr, dim=-1, norm="ortho")
    oy = fy.new_zeros(B, 64, M, N // 2 + 1)
    oy[..., :ky] = torch.einsum("bixy,ioy->boxy", fy[..., :ky], torch.view_as_complex(wyr[:, :, :ky]))
    fx = torch.fft.rfft(xr, dim=-2, norm="ortho")
    ox = fx.new_zeros(B, 64, M // 2 + 1, N)
    ox[:, :, :kx] = torch.einsum("bixy,iox->boxy", fx[:, :, :kx], torch.view_as_complex(wxr[:, :, :kx]))
    ref = (torch.fft.irfft(oy, n=N, dim=-1, norm="ortho") + torch.fft.irfft(ox, n=M, dim=-2, norm="ortho")).permute(0, 2, 3, 1)
    ref.backward(g.cpu().double())
    assert _rel(out.cpu(), ref.detach()) < 2e-6
    assert _rel(dwx.cpu(), wxr.grad) < 5e-6 and _rel(dwy.cpu(), wyr.grad) < 5e-6
    assert float(dwx[:, :, 17:].abs().max()) > 0           # the x axis keeps all 20 modes
    assert float(dwy[:, :, 17:].abs().max()) == 0          # the y axis (n = 32) only 17


def test_fused_spectral_lowpass_and_skip_gradient(gpu_device):
    from rpde import ops
    torch.manual_seed(3)
    x = torch.randn(2, 64, 64, 64, device=gpu_device, requires_grad=True)
    res = {}
    for tag, env in (("fused", {}), ("plain", {"RPDE_FUSED_SPECTRAL": "0"})):
        with _env(**env):
            xs = x.detach().clone().requires_grad_(True)
            out, skip = ops.fspectral2d(xs, None, None, 12, mode="low-pass", with_skip=True)
            (out * 2 + skip).sum().backward()           # the skip gradient is added by the last backward kernel
            res[tag] = (out.detach(), xs.grad)
    assert _rel(res["fused"][0], res["plain"][0]) < 2e-6 and _rel(res["fused"][1], res["plain"][1]) < 2e-6


def _ff(ff, x, res, g, train):
    ff.train(train)
    xs = x.clone().requires_grad_(True)
    torch.manual_seed(11)                       # same dropout seed draw in both runs
    out = ff(xs, residual=res)
    out.backward(g)
    grads = [p.grad.clone() for p in ff.parameters()]
    for p in ff.parameters():
        p.grad = None
    return [out.detach(), xs.grad] + grads


@pytest.mark.parametrize("scale", [1.0, 1e-6, 1e4])
@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_fused_feedforward_equals_gemm_path(gpu_device, scale, dropout):
    from models.custom_layer import FeedForward
    torch.manual_seed(5)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=dropout).to(gpu_device)
    with torch.no_grad():
        for p in ff.parameters():
            p.mul_(1.0 + 0.3 * torch.randn_like(p))
    P = 17 * 37 * 41                             # 25789 points = 806 tiles: several per persistent workgroup, last one partial
    x = torch.randn(P, 64, device=gpu_device) * scale
    x[::5] *= 1e-5
    res = torch.randn(P, 64, device=gpu_device)
    g = torch.randn(P, 64, device=gpu_device)
    with _env(RPDE_FF_STASH="u"):                         # recompute mode: only u = dropout(z) is saved
        fused = _ff(ff, x, res, g, True)
    fused_hd = _ff(ff, x, res, g, True)                   # default: the forward kernel stores h and d
    with _env(RPDE_FUSED_FF="0", RPDE_WGRAD_H2="0"):      # plain leg: per-layer GEMMs only
        plain = _ff(ff, x, res, g, True)
    names = ["out", "dx"] + [n for n, _ in ff.named_parameters()]
    for name, a, a2, b in zip(names, fused, fused_hd, plain):
        assert torch.isfinite(a).all() and torch.isfinite(a2).all(), name
        # (weight gradients are sums over 25789 points whose inputs span nine orders of magnitude: two fp32-class
        #  evaluations of such a sum agree to a few 1e-6 .. 1e-5)
        tol = 2e-6 if name == "out" else 3e-5
        assert _rel(a, b) < tol and _rel(a2, b) < tol, (name, _rel(a, b), _rel(a2, b))
    assert torch.equal(fused[0], fused_hd[0])             # the forward arithmetic does not depend on what is saved
    # evaluation: nothing but the output is written, same numbers as the training-mode kernel without dropout
    ff.eval()
    with torch.no_grad():
        e_f = ff(x, residual=res)
        with _env(RPDE_FUSED_FF="0"):
            e_p = ff(x, residual=res)
    assert _rel(e_f, e_p) < 2e-6


def test_fused_feedforward_is_reproducible_and_tail_safe(gpu_device):
    from models.custom_layer import FeedForward
    torch.manual_seed(1)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=0.0).to(gpu_device).train()
    x = torch.randn(1000 + 17, 64, device=gpu_device)
    outs = []
    for _ in range(2):
        xs = x.clone().requires_grad_(True)
        o = ff(xs)
        o.square().sum().backward()
        outs.append((o.detach().clone(), xs.grad.clone(), ff.layers[1][0].weight.grad.clone(), ff.layers[1][0].bias.grad.clone()))
        for p in ff.parameters():
            p.grad = None
    for a, b in zip(*outs):
        assert torch.equal(a, b)                 # no atomics anywhere: bitwise identical run to run


@pytest.mark.parametrize("scale", [1.0, 1e-6, 1e5])
def test_channels_first_h2_dft_equals_gemm_path(gpu_device, scale):
    """cf_dft.hip (SpectralConv1d/2d stage along the contiguous axis, spectral resize) against the GEMM path"""
    from rpde import ops
    torch.manual_seed(9)
    x2 = torch.randn(2, 8, 64, 256, device=gpu_device) * scale
    x2[:, ::3] *= 1e-4
    w1 = (torch.rand(8, 8, 6, 12, dtype=torch.cfloat) / 64).to(gpu_device)
    w2 = (torch.rand(8, 8, 6, 12, dtype=torch.cfloat) / 64).to(gpu_device)
    x1 = torch.randn(3, 8, 1024, device=gpu_device) * scale
    w = (torch.rand(8, 8, 16, dtype=torch.cfloat) / 64).to(gpu_device)
    g2 = torch.randn(2, 8, 64, 256, device=gpu_device) / scale
    g1 = torch.randn(3, 8, 1024, device=gpu_device) / scale

    def run():
        a = x2.clone().requires_grad_(True)
        p, q = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        o2 = ops.spectral2d(a, p, q)
        o2.backward(g2)
        b = x1.clone().requires_grad_(True)
        r = w.clone().requires_grad_(True)
        o1 = ops.spectral1d(b, r)
        o1.backward(g1)
        with torch.no_grad():
            rz = ops.resize2d(x2, (128, 128))
            rz1 = ops.resize1d(x1, 384)
        return [o2.detach(), a.grad, torch.view_as_real(p.grad), torch.view_as_real(q.grad), o1.detach(), b.grad,
                torch.view_as_real(r.grad), rz, rz1]

    fused = run()
    with _env(RPDE_FUSED_CF="0"):
        plain = run()
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.isfinite(a).all(), i
        assert _rel(a, b) < 3e-6, (i, _rel(a, b))


def test_fno2d_evaluation_path_equals_training_path_forward(gpu_device):
    """without autograd every FNO block stores its activated output once (conv1x1 with a fused output activation);
    with autograd the blocks pass pre-activations: same numbers"""
    from models.fno import FNO1d, FNO2d
    torch.manual_seed(2)
    m2 = FNO2d(1, 1, modes1=6, modes2=6, width=16).to(gpu_device)
    x = torch.randn(2, 1, 128, 128, device=gpu_device)
    with torch.no_grad():
        e = m2(x)
        be = m2.fno_blocks[0](torch.randn(2, 16, 128, 128, device=gpu_device))
    t = m2(x.requires_grad_(True))
    assert _rel(e, t.detach()) < 2e-6 and torch.isfinite(be).all()
    m1 = FNO1d(1, 1, modes=8, width=16).to(gpu_device)
    x1 = torch.randn(2, 1, 256, device=gpu_device)
    with torch.no_grad():
        e1 = m1(x1)
    assert _rel(e1, m1(x1.requires_grad_(True)).detach()) < 2e-6


@pytest.mark.parametrize("shape", [(2, 128, 128, 6, 6), (1, 512, 512, 12, 12), (3, 40, 256, 5, 9), (2, 77, 64, 3, 4)])
@pytest.mark.parametrize("user_grid", [False, True])
def test_fno2d_lifted_first_block_equals_the_unfused_path(gpu_device, shape, user_grid):
    """rpde_fno2d_lift_block_eval_fwd (grid concat + lifting + first FNO block without the lifted field: row spectra by
    linearity in k_col_analysis<.., LIFT>, the slab formed on the fly in k_conv_syn_h2<.., LIFT>) against the model with
    RPDE_LIFT_FUSED=0 (concat kernel, lifting convolution, fnoblock2d_eval) and against the training-mode forward"""
    import numpy as np
    from models.fno import FNO2d
    from rpde import _lib
    B, M, N, m1, m2 = shape
    torch.manual_seed(M + N)
    grid = None
    if user_grid:
        grid = (np.sort(np.random.RandomState(1).rand(M)).astype(np.float32), np.linspace(-2.0, 3.0, N).astype(np.float32))
    model = FNO2d(1, 1, modes1=m1, modes2=m2, width=32, grid=grid).to(gpu_device).eval()
    x = torch.randn(B, 1, M, N, device=gpu_device) * torch.logspace(-2, 2, M, device=gpu_device).view(1, 1, M, 1)
    assert _lib.load().rpde_fno2d_lift_block_eval_ok(1, 32, 32, M, N, m1, m2) == 1
    with torch.no_grad():
        fused = model(x)
    old = os.environ.get("RPDE_LIFT_FUSED")
    os.environ["RPDE_LIFT_FUSED"] = "0"
    try:
        with torch.no_grad():
            plain = model(x)
    finally:
        if old is None:
            os.environ.pop("RPDE_LIFT_FUSED")
        else:
            os.environ["RPDE_LIFT_FUSED"] = old
    train = model(x.clone().requires_grad_(True)).detach()
    assert torch.isfinite(fused).all()
    # per row (the rows differ by four orders of magnitude)
    num = (fused - plain).flatten(0, 1).norm(dim=-1)
    den = plain.flatten(0, 1).norm(dim=-1).clamp_min(1e-30)
    assert float((num / den).max()) < 1e-5, float((num / den).max())
    assert _rel(fused, plain) < 2e-6, _rel(fused, plain)
    assert _rel(fused, train) < 2e-6, _rel(fused, train)


@pytest.mark.parametrize("in_f,out_f", [(256, 256), (64, 256), (256, 64)])
@pytest.mark.parametrize("P", [32 * 640, 32 * 811 + 7])
def test_streaming_weight_gradient_kernel(gpu_device, in_f, out_f, P):
    """csrc/wgrad_h2.hip (running-exponent f16 pieces, transposing LDS reads) against float64 and against the generic
    split-bf16 GEMM it replaces for the headline FeedForward shapes; rows whose magnitudes differ by 1e9 and a step
    in which the running maximum jumps exercise the accumulator rescaling"""
    from rpde import ops
    torch.manual_seed(in_f + out_f + P)
    x = torch.randn(P, in_f, device=gpu_device)
    g = torch.randn(P, out_f, device=gpu_device)
    x[::7] *= 1e-4
    g[::5] *= 1e-5
    x[P // 2: P // 2 + 3] *= 3e4                       # late, isolated huge rows: the running exponents move mid-stream
    g[3 * P // 4] *= 1e5
    w = torch.randn(out_f, in_f, device=gpu_device) / in_f ** 0.5
    b = torch.zeros(out_f, device=gpu_device)

    def grads():
        ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ops.linear(x, ws, bs).backward(g)
        return ws.grad, bs.grad
    gw, gb = grads()
    gw2, _ = grads()
    assert torch.equal(gw, gw2)                                                # fixed-order slabs: reproducible
    with _env(RPDE_WGRAD_H2="0"):
        gw_gemm, _ = grads()
    ref = g.double().t() @ x.double()
    assert _rel(gw, ref) < 2e-6, _rel(gw, ref)
    assert _rel(gw_gemm, ref) < 2e-6
    assert not torch.equal(gw, gw_gemm)                                        # really two different kernels
    assert _rel(gb, g.double().sum(0)) < 2e-6


@pytest.mark.parametrize("in_f,out_f", [(3, 64), (1, 64), (2, 32), (4, 256), (64, 1), (128, 1), (64, 2), (32, 3), (256, 4)])
@pytest.mark.parametrize("P", [70001, 37])
def test_thin_linear_kernels(gpu_device, in_f, out_f, P):
    """csrc/thin_linear.hip (lifting / projection layers as streaming kernels) against float64 and against the GEMM
    path (RPDE_THIN_LINEAR=0): forward, data gradient, weight and bias gradients; reproducible bit for bit"""
    from rpde import ops
    torch.manual_seed(in_f * 1000 + out_f + P)
    x = torch.randn(P, in_f, device=gpu_device)
    g = torch.randn(P, out_f, device=gpu_device)
    w = torch.randn(out_f, in_f, device=gpu_device) / in_f ** 0.5
    b = torch.randn(out_f, device=gpu_device)

    def run():
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.linear(xs, ws, bs)
        y.backward(g)
        return y.detach(), xs.grad, ws.grad, bs.grad
    got, again = run(), run()
    for a_, b_ in zip(got, again):
        assert torch.equal(a_, b_)
    with _env(RPDE_THIN_LINEAR="0"):
        gemm = run()
    xd, wd, bd, gd = x.double(), w.double(), b.double(), g.double()
    ref = (xd @ wd.t() + bd, gd @ wd, gd.t() @ xd, gd.sum(0))
    for name, a_, m_, r_ in zip(("y", "dx", "dw", "db"), got, gemm, ref):
        assert a_.shape == r_.shape
        tol = 2e-6 if name in ("y", "dx") else 1e-5      # dw, db: fp32 sums of up to 70001 signed terms (cancellation)
        assert _rel(a_, r_) < tol, (name, _rel(a_, r_))
        assert _rel(m_, r_) < tol, (name, "gemm", _rel(m_, r_))


@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 32), (128, 1), (64, 2), (5, 20), (17, 7)])
@pytest.mark.parametrize("act_in,act_out", [("identity", "identity"), ("gelu", "gelu"), ("relu", "identity")])
def test_small_channel_conv_kernel(gpu_device, cin, cout, act_in, act_out):
    """csrc/conv_small.hip (streaming 1x1 conv, Cout <= 32, channels-first) against the GEMM path
    (RPDE_CONV_SMALL=0) and float64: plain, accumulating into the spectral branch's output, and the evaluation
    form with the output activation in the epilogue"""
    import torch.nn.functional as F
    from rpde import ops
    torch.manual_seed(cin * 100 + cout)
    x = torch.randn(5, cin, 512, 412, device=gpu_device)              # 1.05 M points: above the kernel's size threshold
    w = torch.randn(cout, cin, 1, 1, device=gpu_device) / cin ** 0.5
    b = torch.randn(cout, device=gpu_device)
    acc0 = torch.randn(5, cout, 512, 412, device=gpu_device)
    act = {"identity": lambda t: t, "gelu": F.gelu, "relu": F.relu}

    def run():
        with torch.no_grad():
            plain = ops.conv1x1(x, w, b, act_in)
            accum = ops.conv1x1(x, w, b, act_in, acc=acc0.clone(), acc_owned=True)
            ev = ops.conv1x1_act_eval(x, w, b, acc0.clone(), act_out)
        return plain, accum, ev
    got = run()
    with _env(RPDE_CONV_SMALL="0"):
        gemm = run()
    lin = torch.einsum("oi,bihw->bohw", w[:, :, 0, 0].double(), act[act_in](x.double())) + b.double()[None, :, None, None]
    lin0 = torch.einsum("oi,bihw->bohw", w[:, :, 0, 0].double(), x.double()) + b.double()[None, :, None, None]
    refs = (lin, acc0.double() + lin, act[act_out](acc0.double() + lin0))
    for name, a_, m_, r_ in zip(("plain", "accumulate", "eval"), got, gemm, refs):
        assert _rel(a_, r_) < 2e-6, (name, _rel(a_, r_))
        assert _rel(m_, r_) < 2e-6, (name, "gemm", _rel(m_, r_))


@pytest.mark.parametrize("shape", [(9, 64, 64, 20), (8, 128, 128, 12), (5, 64, 64, 16), (2, 64, 64, 24), (3, 128, 64, 8),
                                   (1, 192, 192, 20), (17, 64, 64, 4), (2, 256, 256, 20), (3, 96, 96, 12), (2, 160, 160, 20),
                                   (40, 32, 32, 8), (70, 64, 64, 20)])
def test_round3_spectral_kernels_over_batch_grid_and_mode_variants(gpu_device, shape):
    """the round-3 kernels (k_dft_analysis_sq_h2: XCD groups with uneven sample counts, grids of 32 .. 256 incl. those whose
    rows do not fill a workgroup's eight waves, more groups than samples and fewer; k_mix_h2 /
    k_mix_wgrad_h2: every (K32, TG) operand layout incl. the packed tails; k_dft_synthesis3_h2: persistent tiles with
    1 .. many tiles per workgroup, with and without the skip gradient) against the per-GEMM path"""
    B, M, N, K = shape
    torch.manual_seed(B * 1000 + M + N + K)
    x = torch.randn(B, M, N, 64, device=gpu_device)
    x = x * torch.logspace(-3, 3, B, device=gpu_device).view(B, 1, 1, 1)          # samples of very different magnitude
    wy = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    wx = torch.randn(64, 64, K, 2, device=gpu_device) * 0.1
    g = torch.randn(B, M, N, 64, device=gpu_device)

    def run():
        from rpde import ops
        xs = x.clone().requires_grad_(True)
        a, b = wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
        out, skip = ops.fspectral2d(xs, a, b, K, with_skip=True)
        (out * g + skip * (0.5 * g)).sum().backward()          # the skip gradient is added by the adjoint synthesis
        return out.detach(), xs.grad, a.grad, b.grad

    with _env(RPDE_FUSED_SPECTRAL="0"):
        plain = run()
    # default dispatch (one-pass analysis on every square grid), the two-read analysis kernel forced (RPDE_ANA_SQ=0)
    for leg in ({}, {"RPDE_ANA_SQ": "0"}):
        with _env(**leg):
            fused = run()
        for name, p, q in zip(("out", "dx", "dWy", "dWx"), fused, plain):
            assert torch.isfinite(p).all(), (leg, name)
            # per sample for the fields (a sample 1e6 smaller than its neighbour must keep its own accuracy)
            if name in ("out", "dx"):
                worst = max(_rel(p[i], q[i]) for i in range(B))
                assert worst < 3e-6, (leg, name, worst)
            else:
                assert _rel(p, q) < 3e-6, (leg, name, _rel(p, q))
