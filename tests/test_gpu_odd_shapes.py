"""GPU: shapes that leave the 16-byte fast path (widths not divisible by 4, odd grids, tiny batches) run
through the scalar-load kernels and the generic epilogue; checked against the CPU oracle directly."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _compare(model, oracle_fwd, x, tol_f=1e-5, tol_g=1e-4):
    from oracle import reference_path as R
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    y = torch.randn(x.shape[0], model.out_channels if hasattr(model, "out_channels") else 1, *x.shape[2:])
    from utils.loss import RelativeL2Loss
    xg = x.clone().to("cuda:0").requires_grad_(True)
    pred = model(xg)
    loss = RelativeL2Loss()(pred, y.to("cuda:0"))
    loss.backward()
    p = R.make_params(sd)
    xr = x.clone().requires_grad_(True)
    ref = oracle_fwd(p, xr)
    R.relative_l2(ref, y).backward()
    ef = float((pred.detach().cpu() - ref.detach()).norm() / ref.detach().norm())
    assert ef < tol_f, ef
    assert float((xg.grad.cpu() - xr.grad).norm() / xr.grad.norm()) < tol_g
    for k, v in model.named_parameters():
        g, r = v.grad.cpu(), p[k].grad
        if r.norm() > 1e-6 * max(1.0, float(max(q.grad.norm() for q in p.values()))):
            assert float((g - r).norm() / r.norm()) < tol_g, k


def test_ffno2d_odd_width_and_grid(gpu_device):
    from models.ffno import FFNO2D
    from oracle import reference_path as R
    torch.manual_seed(0)
    cfg = dict(width=10, n_layers=2, n_modes=3, factor=3, ff_weight_norm=True, n_ff_layers=2, layer_norm=True)
    m = FFNO2D(3, 2, **cfg).to(gpu_device).train()
    x = torch.randn(1, 3, 10, 14)
    _compare(m, lambda p, xx: R.ffno2d_forward(p, xx, 2, 3, 2, True), x)


def test_ffno1d_odd_everything(gpu_device):
    from models.ffno import FFNO1D
    from oracle import reference_path as R
    torch.manual_seed(1)
    m = FFNO1D(2, 1, width=6, n_layers=2, n_modes=9, factor=2, ff_weight_norm=False, n_ff_layers=3, layer_norm=False,
               activation="relu").to(gpu_device).train()
    x = torch.randn(3, 2, 21)
    _compare(m, lambda p, xx: R.ffno1d_forward(p, xx, 2, 9, 3, False, activation="relu"), x, tol_g=3e-4)


def test_fno2d_and_fno1d_odd(gpu_device):
    from models.fno import FNO1d, FNO2d
    from oracle import reference_path as R
    torch.manual_seed(2)
    m = FNO2d(1, 1, 3, 2, 6, n_blocks=2).to(gpu_device).train()
    _compare(m, lambda p, xx: R.fno2d_forward(p, xx, 2), torch.randn(2, 1, 9, 10))
    import torch.nn.functional as F
    m1 = FNO1d(3, 1, 5, 7, activation=F.gelu, n_blocks=2).to(gpu_device).train()
    _compare(m1, lambda p, xx: R.fno1d_forward(p, xx, 2, "gelu"), torch.randn(2, 3, 19))


@pytest.mark.parametrize("dim,factor,n_layers,P", [(32, 2, 1, 700), (48, 4, 4, 2500), (32, 2, 5, 1500), (16, 3, 6, 1200),
                                                   (128, 4, 3, 8192)])
def test_feedforward_depths_around_the_one_launch_limits(gpu_device, dim, factor, n_layers, P):
    """The GEMM-path FeedForward splits the weight images of up to four layers in one launch and folds up to twelve slab
    sets in one launch (csrc/feedforward.hip: SplitJobs, FoldJobs); deeper stacks fall back to per-layer launches for
    what does not fit.  Forward and every gradient against the float64 restatement of models/custom_layer.py:49-68 for
    1 .. 6 layers (and the 128-wide yaml shape of FFNO1D)."""
    from models.custom_layer import FeedForward
    from oracle import reference_path as R
    torch.manual_seed(n_layers * 100 + dim)
    ff = FeedForward(dim, factor, n_layers=n_layers, layer_norm=True, dropout=0.0).to(gpu_device).train()
    x = torch.randn(P, dim, device=gpu_device, requires_grad=True)
    probe = torch.randn(P, dim, device=gpu_device)
    out = ff(x)
    (out * probe).sum().backward()
    sd = {k: v.detach().double().cpu().requires_grad_() for k, v in ff.state_dict().items()}
    xd = x.detach().double().cpu().requires_grad_()
    ref = R.feedforward(xd, sd, "", n_layers, True)
    (ref * probe.double().cpu()).sum().backward()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))

    assert rel(out.detach(), ref.detach()) < 5e-6
    assert rel(x.grad, xd.grad) < 2e-5
    for k, p in ff.named_parameters():
        assert rel(p.grad, sd[k].grad) < 2e-5, k
