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
