"""Evaluation with frozen weights (rpde.ops.frozen_weights -> rpde_feedforward_prepare / _fwd_prepared,
rpde_fspectral2d_prepare / _fwd_prepared): the prepared path must give the very bits of the ordinary evaluation path
(same kernels, same weight fragments -- only built once), must notice a torch in-place weight update, and must not
outlive its scope.  Reference behaviour being mirrored: evaluation loops of train/training.py:78-146 and the rollout of
utils/autoregressive_step.py:284-309, where the model is called repeatedly between two optimiser steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG3 = dict(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True,
            n_ff_layers=3, layer_norm=True, dropout=0.0)


def _model(gpu_device, **over):
    from models.ffno import FFNO2D
    torch.manual_seed(3)
    return FFNO2D(**{**CFG3, **over}).to(gpu_device).eval()


@pytest.mark.parametrize("res,batch", [(64, 3), (128, 2), (256, 1)])
def test_frozen_scope_is_bit_identical_to_plain_evaluation(gpu_device, res, batch):
    from rpde import ops
    from utils.synthetic import random_fields
    model = _model(gpu_device)
    x = random_fields(batch, res, 2, seed=5).to(gpu_device)
    with torch.no_grad():
        plain = model(x)
        with ops.frozen_weights():
            first = model(x)
            n_entries = len(ops._FROZEN)
            second = model(x)                                  # served from the prepared buffers
            assert len(ops._FROZEN) == n_entries
    assert ops._FROZEN is None
    # 4 layers x (one spectral pair + backcast FeedForward) on fused shapes
    assert n_entries >= 4, n_entries
    assert torch.equal(plain, first) and torch.equal(plain, second)


def test_frozen_scope_sees_a_torch_inplace_update_and_nests(gpu_device):
    from rpde import ops
    from utils.synthetic import random_fields
    model = _model(gpu_device)
    x = random_fields(2, 64, 2, seed=6).to(gpu_device)
    with torch.no_grad(), ops.frozen_weights():
        a = model(x)
        with ops.frozen_weights():                             # nested: the outer scope's entries stay
            assert torch.equal(model(x), a)
        assert ops._FROZEN is not None
        for p in model.parameters():
            p.mul_(1.25)                                       # bumps _version: every entry is rebuilt
        b = model(x)
    with torch.no_grad():
        ref = model(x)
    assert not torch.equal(a, b)
    assert torch.equal(b, ref)


def test_prepared_feedforward_direct_call_matches(gpu_device):
    """C ABI: rpde_feedforward_prepare + rpde_feedforward_fwd_prepared against rpde_feedforward_fwd, with a partial last
    tile, residual and post-activation"""
    from models.custom_layer import FeedForward
    from rpde import ops
    P = 4096 + 37
    torch.manual_seed(1)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=0.0).to(gpu_device).eval()
    x = torch.randn(P, 64, device=gpu_device)
    res = torch.randn(P, 64, device=gpu_device)
    with torch.no_grad():
        plain = ff(x, residual=res, post_act="gelu")
        with ops.frozen_weights():
            prepared = ff(x, residual=res, post_act="gelu")
            again = ff(x * 2, residual=res, post_act="gelu")
        ref2 = ff(x * 2, residual=res, post_act="gelu")
    assert torch.equal(plain, prepared)
    assert torch.equal(again, ref2)


def test_rollout_and_sweep_use_the_scope(gpu_device, monkeypatch):
    from rpde import ops
    from utils.autoregressive_step import perform_rollout_2d
    from utils.synthetic import random_fields
    model = _model(gpu_device)
    seen = []
    real = ops._frozen_entry

    def spy(kind, tensors, extra, nbytes, build, originals=None):
        built = []
        r = real(kind, tensors, extra, nbytes, lambda buf: (built.append(1), build(buf)), originals=originals)
        seen.append((kind, bool(built)))
        return r

    monkeypatch.setattr(ops, "_frozen_entry", spy)
    x0 = random_fields(2, 64, 2, seed=8).to(gpu_device)[:, 0]
    steps = 5
    preds = perform_rollout_2d(model, x0, steps, device=gpu_device)
    assert preds.shape == (2, steps, 64, 64)
    builds = sum(1 for _, b in seen if b)
    assert builds * steps == len(seen), (builds, len(seen))   # every layer prepared once, used `steps` times
    # the same rollout without the scope
    monkeypatch.setattr(ops, "_frozen_entry", lambda *a, **k: None)
    assert torch.equal(perform_rollout_2d(model, x0, steps, device=gpu_device), preds)


def test_eight_layer_model_runs_every_layer_on_the_fused_kernels(gpu_device):
    """`n_layers: 8` (the reference yaml's other setting, conf/model/ffno_2d/ffno_2d.yaml:7) changes how often the layer
    kernels run, not which ones: every layer of an 8-layer FFNO2D prepares one fused spectral pair and one fused
    FeedForward (a shape on the per-GEMM path prepares nothing)."""
    from rpde import ops
    from utils.synthetic import random_fields
    model = _model(gpu_device, n_layers=8)
    x = random_fields(2, 64, 2, seed=9).to(gpu_device)
    lib = ops.load()
    assert lib.rpde_feedforward_is_fused(64, 4, 3, 2 * 64 * 64) == 1
    with torch.no_grad(), ops.frozen_weights():
        y = model(x)
        kinds = sorted(k[0] if isinstance(k, tuple) else k for k in ops._FROZEN)
    assert torch.isfinite(y).all()
    assert len(kinds) >= 16, kinds
