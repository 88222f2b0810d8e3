"""GPU: the training-side behaviour that golden vectors cannot pin -- dropout
(statistical parity only, SURVEY hard part 3), mixed-resolution batches, and
the Hydra-shaped entry points end to end."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dropout_statistics_and_eval_mode(gpu_device):
    from models.custom_layer import FeedForward
    torch.manual_seed(0)
    ff = FeedForward(64, 4, n_layers=3, layer_norm=False, dropout=0.25).to(gpu_device)
    x = torch.randn(4096, 64, device=gpu_device)
    ff.eval()
    e1, e2 = ff(x), ff(x)
    assert torch.equal(e1, e2)
    ff.train()
    t1, t2 = ff(x), ff(x)
    assert not torch.equal(t1, t2)                       # fresh mask per call
    # last layer: out = dropout(z3) -> a fraction p of the outputs is exactly zero
    frac = (t1 == 0).float().mean().item()
    assert abs(frac - 0.25) < 0.01, frac
    # inverted dropout keeps the expectation: the mean over many masks approaches eval output
    acc = torch.zeros_like(e1)
    n = 64
    for _ in range(n):
        acc += ff(x)
    rel = float((acc / n - e1).norm() / e1.norm())
    assert rel < 0.35, rel                               # E[gelu(drop z)] != gelu(z) exactly; loose statistical bound


def test_dropout_backward_regenerates_the_forward_mask(gpu_device):
    """directional finite difference of the *stochastic* layer with a pinned seed"""
    from models.custom_layer import FeedForward
    torch.manual_seed(1)
    ff = FeedForward(32, 2, n_layers=3, layer_norm=True, dropout=0.3).to(gpu_device).train()
    x = torch.randn(512, 32, device=gpu_device, dtype=torch.float32)
    cot = torch.randn(512, 32, device=gpu_device)
    v = torch.randn_like(x)
    v /= v.norm()

    def loss_at(xx):
        torch.manual_seed(1234)                          # the layer draws its mask seed from the CPU generator
        return (ff(xx) * cot).sum()

    xr = x.clone().requires_grad_(True)
    loss_at(xr).backward()
    analytic = float((xr.grad * v).sum())
    eps = 2e-2
    numeric = float((loss_at(x + eps * v).double() - loss_at(x - eps * v).double()) / (2 * eps))
    assert abs(analytic - numeric) <= 3e-2 * max(1.0, abs(numeric)), (analytic, numeric)
    # parameter gradients too: perturb the first weight along a random direction
    w = ff.layers[0][0].weight
    dw = torch.randn_like(w)
    dw /= dw.norm()
    ff.zero_grad()
    loss_at(x).backward()
    analytic_w = float((w.grad * dw).sum())
    with torch.no_grad():
        w.add_(eps * dw)
        lp = loss_at(x).double()
        w.sub_(2 * eps * dw)
        lm = loss_at(x).double()
        w.add_(eps * dw)
    numeric_w = float((lp - lm) / (2 * eps))
    assert abs(analytic_w - numeric_w) <= 3e-2 * max(1.0, abs(numeric_w)), (analytic_w, numeric_w)


def test_mixed_resolution_batches_share_weights(gpu_device):
    """the same FFNO2D weights run 32^2 (clamped modes), 48x64 and 64^2 batches back to back"""
    from models.ffno import FFNO2D
    from utils.loss import RelativeL2Loss
    torch.manual_seed(0)
    m = FFNO2D(1, 1, width=16, n_layers=2, n_modes=20, factor=2, ff_weight_norm=True, n_ff_layers=2,
               layer_norm=True, dropout=0.1).to(gpu_device).train()
    loss_fn = RelativeL2Loss()
    for shape in [(2, 1, 32, 32), (3, 1, 48, 64), (1, 1, 64, 64), (2, 1, 32, 32)]:
        x = torch.randn(*shape, device=gpu_device)
        loss = loss_fn(m(x), torch.randn(*shape, device=gpu_device))
        loss.backward()
        assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_main_2d_trains_and_checkpoints(gpu_device, tmp_path, capsys):
    from rpde.entry import run
    l2 = run(2, ["model=ffno_2d/ffno_2d", "dataset=synthetic/ns_mres", "dataset.resolutions={32: 24, 48: 24}",
                 "dataset.n_val=8", "dataset.n_test=8", "model.width=16", "model.n_layers=2", "model.n_modes=8",
                 "model.factor=2", "training.epochs=6", "training.batch_size=8", "training.learning_rate=0.003",
                 f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    first = [ln for ln in out.splitlines() if '"epoch": 0' in ln]
    assert first and l2 < 1.0
    import json
    e0 = json.loads(first[0])
    assert l2 < e0["val_loss"], (l2, e0)                 # learning happened
    ck = torch.load(tmp_path / "rpde_mi355x_2d.pt", weights_only=True)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "loss_history", "val_loss_history", "l2_loss",
                       "resolution_rel_l2"}
    # the reference's post-training sequence (main_2d.py:287): one record per resolution up to the test grid
    rec = [json.loads(ln) for ln in out.splitlines() if '"resolution_rel_l2"' in ln]
    assert rec and rec[0]["evaluation_type"] == "naive_downsample"
    assert list(rec[0]["resolution_rel_l2"]) == ["48"] and 0 < rec[0]["resolution_rel_l2"]["48"] < 1.0
    assert abs(rec[0]["resolution_rel_l2"]["48"] - l2) < 0.05 * l2 + 1e-3      # same samples at the native resolution


def test_main_1d_fno_and_ffno(gpu_device, tmp_path):
    from rpde.entry import run
    for model in ("ffno_1d/ffno_1d", "fno_1d/fno_1d"):
        extra = ["model.width=16", "model.n_layers=2", "model.n_modes=8", "model.factor=2"] if "ffno" in model else \
                ["model.width=16", "model.modes=8"]
        l2 = run(1, [f"model={model}", "dataset=synthetic/ks_512", "dataset.resolutions={64: 32}", "dataset.n_val=8",
                     "dataset.n_test=8", "training.epochs=3", "training.batch_size=8", f"checkpoint_dir={tmp_path}"]
                 + extra)
        assert l2 == l2 and l2 < 2.0
        # post-training: resolutions [32, 64] and the autoregressive rollout at each (main_1d.py:250,272)
        last = run.last
        assert sorted(last["resolution_rel_l2"]) == [32, 64] and all(v == v and v > 0 for v in last["resolution_rel_l2"].values())
        assert sorted(last["rollout_rel_l2"]) == [32, 64] and all(v == v and v > 0 for v in last["rollout_rel_l2"].values())


def test_all_resolution_evaluator_and_rollouts(gpu_device):
    """f2/f3 rows: evaluator core over [32,64,128] in both evaluation modes; 1-D and 2-D rollouts"""
    from models.ffno import FFNO1D, FFNO2D
    from utils.autoregressive_step import perform_rollout_1d, perform_rollout_2d, rollout_loss
    from utils.resize_utils import evaluate_all_resolutions, get_lower_resolutions
    from utils.synthetic import advance, random_fields
    assert get_lower_resolutions(256) == [32, 64, 128, 256] and get_lower_resolutions(128, 64) == [64, 128]
    torch.manual_seed(0)
    m2 = FFNO2D(1, 1, width=16, n_layers=2, n_modes=12, factor=2, ff_weight_norm=True, n_ff_layers=2,
                layer_norm=True).to(gpu_device)
    x = random_fields(6, 128, 2, seed=3)
    y = advance(x, 2)
    for how in ("naive_downsample", "resize"):
        res = evaluate_all_resolutions(m2, x, y, how=how, batch_size=4, device=gpu_device)
        assert sorted(res) == [32, 64, 128] and all(v == v and v > 0 for v in res.values())
    traj = torch.stack([x[:, 0]] + [advance(x, 2)[:, 0]] * 3, dim=1).to(gpu_device)      # [B, 4, M, N]
    pred = perform_rollout_2d(m2, traj[:, 0], 3, device=gpu_device)
    assert pred.shape == (6, 3, 128, 128) and rollout_loss(pred, traj) > 0
    # rolling out by hand gives the same thing
    with torch.no_grad():
        s = traj[:, 0]
        for t in range(3):
            s = m2(s.unsqueeze(1)).squeeze(1)
            assert torch.allclose(s, pred[:, t], atol=1e-6)
    m1 = FFNO1D(1, 1, width=16, n_layers=2, n_modes=8, factor=2, ff_weight_norm=True, n_ff_layers=2,
                layer_norm=True).to(gpu_device)
    p1 = perform_rollout_1d(m1, torch.randn(5, 64, device=gpu_device), 4, device=gpu_device)
    assert p1.shape == (5, 4, 64) and torch.isfinite(p1).all()


@pytest.mark.parametrize("dims", [1, 2])
def test_skip_gradient_folded_into_spectral_backward(gpu_device, dims):
    """ops.fspectral{1,2}d(..., with_skip=True) hands x back as an alias for the skip connection; the gradient
    arriving through it is added by the last backward GEMM (rpde_gemm_desc.acc_src).  Same numbers as the
    plain formulation x + f(spectral(x)) whose two gradients autograd adds in a separate pass."""
    from rpde import ops
    torch.manual_seed(5)
    C, K = 16, 6
    shape = (3, 40, C) if dims == 1 else (2, 24, 40, C)
    x0 = torch.randn(*shape, device=gpu_device)
    ws = [(torch.randn(C, C, K, 2, device=gpu_device) * 0.2).requires_grad_() for _ in range(dims)]
    probe = torch.randn(*shape, device=gpu_device)

    def run(with_skip):
        x = x0.clone().requires_grad_()
        for w in ws:
            w.grad = None
        if dims == 1:
            out = ops.fspectral1d(x, ws[0], K, with_skip=with_skip)
        else:
            out = ops.fspectral2d(x, ws[0], ws[1], K, with_skip=with_skip)
        y, skip = out if with_skip else (out, x)
        z = torch.tanh(y) * 0.7 + skip * 1.3          # both branches used, with different weights
        (z * probe).sum().backward()
        return z.detach(), x.grad.clone(), [w.grad.clone() for w in ws]

    z0, gx0, gw0 = run(False)
    z1, gx1, gw1 = run(True)
    assert torch.equal(z0, z1)
    assert float((gx0 - gx1).norm() / gx0.norm()) < 1e-6
    for a, b in zip(gw0, gw1):
        assert torch.equal(a, b)
    # and a model-level check: the FFNO layer takes this route when the residual is its own input
    from models.spectral_convolution import FSpectralConv2d
    layer = FSpectralConv2d(C, K, factor=2, n_ff_layers=2, layer_norm=True).to(gpu_device)
    h = torch.randn(2, 16, 16, C, device=gpu_device, requires_grad=True)
    b, _ = layer(h, residual=h)
    b.sum().backward()
    g_fused = h.grad.clone()
    h.grad = None
    hh = h.detach().clone().requires_grad_()
    b2, _ = layer(hh, residual=hh.detach() * 1.0)      # a different tensor object: plain route, skip not differentiated
    (b2.sum() + hh.sum()).backward()                   # d(residual)/dh = 1 added by hand
    assert float((hh.grad - g_fused).norm() / g_fused.norm()) < 1e-6


def test_graphed_train_step_matches_eager(gpu_device):
    """rpde.graph.GraphedTrainStep: the captured hipGraph replays the same step as eager execution."""
    import copy
    from models.fno import FNO1d
    from rpde.graph import GraphedTrainStep
    from utils.loss import RelativeL2Loss
    torch.manual_seed(3)
    m_e = FNO1d(1, 1, modes=8, width=16).to(gpu_device).train()
    m_g = copy.deepcopy(m_e)
    xs = [torch.randn(4, 1, 128, device=gpu_device) for _ in range(4)]
    ys = [torch.randn(4, 1, 128, device=gpu_device) for _ in range(4)]
    loss_fn = RelativeL2Loss(size_average=True)
    o_e = torch.optim.AdamW(m_e.parameters(), lr=1e-3, capturable=True)
    o_g = torch.optim.AdamW(m_g.parameters(), lr=1e-3, capturable=True)
    warm = 2
    for _ in range(warm):                                   # the graphed step warms up on its example batch
        o_e.zero_grad(set_to_none=False)
        loss_fn(m_e(xs[0]), ys[0]).backward()
        o_e.step()
    step = GraphedTrainStep(m_g, loss_fn, o_g, xs[0], ys[0], warmup=warm)
    # capture itself executes nothing; both models are now `warm` steps in
    for x, y in zip(xs, ys):
        o_e.zero_grad(set_to_none=False)
        le = loss_fn(m_e(x), y)
        le.backward()
        o_e.step()
        lg = step(x, y)
        assert abs(float(le.detach()) - float(lg)) <= 1e-6 * max(1.0, abs(float(le.detach())))
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        a, b = torch.view_as_real(pe) if pe.is_complex() else pe, torch.view_as_real(pg) if pg.is_complex() else pg
        assert float((a - b).norm() / (a.norm() + 1e-30)) < 1e-6
    # a torch.nn.Dropout draws from the host generator: a replay would repeat one mask -> refused
    class WithTorchDropout(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.drop, self.net = torch.nn.Dropout(0.1), FNO1d(1, 1, modes=4, width=8)

        def forward(self, x):
            return self.net(self.drop(x))
    with pytest.raises(ValueError, match="Dropout"):
        md = WithTorchDropout().to(gpu_device).train()
        GraphedTrainStep(md, loss_fn, torch.optim.AdamW(md.parameters(), capturable=True),
                         torch.randn(2, 1, 64, device=gpu_device), torch.randn(2, 1, 64, device=gpu_device))


def test_train_loop_with_graph_follows_the_eager_loop_and_the_scheduler(gpu_device):
    """train(..., graph=True) (reference loop: train/training.py:19-88): GRAPH_AFTER eager steps per batch shape, then one
    hipGraph replay per step.  Same losses and weights as the eager loop over three epochs in which a StepLR halves the
    learning rate every epoch (FlatAdamW keeps lr / weight decay on the device: no new capture), with two batch shapes
    (a ragged last batch) and a validation pass between the epochs."""
    import copy
    from models.fno import FNO1d
    from rpde.optim import FlatAdamW
    from train.training import train
    torch.manual_seed(11)
    m_e = FNO1d(1, 1, modes=8, width=16).to(gpu_device)
    m_g = copy.deepcopy(m_e)
    data = [(torch.randn(1, 128), torch.randn(1, 128)) for _ in range(22)]        # batches of 4: five full + one of 2
    loader = lambda: torch.utils.data.DataLoader(data, batch_size=4, shuffle=False)   # noqa: E731
    hist = []
    for m, graph in ((m_e, False), (m_g, True)):
        opt = FlatAdamW(m.parameters(), lr=2e-3, weight_decay=1e-2, capturable=graph)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
        hist.append(train(m, loader(), loader(), opt, sched, epochs=3, device=gpu_device, graph=graph))
        assert abs(opt.param_groups[0]["lr"] - 2e-3 * 0.125) < 1e-12
    (tl_e, vl_e), (tl_g, vl_g) = hist
    for a, b in zip(tl_e + vl_e, tl_g + vl_g):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (tl_e, tl_g, vl_e, vl_g)
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        a, b = torch.view_as_real(pe) if pe.is_complex() else pe, torch.view_as_real(pg) if pg.is_complex() else pg
        assert float((a - b).norm() / (a.norm() + 1e-30)) < 2e-6
    # had the replays kept the captured learning rate, the third epoch would have stepped 4x too far: make sure this
    # test can tell (one more eager epoch at the wrong rate moves the weights measurably)
    m_w = copy.deepcopy(m_e)
    opt = FlatAdamW(m_w.parameters(), lr=2e-3, weight_decay=1e-2)
    train(m_w, loader(), loader(), opt, None, epochs=3, device=gpu_device, graph=False)
    pe, pw = next(m_e.parameters()), next(m_w.parameters())
    assert float((pe - pw).norm() / pe.norm()) > 1e-4


@pytest.mark.parametrize("dim,factor", [(64, 4), (32, 2)])          # the fused kernels / the per-GEMM path
def test_dropout_device_epoch_changes_masks_and_keeps_backward_in_step(gpu_device, dim, factor):
    """rpde_ff_params.seed_epoch: the device counter every dropout kernel mixes into its seed (what lets a captured
    hipGraph draw new masks per replay).  Same host seed: equal counter -> equal bits, other counter -> other masks with
    the same drop rate; and at a non-zero counter the backward still regenerates the forward's masks (directional
    finite difference)."""
    from models.custom_layer import FeedForward
    from rpde import ops
    torch.manual_seed(2)
    ff = FeedForward(dim, factor, n_layers=3, layer_norm=False, dropout=0.25).to(gpu_device).train()
    x = torch.randn(4096, dim, device=gpu_device)
    ep = ops.drop_epoch(gpu_device)

    def run(epoch, xx=x):
        ep.fill_(epoch)
        torch.manual_seed(77)                            # the same host-side seed draw every time
        return ff(xx)
    try:
        with torch.no_grad():
            a0, a0b, a1 = run(0), run(0), run(1)
        assert torch.equal(a0, a0b)
        assert not torch.equal(a0, a1)
        for a in (a0, a1):                               # last layer: out = dropout(z3): a fraction p is exactly zero
            assert abs(float((a == 0).float().mean()) - 0.25) < 0.02
        assert float(((a0 == 0) & (a1 == 0)).float().mean()) < 0.12        # independent masks overlap in ~p^2 = 6 %
        cot = torch.randn_like(x)
        v = torch.randn_like(x)
        v /= v.norm()
        xr = x.clone().requires_grad_(True)
        (run(7, xr) * cot).sum().backward()
        analytic = float((xr.grad * v).sum())
        eps = 2e-2
        with torch.no_grad():
            numeric = float(((run(7, x + eps * v) * cot).sum().double() - (run(7, x - eps * v) * cot).sum().double()) / (2 * eps))
        assert abs(analytic - numeric) <= 3e-2 * max(1.0, abs(numeric)), (analytic, numeric)
    finally:
        ep.zero_()


def test_graphed_headline_shape_step_draws_new_masks_per_replay(gpu_device):
    """GraphedTrainStep on a model WITH FeedForward dropout (the headline layer shape, small grid): the captured step
    advances the device-side mask counter, so two replays on the same batch see different masks (different losses)
    while an evaluation forward stays deterministic; weights stay finite and the loss goes down over a few replays."""
    from models.ffno import FFNO2D
    from rpde import ops
    from rpde.graph import GraphedTrainStep
    from rpde.optim import FlatAdamW
    from utils.loss import RelativeL2Loss
    from utils.synthetic import advance, random_fields
    torch.manual_seed(0)
    m = FFNO2D(1, 1, width=64, n_layers=2, n_modes=12, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True,
               dropout=0.2).to(gpu_device).train()
    opt = FlatAdamW(m.parameters(), lr=1e-3, capturable=True)
    x = random_fields(4, 32, 2, seed=3)
    y = advance(x, 2).to(gpu_device)
    x = x.to(gpu_device)
    e0 = int(ops.drop_epoch(gpu_device).item())
    step = GraphedTrainStep(m, RelativeL2Loss(), opt, x, y, warmup=2)
    losses = [float(step(x, y)) for _ in range(12)]
    assert int(ops.drop_epoch(gpu_device).item()) == e0 + 2 + 12           # warm-up steps + replays (capture runs nothing)
    assert len({round(v, 7) for v in losses}) > 8, losses                     # not one frozen mask
    assert all(torch.isfinite(p_).all() for p_ in m.parameters())
    assert min(losses[-3:]) < losses[0]
    ops.drop_epoch(gpu_device).zero_()


def test_main_2d_on_a_file_dataset_with_normalisers(gpu_device, tmp_path, capsys):
    """dataset=ns/ns_file: the reference's dataset_params convention -> dataloaders/ns_naive_markov.py (row f4),
    unit-Gaussian normalisers decoded inside the loss, on a small synthetic .npz of smooth advected fields."""
    import numpy as np
    from rpde.entry import run
    from utils.synthetic import advance, random_fields
    u0 = random_fields(24, 32, 2, seed=3)                       # [24,1,32,32]
    frames = [u0]
    for _ in range(7):
        frames.append(advance(frames[-1], 2))
    u = torch.cat(frames, dim=1).numpy().astype(np.float32) * 3.0 + 1.5          # [N,T,H,W], not unit scale
    np.savez(tmp_path / "ns_32_synth.npz", u=u)
    l2 = run(2, ["model=ffno_2d/ffno_2d", "dataset=ns/ns_file", "dataset.dataset_params.filename=ns_32_synth.npz",
                 f"dataset.dataset_params.saved_folder={tmp_path}", "model.width=16", "model.n_layers=2",
                 "model.n_modes=8", "model.factor=2", "training.epochs=6", "training.batch_size=8",
                 "training.learning_rate=0.003", "training.use_normalizer=true", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    import json
    e0 = json.loads([ln for ln in out.splitlines() if '"epoch": 0' in ln][0])
    assert l2 < 1.0 and l2 < e0["val_loss"], (l2, e0)


def _smooth_1d(n, res, steps, seed):
    from utils.synthetic import advance, random_fields
    frames = [random_fields(n, res, 1, seed=seed)]               # [n,1,res]
    for _ in range(steps):
        frames.append(advance(frames[-1], 1))
    return torch.cat(frames, dim=1).numpy()                      # [n,T,res]


def test_main_1d_on_ks_files(gpu_device, tmp_path, capsys):
    """dataset=ks/ks_naive: three .npz archives with the HDF5 member names -> dataloaders/ks_naive_markov.py; the
    6-tuple (train, val, test, rollout, x_normalizer, y_normalizer) is unpacked as main_1d.py:71-76 does, the
    rollout set drives the autoregressive evaluation"""
    import json
    import numpy as np
    from rpde.entry import run
    for split, name, n, seed in (("train", "KS_train_64.npz", 24, 1), ("valid", "KS_valid.npz", 6, 2), ("test", "KS_test.npz", 6, 3)):
        u = (_smooth_1d(n, 64, 8, seed) * 2.0 + 0.5).astype(np.float32)
        np.savez(tmp_path / name, **{f"{split}/pde_9-64": u, f"{split}/x": np.linspace(0, 64, 64, dtype=np.float32)})
    l2 = run(1, ["model=ffno_1d/ffno_1d", "dataset=ks/ks_naive", "dataset.dataset_params.filename=KS_train_64.npz",
                 "dataset.dataset_params.val_filename=KS_valid.npz", "dataset.dataset_params.test_filename=KS_test.npz",
                 f"dataset.dataset_params.saved_folder={tmp_path}", "dataset.dataset_params.reduced_resolution=1",
                 "dataset.rollout_steps=4", "model.width=16", "model.n_layers=2", "model.n_modes=8", "model.factor=2",
                 "training.epochs=6", "training.batch_size=16", "training.learning_rate=0.003",
                 "training.use_normalizer=true", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    e0 = json.loads([ln for ln in out.splitlines() if '"epoch": 0' in ln][0])
    assert l2 < e0["val_loss"], (l2, e0)
    roll = run.last["rollout_rel_l2"]
    assert set(roll) == {32, 64} and all(np.isfinite(v) for v in roll.values())


@pytest.mark.parametrize("norm", ["simple", "minmax"])
def test_main_1d_on_burgers_file(gpu_device, tmp_path, capsys, norm):
    """dataset=burger/burger_naive with both normalisations: "minmax" returns four range values instead of two
    normalisers (reference burger_naive_markov.py:449-453) and evaluate() de-normalises with them"""
    import json
    import numpy as np
    from rpde.entry import run
    u = (_smooth_1d(40, 64, 6, 7) * 2.0 + 0.5).astype(np.float32)
    np.savez(tmp_path / "burgers.npz", **{"tensor": u, "x-coordinate": np.linspace(0, 1, 64, dtype=np.float32)})
    l2 = run(1, ["model=fno_1d/fno_1d", "dataset=burger/burger_naive", "dataset.dataset_params.filename=burgers.npz",
                 f"dataset.dataset_params.saved_folder={tmp_path}", "dataset.dataset_params.reduced_resolution=1",
                 "dataset.dataset_params.reduced_batch=1", f"dataset.dataset_params.normalization_type={norm}",
                 "dataset.rollout_steps=3", "model.width=16", "model.modes=8", "training.epochs=5",
                 "training.batch_size=16", "training.learning_rate=0.003", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    e0 = json.loads([ln for ln in out.splitlines() if '"epoch": 0' in ln][0])
    assert np.isfinite(l2) and l2 < 1.5 * e0["val_loss"] + 1.0, (l2, e0)
    assert all(np.isfinite(v) for v in run.last["rollout_rel_l2"].values())


@pytest.mark.parametrize("eval_file", [True, False])
def test_main_2d_on_true_multires_files(gpu_device, tmp_path, capsys, eval_file):
    """dataset=ns/ns_naive_true_mres: the north-star dataset module.  Files at 32^2 plus stride-subsampled 16^2
    samples -> two resolution groups in one training run, every batch single-resolution"""
    import json
    import numpy as np
    from rpde.entry import run
    from utils.synthetic import advance, random_fields
    frames = [random_fields(20, 32, 2, seed=5)]
    for _ in range(6):
        frames.append(advance(frames[-1], 2))
    u = (torch.cat(frames, dim=1).numpy() * 3.0 + 1.0).astype(np.float32)         # [N,T,H,W]
    np.savez(tmp_path / "ns_32_1e-3.npz", u=u)
    l2 = run(2, ["model=ffno_2d/ffno_2d", "dataset=ns/ns_naive_true_mres", f"dataset.dataset_params.saved_folder={tmp_path}",
                 "dataset.dataset_params.file_extension=.npz", "dataset.dataset_params.viscosity=1e-3",
                 "dataset.dataset_params.data_mres_size={32: 20}", "dataset.dataset_params.add_res=[16]",
                 "dataset.dataset_params.add_res_samples={16: 10}", "dataset.dataset_params.downsample_from_res=32",
                 "dataset.dataset_params.use_low_pass_filter=false",
                 "dataset.dataset_params.eval_filename=ns_32_1e-3.npz" if eval_file else "dataset.dataset_params.eval_dataset_target=null",
                 f"dataset.dataset_params.eval_saved_folder={tmp_path}", "model.width=16", "model.n_layers=2", "model.n_modes=8",
                 "model.factor=2", "training.epochs=4", "training.batch_size=8", "training.learning_rate=0.003",
                 "training.use_normalizer=true", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    head = json.loads([ln for ln in out.splitlines() if '"train_batches"' in ln][0])
    assert head["train_batches"] == (16 * 5) // 8 + (8 * 5) // 8                 # 80 pairs at 32^2 + 40 pairs at 16^2
    assert np.isfinite(l2) and set(run.last["resolution_rel_l2"]) == {32}


def test_main_1d_on_ks_true_multires_files(gpu_device, tmp_path, capsys):
    """dataset=ks/ks_naive_true_mres: one file per resolution in the reference's directory layout plus stride-subsampled
    samples -> three resolution groups in one FFNO1D run; the all-resolution evaluation reads the top file through
    the single-file loader (eval_dataset_target), the rollout set comes from the multi-resolution loader"""
    import json
    import numpy as np
    from rpde.entry import run
    sub = "visc_0.075_L64.0_lmax8_et5.0_nte51_nt51"
    for res, n, seed in ((64, 30, 1), (32, 20, 2)):
        (tmp_path / f"res_{res}" / sub).mkdir(parents=True)
        u = (_smooth_1d(n, res, 8, seed) * 2.0 + 0.5).astype(np.float32)
        np.savez(tmp_path / f"res_{res}" / sub / "KS_train_2048.npz", **{f"train/pde_9-{res}": u})
    # the evaluation loader (ks_markov_dataset) reads one file per split from the top resolution's directory, under
    # its default names KS_valid.h5 / KS_test.h5 -- here their .npz stand-ins
    for split, name, seed in (("valid", "KS_valid.npz", 5), ("test", "KS_test.npz", 6)):
        np.savez(tmp_path / "res_64" / sub / name, **{f"{split}/pde_9-64": (_smooth_1d(6, 64, 8, seed) * 2.0 + 0.5).astype(np.float32)})
    l2 = run(1, ["model=ffno_1d/ffno_1d", "dataset=ks/ks_naive_true_mres", f"dataset.dataset_params.saved_folder={tmp_path}",
                 "dataset.dataset_params.data_mres_size={64: 30, 32: 20}", "dataset.dataset_params.add_res=[16]",
                 "dataset.dataset_params.add_res_samples={16: 10}", "dataset.dataset_params.downsample_from_res=64",
                 "dataset.dataset_params.eval_filename=KS_train_2048.npz",
                 f"dataset.dataset_params.eval_saved_folder={tmp_path}/res_64/{sub}", "dataset.original_res=64",
                 "dataset.max_test_resolution=64", "dataset.rollout_steps=4", "model.width=16", "model.n_layers=2",
                 "model.n_modes=8", "model.factor=2", "training.epochs=4", "training.batch_size=8",
                 "training.learning_rate=0.003", "training.use_normalizer=true", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    head = json.loads([ln for ln in out.splitlines() if '"train_batches"' in ln][0])
    # train split: 24 trajectories at 64, 16 at 32, 8 draws at 16; 8 pairs each (KS keeps the first step)
    assert head["train_batches"] == (24 * 8) // 8 + (16 * 8) // 8 + (8 * 8) // 8
    assert np.isfinite(l2) and all(np.isfinite(v) for v in run.last["rollout_rel_l2"].values())


def test_main_1d_on_burgers_true_multires_files(gpu_device, tmp_path, capsys):
    """dataset=burger/burger_naive_true_mres with the loader's "minmax" statistics on a mixed-grid training set (the
    reference's DataLoader-based statistics would refuse it) and FNO1d"""
    import json
    import numpy as np
    from rpde.entry import run
    for res, n, seed in ((64, 30, 3), (32, 20, 4)):
        (tmp_path / f"burgers_{res}_0.001").mkdir()
        u = (_smooth_1d(n, res, 7, seed) * 2.0 + 0.5).astype(np.float32)
        np.savez(tmp_path / f"burgers_{res}_0.001" / "1D_Burgers_Sols_Nu0.001.npz", tensor=u,
                 **{"x-coordinate": np.linspace(0, 1, res, dtype=np.float32)})
    l2 = run(1, ["model=fno_1d/fno_1d", "dataset=burger/burger_naive_true_mres", f"dataset.dataset_params.saved_folder={tmp_path}",
                 "dataset.dataset_params.viscosity=0.001", "dataset.dataset_params.data_mres_size={64: 30, 32: 20}",
                 "dataset.dataset_params.add_res=null", "dataset.dataset_params.downsample_from_res=64",
                 "dataset.dataset_params.normalization_type=minmax",
                 "dataset.dataset_params.eval_filename=1D_Burgers_Sols_Nu0.001.npz",
                 f"dataset.dataset_params.eval_saved_folder={tmp_path}/burgers_64_0.001", "dataset.original_res=64",
                 "dataset.max_test_resolution=64", "dataset.rollout_steps=3", "model.width=16", "model.modes=8",
                 "training.epochs=4", "training.batch_size=8", "training.learning_rate=0.003", f"checkpoint_dir={tmp_path}"])
    out = capsys.readouterr().out
    head = json.loads([ln for ln in out.splitlines() if '"train_batches"' in ln][0])
    assert head["train_batches"] == (24 * 6) // 8 + (16 * 6) // 8          # Burgers pairs drop the first step: T - 2 = 6
    assert np.isfinite(l2) and set(run.last["resolution_rel_l2"]) == {32, 64}


def test_warm_plans_leaves_no_plan_to_build_inside_a_step(gpu_device):
    """rpde.ops.warm_plans: after it has seen the run's grids, forward + backward at those grids add nothing to
    the plan cache (no hipMalloc / stream sync inside a training step); an unseen grid does"""
    from models.ffno import FFNO2D
    from models.fno import FNO1d
    from rpde._lib import load
    from rpde.ops import warm_plans
    from utils.loss import RelativeL2Loss
    lib = load()
    loss_fn = RelativeL2Loss(size_average=True)
    m2 = FFNO2D(1, 1, width=16, n_layers=2, n_modes=7, factor=2, n_ff_layers=2, layer_norm=True).to(gpu_device).train()
    m1 = FNO1d(1, 1, modes=9, width=8).to(gpu_device).train()
    warm_plans(m2, [44, 88], 2, device=gpu_device)
    warm_plans(m1, [200], 1, device=gpu_device)
    assert m2.training and m1.training
    n0 = lib.rpde_plan_cache_count()
    for r in (44, 88):
        x = torch.randn(2, 1, r, r, device=gpu_device)
        loss_fn(m2(x), torch.randn_like(x)).backward()
    x = torch.randn(3, 1, 200, device=gpu_device)
    loss_fn(m1(x), torch.randn_like(x)).backward()
    assert lib.rpde_plan_cache_count() == n0
    m2(torch.randn(1, 1, 52, 52, device=gpu_device))
    assert lib.rpde_plan_cache_count() > n0


def test_flat_adamw_matches_torch_adamw(gpu_device):
    """rpde.optim.FlatAdamW (one kernel over flat buffers) against torch.optim.AdamW on a model with complex and real
    parameters, weight decay, a learning-rate change and a parameter that takes no part in some steps; state_dict
    interchange in both directions; the capturable variant inside a hipGraph"""
    import copy
    from models.fno import FNO2d
    from rpde.graph import GraphedTrainStep
    from rpde.optim import FlatAdamW
    from utils.loss import RelativeL2Loss
    torch.manual_seed(3)
    m_t = FNO2d(1, 1, modes1=4, modes2=3, width=8, n_blocks=2).to(gpu_device).train()
    m_f = copy.deepcopy(m_t)
    extra_t = torch.nn.Parameter(torch.randn(7, device=gpu_device))           # used only in odd steps
    extra_f = torch.nn.Parameter(extra_t.detach().clone())
    o_t = torch.optim.AdamW(list(m_t.parameters()) + [extra_t], lr=2e-3, weight_decay=0.05)
    o_f = FlatAdamW(list(m_f.parameters()) + [extra_f], lr=2e-3, weight_decay=0.05)
    loss_fn = RelativeL2Loss(size_average=True)
    x = torch.randn(3, 1, 16, 20, device=gpu_device)
    y = torch.randn(3, 1, 16, 20, device=gpu_device)

    def rel(a, b):
        a = torch.view_as_real(a) if a.is_complex() else a
        b = torch.view_as_real(b) if b.is_complex() else b
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    for step in range(6):
        if step == 3:
            for o in (o_t, o_f):
                o.param_groups[0]["lr"] = 5e-4
        for model, opt, extra in ((m_t, o_t, extra_t), (m_f, o_f, extra_f)):
            opt.zero_grad(set_to_none=True) if opt is o_t else opt.zero_grad()
            loss = loss_fn(model(x), y)
            if step % 2:
                loss = loss + (extra * extra).sum() * 1e-2
            loss.backward()
            if opt is o_f:
                opt.bucket.detach_untouched()
            opt.step()
    for (n, a), b in zip(m_f.named_parameters(), m_t.parameters()):
        assert rel(a, b) < 2e-6, (n, rel(a, b))
    assert rel(extra_f, extra_t) < 2e-6
    sd_f, sd_t = o_f.state_dict(), o_t.state_dict()
    assert [float(s["step"]) for s in sd_f["state"].values()] == [float(s["step"]) for s in sd_t["state"].values()]
    assert float(list(sd_f["state"].values())[-1]["step"]) == 3.0            # the sometimes-unused parameter
    for sf, st_ in zip(sd_f["state"].values(), sd_t["state"].values()):
        assert rel(sf["exp_avg"], st_["exp_avg"]) < 1e-5 and rel(sf["exp_avg_sq"], st_["exp_avg_sq"]) < 1e-5
    # checkpoints interchange: torch -> flat, continue, compare with torch continuing
    o_f.load_state_dict(copy.deepcopy(sd_t))
    with torch.no_grad():
        for a, b in zip(list(m_f.parameters()) + [extra_f], list(m_t.parameters()) + [extra_t]):
            a.copy_(b)
    for model, opt in ((m_t, o_t), (m_f, o_f)):
        opt.zero_grad()
        loss_fn(model(x), y).backward()
        if opt is o_f:
            opt.bucket.detach_untouched()
        opt.step()
    for (n, a), b in zip(m_f.named_parameters(), m_t.parameters()):
        assert rel(a, b) < 2e-6, (n, rel(a, b))
    # capturable: the step counter lives on the device, the whole step replays as a graph
    m_g, m_e = copy.deepcopy(m_t), copy.deepcopy(m_t)
    o_g = FlatAdamW(m_g.parameters(), lr=1e-3, capturable=True)
    o_e = torch.optim.AdamW(m_e.parameters(), lr=1e-3)
    gstep = GraphedTrainStep(m_g, loss_fn, o_g, x, y, warmup=2)             # 2 warm-up + 1 captured (not replayed) step
    for _ in range(3):
        gstep(x, y)
    for _ in range(2 + 3):
        o_e.zero_grad()
        loss_fn(m_e(x), y).backward()
        o_e.step()
    for (n, a), b in zip(m_g.named_parameters(), m_e.parameters()):
        assert rel(a, b) < 5e-6, (n, rel(a, b))
    assert float(next(iter(o_g.state_dict()["state"].values()))["step"]) == 5.0


def test_flat_adamw_refuses_moved_parameters(gpu_device):
    from rpde.optim import FlatAdamW
    lin = torch.nn.Linear(8, 8).to(gpu_device)
    opt = FlatAdamW(lin.parameters(), lr=1e-3)
    lin.weight.data = lin.weight.data.clone()                  # what model.to(other_device) / assign=True loading would do
    opt.zero_grad()
    lin(torch.randn(2, 8, device=gpu_device)).sum().backward()
    with pytest.raises(RuntimeError, match="storage changed"):
        opt.step()
    # load_state_dict into the model (copy_ into the existing storage) is fine
    lin2 = torch.nn.Linear(8, 8).to(gpu_device)
    opt2 = FlatAdamW(lin2.parameters(), lr=1e-3)
    lin2.load_state_dict(torch.nn.Linear(8, 8).state_dict())
    opt2.zero_grad()
    lin2(torch.randn(2, 8, device=gpu_device)).sum().backward()
    opt2.step()


def test_flat_adamw_loads_a_torch_state_dict_with_a_never_stepped_parameter(gpu_device):
    """torch.optim.AdamW keeps no state for a parameter that never received a gradient (fourier_weight in
    mode='low-pass', an unused forecast_ff): loading such a checkpoint must not raise, and must leave that
    parameter with fresh moments and step 0"""
    from rpde.optim import FlatAdamW
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(8, 4, device=gpu_device))
    b = torch.nn.Parameter(torch.randn(5, device=gpu_device))          # never used
    ref = torch.optim.AdamW([a, b], lr=1e-2)
    for _ in range(3):
        ref.zero_grad()
        (a.square().sum()).backward()
        ref.step()
    sd = ref.state_dict()
    assert 1 not in sd["state"]                                       # torch has nothing for `b`
    a2, b2 = torch.nn.Parameter(a.detach().clone()), torch.nn.Parameter(b.detach().clone())
    opt = FlatAdamW([a2, b2], lr=1e-2)
    opt.load_state_dict(sd)
    assert opt._steps == [3, 0]
    assert float(opt.state[b2]["exp_avg"].abs().max()) == 0.0
    # one more step on both sides, `b` now in the graph: identical updates
    for o, (pa, pb) in ((ref, (a, b)), (opt, (a2, b2))):
        o.zero_grad()
        (pa.square().sum() + pb.sum()).backward()
        o.step()
    assert torch.allclose(a, a2, rtol=1e-6, atol=1e-7) and torch.allclose(b, b2, rtol=1e-6, atol=1e-7)


def test_graphed_step_follows_or_refuses_a_changed_learning_rate(gpu_device):
    """FlatAdamW(capturable=True) keeps lr / weight decay on the device (rpde_adamw_set_hyper_dev): a replay after a
    scheduler step trains at the new rate -- the same weights as eager steps at that rate.  An optimizer that bakes the
    rate into the captured launches is refused instead of silently training at the old one."""
    import copy
    from models.fno import FNO1d
    from rpde.graph import GraphedTrainStep
    from rpde.optim import FlatAdamW
    from utils.loss import RelativeL2Loss
    torch.manual_seed(0)
    m = FNO1d(1, 1, modes=8, width=16).to(gpu_device).train()
    m_ref = copy.deepcopy(m)
    loss_fn = RelativeL2Loss()
    opt = FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, capturable=True)
    x, y = torch.randn(4, 1, 128, device=gpu_device), torch.randn(4, 1, 128, device=gpu_device)
    step = GraphedTrainStep(m, loss_fn, opt, x, y, warmup=2)
    step(x, y)
    opt.param_groups[0]["lr"] = 5e-4                                   # what a scheduler does
    opt.param_groups[0]["weight_decay"] = 3e-2
    step(x, y)
    step(x, y)
    ref = FlatAdamW(m_ref.parameters(), lr=1e-3, weight_decay=1e-2)
    for i in range(5):                                                 # 2 warm-up + 1 at the old rate, 2 at the new one
        if i == 3:
            ref.param_groups[0]["lr"], ref.param_groups[0]["weight_decay"] = 5e-4, 3e-2
        ref.zero_grad()
        loss_fn(m_ref(x), y).backward()
        ref.step()
    for pg, pe in zip(m.parameters(), m_ref.parameters()):
        a, b = (torch.view_as_real(t) if t.is_complex() else t for t in (pg, pe))
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 1e-6
    m2 = FNO1d(1, 1, modes=8, width=16).to(gpu_device).train()
    topt = torch.optim.AdamW(m2.parameters(), lr=1e-3, capturable=True)
    step2 = GraphedTrainStep(m2, loss_fn, topt, x, y)
    step2(x, y)
    topt.param_groups[0]["lr"] = 5e-4
    with pytest.raises(RuntimeError, match="changed after capture"):
        step2(x, y)


def test_rollout_fused_renormalisation_equals_decode_then_encode(gpu_device):
    """f3: between rollout steps the reference decodes with the y-statistics and encodes with the x-statistics
    (utils/autoregressive_step.py:296-303); with global statistics the product does it as one multiply-add"""
    from dataloaders.ns_naive_markov import SimpleNormalizer
    from models.ffno import FFNO1D
    from utils.autoregressive_step import perform_rollout_1d, rollout_loss
    torch.manual_seed(0)
    m = FFNO1D(1, 1, width=16, n_layers=2, n_modes=8, factor=2, n_ff_layers=2, layer_norm=True).to(gpu_device).eval()
    xn, yn = SimpleNormalizer(0.3, 1.7), SimpleNormalizer(-0.2, 0.9)
    x0 = torch.randn(5, 64, device=gpu_device)
    fused = perform_rollout_1d(m, x0, 4, device=gpu_device, x_normalizer=xn, y_normalizer=yn)
    # the two-call leg, step by step
    state, ref = x0, []
    with torch.no_grad():
        for _ in range(4):
            nxt = m(state.unsqueeze(1)).squeeze(1)
            ref.append(nxt.unsqueeze(1))
            state = xn.encode(yn.decode(nxt, device=gpu_device))
    ref = torch.cat(ref, dim=1)
    assert float((fused - ref).norm() / ref.norm()) < 1e-5
    traj = torch.randn(5, 5, 64, device=gpu_device)
    per_step = sum(float((fused[:, t] - traj[:, t + 1]).norm(dim=-1).div(traj[:, t + 1].norm(dim=-1) + 1e-8).mean()) for t in range(4)) / 4
    assert abs(rollout_loss(fused, traj) - per_step) < 1e-5
