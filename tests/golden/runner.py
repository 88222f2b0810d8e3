"""Runs one golden case on a backend and returns the named result tensors.

Backends:
  * ``ModuleBackend(ns)``  -- ``ns`` exposes nn.Module classes with the
    reference's names (the imported reference in make_golden.py; the HIP
    drop-in modules in the GPU tests).
  * ``OracleBackend()``    -- the functional CPU restatement in ``oracle/``.
Both consume the same numpy-seeded state_dict / inputs (synth.py).
"""
from __future__ import annotations

from typing import Callable, Dict, Mapping, Optional

import torch

from . import synth
from .cases import MODEL_KINDS

LAYER_KINDS = ("SpectralConv1d", "SpectralConv2d", "FSpectralConv1d", "FSpectralConv2d",
               "FeedForward", "WNLinear", "FNOBlock1d", "FNOBlock2d", "MLP1d", "MLP2d")


class Instance:
    """leaves: name -> grad-requiring tensor; call(x) -> tensor or tuple."""

    def __init__(self, leaves: Dict[str, torch.Tensor], call: Callable, loss: Callable):
        self.leaves, self.call, self.loss = leaves, call, loss


class ModuleBackend:
    def __init__(self, ns, device: str = "cpu"):
        self.ns, self.device = ns, device

    @staticmethod
    def ctor(case) -> Dict:
        c = dict(case["ctor"])
        if "act" in case:           # activation callables are not JSON data: named in the case
            import torch.nn.functional as F
            c["activation"] = {"gelu": F.gelu, "relu": F.relu}[case["act"]]
        return c

    def spec(self, case) -> Dict:
        kind = case.get("model", case["kind"])
        if kind in ("Resize1d", "Resize2d"):
            return {}
        if kind == "Rollout1d":
            kind = "FFNO1D"
        return synth.spec_of(getattr(self.ns, kind)(**self.ctor(case)).state_dict())

    def make(self, case, sd: Mapping[str, torch.Tensor]) -> Instance:
        kind = case.get("model", case["kind"])
        if kind == "Rollout1d":
            kind = "FFNO1D"
        if kind == "RelativeL2Loss":
            mod = self.ns.RelativeL2Loss(**case["ctor"])
            return Instance({}, None, mod)
        if kind in ("Resize1d", "Resize2d"):
            fn = self.ns.resize_1d if kind == "Resize1d" else self.ns.resize
            return Instance({}, fn, None)
        mod = getattr(self.ns, kind)(**self.ctor(case))
        got = synth.spec_of(mod.state_dict())
        want = synth.spec_of(sd)
        assert got == want, f"state_dict layout differs for {case['name']}:\n{got}\n{want}"
        mod.load_state_dict({k: v.clone() for k, v in sd.items()})
        mod = mod.to(self.device)
        mod.train()
        leaves = dict(mod.named_parameters())
        loss = self.ns.RelativeL2Loss(size_average=True)
        inst = Instance(leaves, mod, loss)
        inst.module = mod
        return inst


class OracleBackend:
    device = "cpu"

    def spec(self, case):
        raise RuntimeError("the oracle takes its layout from the fixture")

    def make(self, case, sd: Mapping[str, torch.Tensor]) -> Instance:
        from oracle import reference_path as R
        kind = case.get("model", case["kind"])
        c = dict(case["ctor"])
        if kind == "RelativeL2Loss":
            return Instance({}, None, lambda a, b: R.relative_l2(a, b, **c))
        if kind == "Resize1d":
            return Instance({}, R.resize_1d, None)
        if kind == "Resize2d":
            return Instance({}, R.resize_2d, None)
        p = R.make_params(sd)
        if kind == "SpectralConv1d":
            call = lambda x: R.spectral_conv1d(x, p["weights1"])
        elif kind == "SpectralConv2d":
            call = lambda x: R.spectral_conv2d(x, p["weights1"], p["weights2"])
        elif kind == "FSpectralConv1d":
            def call(x):
                t = x
                if c.get("mode", "full") != "no-fourier":
                    t = R.fspectral1d_fourier(t, p["fourier_weight.0"], c["modes"], c.get("mode", "full"),
                                              c.get("fft_norm", "ortho"))
                t = R.feedforward(t, p, "backcast_ff.", c.get("n_ff_layers", 2), c.get("layer_norm", False))
                return R._act(c.get("activation", "identity"))(t), None
        elif kind == "FSpectralConv2d":
            def call(x):
                t = x
                if c.get("mode", "full") != "no-fourier":
                    t = R.fspectral2d_fourier(t, p["fourier_weight.0"], p["fourier_weight.1"], c["modes"],
                                              c.get("mode", "full"))
                b = R.feedforward(t, p, "backcast_ff.", c.get("n_ff_layers", 2), c.get("layer_norm", False))
                f = None
                if c.get("use_fork", False):
                    f = R.feedforward(t, p, "forecast_ff.", c.get("n_ff_layers", 2), c.get("layer_norm", False))
                return b, f
        elif kind == "FeedForward":
            call = lambda x: R.feedforward(x, p, "", c.get("n_layers", 2), c.get("layer_norm", False))
        elif kind == "WNLinear":
            call = lambda x: R.wn_linear(x, p, "")
        elif kind in ("FNOBlock1d", "FNOBlock2d"):
            call = lambda x: R.fno_block(p, x, "", case.get("act", "relu" if kind == "FNOBlock1d" else "gelu"))
        elif kind in ("MLP1d", "MLP2d"):
            call = lambda x: R.conv_mlp(p, x, "")
        elif kind == "FNO1d":
            call = lambda x: R.fno1d_forward(p, x, c.get("n_blocks", 4), case.get("act", "relu"))
        elif kind == "FNO2d":
            call = lambda x: R.fno2d_forward(p, x, c.get("n_blocks", 4), case.get("act", "gelu"))
        elif kind in ("FFNO1D", "Rollout1d"):
            call = lambda x: R.ffno1d_forward(
                p, x, c.get("n_layers", 4), c.get("n_modes", 16), c.get("n_ff_layers", 2),
                c.get("layer_norm", False), 0.0, c.get("mode", "full"), c.get("fft_norm", "ortho"),
                c.get("activation", "identity"), c.get("grid", None))
        elif kind == "FFNO2D":
            call = lambda x: R.ffno2d_forward(
                p, x, c.get("n_layers", 4), c.get("n_modes", 16), c.get("n_ff_layers", 2),
                c.get("layer_norm", False), 0.0, c.get("mode", "full"), c.get("use_grid", True))
        else:
            raise KeyError(kind)
        return Instance(p, call, lambda a, b: R.relative_l2(a, b))


def _out_shape(case):
    x = case["x"]
    return (x[0], case["ctor"]["out_channels"]) + tuple(x[2:])


def run_case(case, backend, sd: Optional[Mapping[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Returns {result-name: tensor (on CPU)} for one case."""
    dev = backend.device
    kind, seed = case["kind"], case["seed"]
    res: Dict[str, torch.Tensor] = {}
    inst = backend.make(case, sd if sd is not None else {})

    if kind == "RelativeL2Loss":
        x = synth.rand_tensor(case["x"], seed, "x").to(dev).requires_grad_(True)
        y = synth.rand_tensor(case["x"], seed, "y")
        if "zero_target_row" in case:
            y[case["zero_target_row"]] = 0.0
        y = y.to(dev)
        out = inst.loss(x, y)
        out.sum().backward()
        res["out"], res["dx"] = out, x.grad
    elif kind in LAYER_KINDS:
        x = synth.rand_tensor(case["x"], seed, "x").to(dev).requires_grad_(True)
        out = inst.call(x)
        f = None
        if isinstance(out, tuple):
            out, f = out
        total = (out * synth.rand_tensor(out.shape, seed, "cot").to(dev)).sum()
        res["out"] = out
        if f is not None:
            total = total + (f * synth.rand_tensor(f.shape, seed, "cot_f").to(dev)).sum()
            res["out_f"] = f
        total.backward()
        res["dx"] = x.grad
        for k, v in inst.leaves.items():
            res["grad/" + k] = v.grad
    elif kind in MODEL_KINDS:
        x = synth.smooth_field(case["x"], seed, "x").to(dev)
        y = synth.smooth_field(_out_shape(case), seed, "y").to(dev)
        if case.get("grads", True):
            x.requires_grad_(True)
            pred = inst.call(x)
            loss = inst.loss(pred, y)
            loss.backward()
            res["dx"] = x.grad
            for k, v in inst.leaves.items():
                res["grad/" + k] = v.grad
        else:
            with torch.no_grad():
                pred = inst.call(x)
                loss = inst.loss(pred, y)
        res["out"], res["loss"] = pred, loss
    elif kind == "AdamWStep":
        x = synth.smooth_field(case["x"], seed, "x").to(dev)
        y = synth.smooth_field(_out_shape(case), seed, "y").to(dev)
        params = list(inst.leaves.values())
        opt = torch.optim.AdamW(params, lr=case["lr"])
        opt.zero_grad()
        loss = inst.loss(inst.call(x), y)
        loss.backward()
        opt.step()
        res["loss"] = loss
        for k, v in inst.leaves.items():
            res["param/" + k] = v
    elif kind in ("Resize1d", "Resize2d"):
        x = synth.rand_tensor(case["x"], seed, "x").to(dev)
        out = case["out"]
        with torch.no_grad():
            res["out"] = inst.call(x, tuple(out) if kind == "Resize2d" else out)
    elif kind == "Rollout1d":
        state = synth.smooth_field((case["x"][0], 1, case["x"][1]), seed, "x")[:, 0].to(dev)
        mean, std = case["mean"], case["std"]
        outs = []
        with torch.no_grad():
            for _ in range(case["steps"]):
                nxt = inst.call(state.unsqueeze(1)).squeeze(1)
                outs.append(nxt.unsqueeze(1))
                state = ((nxt * std + mean) - mean) / std
        res["out"] = torch.cat(outs, dim=1)
    else:
        raise KeyError(kind)
    return {k: v.detach().cpu() for k, v in res.items() if v is not None}
