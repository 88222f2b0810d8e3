"""AdamW with one kernel per step.

``FlatAdamW(params, lr, betas, eps, weight_decay)`` applies the update rule of ``torch.optim.AdamW`` as the
reference builds it (main_1d.py:144, main_2d.py:173: decoupled weight decay, bias-corrected moments, no amsgrad,
one parameter group) through ``rpde_adamw_step``: parameters, gradients and both moments live in flat fp32
buffers with one layout (the parameters' ``.data`` / ``.grad`` and ``state[p]['exp_avg']`` / ``['exp_avg_sq']`` are
views), so a step is one streaming pass instead of torch's eight multi-tensor passes -- 0.13 ms of a 1 ms FNO1d
step.  ``state_dict()`` has torch.optim.AdamW's format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so
checkpoints interchange with the reference's optimizer.

The gradient buffer is a ``rpde.parallel.FlatGradBucket`` (created here unless one is passed): ``zero_grad()`` is
``bucket.zero()``, the data-parallel exchange is ``bucket.all_reduce_mean()``.  A parameter whose ``.grad`` is
None at ``step()`` (outside this step's graph) is skipped exactly as torch skips it:
no moment update, no weight decay, its step count stands still.

``capturable=True`` keeps the step counter on the device (``rpde_adamw_step_dev``), which lets
``rpde.graph.GraphedTrainStep`` capture the step; it requires every parameter to take part in every step.
``lr`` and ``weight_decay`` live in that device state too: an eager ``step()`` stores the group's current values there, a
captured one reads them, and ``GraphedTrainStep`` calls ``sync_hyper_to_device()`` before a replay once a scheduler has
moved them -- a captured step follows the schedule without being captured again.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from ._lib import check, load, stream_ptr
from .parallel import FlatGradBucket


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 bucket: Optional[FlatGradBucket] = None, capturable: bool = False):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FlatAdamW: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=capturable))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW: one parameter group (as the reference's optimizers have)")
        group_params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        self.bucket = bucket if bucket is not None else FlatGradBucket(group_params)
        if [id(p) for p in self.bucket.params] != [id(p) for p in group_params]:
            raise ValueError("FlatAdamW: the bucket must hold exactly this optimizer's parameters, in order")
        if not self.bucket.flat.is_cuda:
            raise ValueError("FlatAdamW runs on the GPU (no CPU fallback)")
        dev = self.bucket.flat.device
        # two regions, as in the bucket (real parameters, then complex ones), each in a storage of its own: a state_dict
        # may not hold float and complex views of one storage (torch.save refuses it)
        bounds = [(0, self.bucket.n_real), (self.bucket.n_real, self.bucket.flat.numel())]
        self._bounds = bounds
        self._p = [torch.zeros(hi - lo, dtype=torch.float32, device=dev) for lo, hi in bounds]
        self._m = [torch.zeros(hi - lo, dtype=torch.float32, device=dev) for lo, hi in bounds]
        self._v = [torch.zeros(hi - lo, dtype=torch.float32, device=dev) for lo, hi in bounds]
        self._offsets: List[int] = list(self.bucket.offsets)
        self._sizes: List[int] = [p.numel() * (2 if p.is_complex() else 1) for p in group_params]
        self._steps: List[int] = [0] * len(group_params)
        self._step_dev = torch.zeros(8, dtype=torch.float32, device=dev) if capturable else None
        for i, (p, gview) in enumerate(zip(group_params, self.bucket._views)):
            r = 1 if p.is_complex() else 0
            off, size = self._offsets[i] - bounds[r][0], self._sizes[i]

            def view(flat):
                chunk = flat[r][off:off + size]
                return torch.view_as_complex(chunk.view(*p.shape, 2)) if p.is_complex() else chunk.view(p.shape)
            pv = view(self._p)
            with torch.no_grad():
                pv.copy_(p.data)
            p.data = pv                                         # the parameter now lives in the flat buffer
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": view(self._m), "exp_avg_sq": view(self._v)}
            assert gview.data_ptr() == self.bucket.flat.data_ptr() + 4 * self._offsets[i]

    def zero_grad(self, set_to_none: bool = False) -> None:     # noqa: ARG002
        """Every ``.grad`` becomes None whatever ``set_to_none`` says (``bucket.zero()``): backward then ASSIGNS each
        parameter's first gradient, and ``gather()`` brings them into the flat buffer.  Code that accumulates into a
        pre-zeroed ``.grad`` or reads ``p.grad`` right after ``zero_grad()`` sees None, as after torch's default
        ``zero_grad(set_to_none=True)``."""
        self.bucket.zero()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(g["lr"]), g["betas"], float(g["eps"]), float(g["weight_decay"])
        self.bucket.gather()                                    # gradients into the flat buffer (no-op when done)
        # the update kernels write the parameters through raw pointers (no _version bump): prepared weight fragments of an
        # open rpde.ops.frozen_weights() scope would go stale silently
        from . import ops as _ops
        _ops.invalidate_frozen()
        params = self.bucket.params
        lib, st = load(), stream_ptr()
        base_g = self.bucket.flat.data_ptr()
        live = []
        for i, p in enumerate(params):
            r = 1 if p.is_complex() else 0
            if p.data_ptr() != self._p[r].data_ptr() + 4 * (self._offsets[i] - self._bounds[r][0]):
                raise RuntimeError("FlatAdamW: a parameter's storage changed after the optimizer was built (model.to(), "
                                   "load with assign=True ...): build the optimizer after the model is in place")
            if p.grad is None:
                continue
            if p.grad.data_ptr() != base_g + 4 * self._offsets[i]:     # .grad replaced after gather(): bring it in
                self.bucket._views[i].copy_(p.grad)
            live.append(i)
        if self._step_dev is not None:
            if len(live) != len(params):
                raise RuntimeError("FlatAdamW(capturable=True): every parameter must receive a gradient in every step")
            for i in live:
                self._steps[i] += 1
            ticked = False
            for r, (lo, hi) in enumerate(self._bounds):          # the counter advances once per step, not per region
                if hi == lo:
                    continue
                fn = lib.rpde_adamw_apply_dev if ticked else lib.rpde_adamw_step_dev
                check(fn(self._p[r].data_ptr(), base_g + 4 * lo, self._m[r].data_ptr(), self._v[r].data_ptr(), hi - lo,
                         lr, b1, b2, eps, wd, self._step_dev.data_ptr(), st), "adamw_step_dev")
                ticked = True
            return loss
        # runs of parameters that are neighbours in a region, live, and share a step count -> one launch each
        # (normally: one per region)
        order = sorted(live, key=lambda i: self._offsets[i])
        k = 0
        while k < len(order):
            j = k
            i0 = order[k]
            r = 1 if params[i0].is_complex() else 0
            t = self._steps[i0] + 1
            end = self._offsets[i0] + (self._sizes[i0] + 3) // 4 * 4
            while (j + 1 < len(order) and self._offsets[order[j + 1]] == end and self._steps[order[j + 1]] + 1 == t
                   and params[order[j + 1]].is_complex() == bool(r)):
                j += 1
                end = self._offsets[order[j]] + (self._sizes[order[j]] + 3) // 4 * 4
            lo = self._offsets[i0]
            rl = lo - self._bounds[r][0]
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            check(lib.rpde_adamw_step(self._p[r].data_ptr() + 4 * rl, base_g + 4 * lo, self._m[r].data_ptr() + 4 * rl,
                                      self._v[r].data_ptr() + 4 * rl, end - lo, 1.0 - lr * wd, 1.0 - b1, b2, 1.0 - b2,
                                      lr / bc1, math.sqrt(bc2), eps, st), "adamw_step")
            for i in order[k:j + 1]:
                self._steps[i] = t
            k = j + 1
        return loss

    def sync_hyper_to_device(self) -> None:
        """capturable only: put the group's current lr / weight_decay where captured steps read them (rpde.graph calls
        this before a replay when a scheduler has moved them)"""
        if self._step_dev is None:
            raise RuntimeError("FlatAdamW.sync_hyper_to_device: build the optimizer with capturable=True")
        g = self.param_groups[0]
        check(load().rpde_adamw_set_hyper_dev(self._step_dev.data_ptr(), float(g["lr"]), float(g["weight_decay"]), stream_ptr()),
              "adamw_set_hyper_dev")

    # ---- torch.optim.AdamW-compatible checkpoints ---------------------------------------------------------------
    def state_dict(self):
        if self._step_dev is not None:                          # graph replays advance only the device counter
            self._steps = [int(float(self._step_dev[0]))] * len(self._steps)
        for i, p in enumerate(self.bucket.params):
            self.state[p]["step"] = torch.tensor(float(self._steps[i]))
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        views = {id(p): (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in self.bucket.params}
        super().load_state_dict(state_dict)
        for i, p in enumerate(self.bucket.params):
            st = self.state[p]
            m, v = views[id(p)]
            # torch.optim.AdamW keeps no state for a parameter that never received a gradient (fourier_weight in
            # mode='low-pass', an unused forecast_ff): such an entry is absent or empty -> fresh moments, step 0
            with torch.no_grad():
                if "exp_avg" in st and "exp_avg_sq" in st:
                    m.copy_(st["exp_avg"].to(m.device))
                    v.copy_(st["exp_avg_sq"].to(v.device))
                else:
                    m.zero_()
                    v.zero_()
            st["exp_avg"], st["exp_avg_sq"] = m, v             # keep the state inside the flat buffers
            self._steps[i] = int(float(st["step"])) if "step" in st else 0
            st.setdefault("step", torch.tensor(0.0))
        if self._step_dev is not None:
            self._step_dev.zero_()
            self._step_dev[0] = float(self._steps[0]) if self._steps else 0.0
