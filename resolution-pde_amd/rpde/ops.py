"""torch.autograd.Function wrappers over the C ABI (include/rpde.h).

PyTorch is plumbing here: it owns device memory (outputs, tensors saved for
backward, workspaces all come from its caching allocator) and provides the
current HIP stream.  Every numerical step of the hot path runs in
librpde_hip.so.
"""
from __future__ import annotations

import contextlib
import os

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT, MODE, NORM, check, load, ptr, ptr_array, stream_ptr, workspace


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _as_float_storage(w: torch.Tensor) -> torch.Tensor:
    """complex64 [..] -> float32 [..,2] view of the same memory"""
    w = w if w.is_contiguous() else w.contiguous()
    return torch.view_as_real(w) if w.is_complex() else w


# ----------------------------------------------------------------------------
# frozen weights: evaluation loops (validation, all-resolution sweeps, rollouts -- reference train/training.py:78-123,
# utils/autoregressive_step.py) call the same layers hundreds of times with weights nobody touches.  Inside
# ``with frozen_weights():`` the no-grad paths of the FeedForward and of the 2-D spectral layer build their weight
# fragments (weight-norm / f16 pieces / mode-mix B operands) ONCE per layer and reuse them; the scope's end drops them.
# Contract: inside the scope the weights are changed, if at all, by torch in-place ops (which bump ``_version`` and
# refresh the entry) or by FlatAdamW.step, which writes through raw pointers and therefore calls
# ``invalidate_frozen()`` itself; any other raw-pointer writer must do the same.
# ----------------------------------------------------------------------------
_FROZEN: Optional[dict] = None


def invalidate_frozen() -> None:
    """drop every prepared buffer of the open scope (the scope itself stays open): for code that changes parameters
    without bumping ``_version`` -- FlatAdamW.step, a load through raw pointers"""
    if _FROZEN is not None:
        _FROZEN.clear()


@contextlib.contextmanager
def frozen_weights():
    global _FROZEN
    outer = _FROZEN
    if outer is None:
        _FROZEN = {}
    try:
        yield
    finally:
        if outer is None:
            _FROZEN = None


def _frozen_entry(kind, tensors, extra, nbytes, build, originals=None):
    """prepared buffer for these weight tensors, built by build(buf) on first use; None when no scope is open (or while
    a HIP graph is being captured: a graph must not hold a pointer whose life ends with the scope)"""
    if _FROZEN is None or torch.cuda.is_current_stream_capturing():
        return None
    # originals: what the caller was handed before _f32c.  Where _f32c had to copy (a parameter in another dtype or
    # layout) the copy has a new address on every call: caching by it would re-prepare every time and keep every copy
    # alive until the scope ends -- such weights take the unprepared path.
    if originals is not None and any(t is not o for t, o in zip(tensors, originals)):
        return None
    key = (kind, extra) + tuple(t.data_ptr() for t in tensors)
    ver = tuple(t._version for t in tensors)
    hit = _FROZEN.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    buf = hit[1] if hit is not None else torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device)
    build(buf)
    _FROZEN[key] = (ver, buf, tensors)          # (the tensors are held so that no data_ptr can be reused by another)
    return buf


# ----------------------------------------------------------------------------
# FSpectralConv{1,2}d.forward_fourier
# ----------------------------------------------------------------------------
class _FSpectral1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, modes: int, mode: int, norm: int, with_skip: bool = False):
        lib = load()
        x = _f32c(x)
        wf = _f32c(w) if w is not None else None
        B, n, Cc = x.shape
        out = torch.empty_like(x)
        spec = torch.empty(lib.rpde_fspectral1d_spec_elems(B, n, Cc, modes), dtype=torch.float32, device=x.device)
        nws = lib.rpde_fspectral1d_ws_bytes(B, n, Cc, modes)
        ws = workspace(nws, x.device)
        check(lib.rpde_fspectral1d_fwd(ptr(x), ptr(wf), ptr(out), ptr(spec), B, n, Cc, modes, mode, norm,
                                       ws.data_ptr(), nws, stream_ptr()), "fspectral1d_fwd")
        ctx.save_for_backward(spec, wf if wf is not None else x.new_empty(0))
        ctx.dims = (B, n, Cc, modes, mode, norm, wf is not None)
        # with_skip: also hand x back (an alias) for the caller's skip connection, so that the gradient
        # arriving through the skip is summed by the last backward GEMM's epilogue, not by a separate pass
        return (out, x.view_as(x)) if with_skip else out

    @staticmethod
    def backward(ctx, g, g_skip=None):
        lib = load()
        spec, wf = ctx.saved_tensors
        B, n, Cc, modes, mode, norm, has_w = ctx.dims
        if g is None:
            return g_skip, None, None, None, None, None
        g = _f32c(g)
        g_skip = _f32c(g_skip) if g_skip is not None else None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and has_w
        gx = torch.empty_like(g) if need_x else None
        gw = torch.empty_like(wf) if need_w else None
        nws = lib.rpde_fspectral1d_ws_bytes(B, n, Cc, modes)
        ws = workspace(nws, g.device)
        check(lib.rpde_fspectral1d_bwd(ptr(g), ptr(spec), ptr(wf) if has_w else None, ptr(gx), ptr(gw),
                                       ptr(g_skip) if need_x else None, B, n, Cc, modes,
                                       mode, norm, ws.data_ptr(), nws, stream_ptr()), "fspectral1d_bwd")
        return gx, gw, None, None, None, None


class _FSpectral2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wy, wx, modes: int, mode: int, with_skip: bool = False):
        lib = load()
        x = _f32c(x)
        has_w = wy is not None
        wyf = _f32c(wy) if has_w else None
        wxf = _f32c(wx) if has_w else None
        B, M, N, Cc = x.shape
        out = torch.empty_like(x)
        spec_y = torch.empty(lib.rpde_fspectral2d_spec_elems(B, M, N, Cc, modes, 0), dtype=torch.float32, device=x.device)
        spec_x = torch.empty(lib.rpde_fspectral2d_spec_elems(B, M, N, Cc, modes, 1), dtype=torch.float32, device=x.device)
        nws = lib.rpde_fspectral2d_ws_bytes(B, M, N, Cc, modes)
        ws = workspace(nws, x.device)
        check(lib.rpde_fspectral2d_fwd(ptr(x), ptr(wyf), ptr(wxf), ptr(out), ptr(spec_y), ptr(spec_x), B, M, N, Cc, modes,
                                       mode, ws.data_ptr(), nws, stream_ptr()), "fspectral2d_fwd")
        if has_w:
            ctx.save_for_backward(spec_y, spec_x, wyf, wxf)
        else:
            ctx.save_for_backward(spec_y, spec_x)
        ctx.dims = (B, M, N, Cc, modes, mode, has_w)
        return (out, x.view_as(x)) if with_skip else out

    @staticmethod
    def backward(ctx, g, g_skip=None):
        lib = load()
        B, M, N, Cc, modes, mode, has_w = ctx.dims
        if g is None:
            return g_skip, None, None, None, None, None
        if has_w:
            spec_y, spec_x, wyf, wxf = ctx.saved_tensors
        else:
            (spec_y, spec_x), wyf, wxf = ctx.saved_tensors, None, None
        g = _f32c(g)
        g_skip = _f32c(g_skip) if g_skip is not None else None
        need_x = ctx.needs_input_grad[0]
        need_w = has_w and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gx = torch.empty_like(g) if need_x else None
        gwy = torch.empty_like(wyf) if need_w else None
        gwx = torch.empty_like(wxf) if need_w else None
        nws = lib.rpde_fspectral2d_ws_bytes(B, M, N, Cc, modes)
        ws = workspace(nws, g.device)
        check(lib.rpde_fspectral2d_bwd(ptr(g), ptr(spec_y), ptr(spec_x), ptr(wyf), ptr(wxf), ptr(gx), ptr(gwy), ptr(gwx),
                                       ptr(g_skip) if need_x else None,
                                       B, M, N, Cc, modes, mode, ws.data_ptr(), nws, stream_ptr()), "fspectral2d_bwd")
        return gx, gwy, gwx, None, None, None


def _fspectral2d_frozen(x, wy, wx, modes: int):
    """evaluation inside frozen_weights(): rpde_fspectral2d_prepare once per layer, then rpde_fspectral2d_fwd_prepared
    (the spectra live in the workspace: nothing is kept for a backward).  None: this shape has nothing to prepare."""
    lib = load()
    x = _f32c(x)
    B, M, N, Cc = x.shape
    npre = lib.rpde_fspectral2d_prep_bytes(M, N, Cc, modes)
    if npre == 0:
        return None
    wyf, wxf = _f32c(wy), _f32c(wx)
    prep = _frozen_entry("fs2d", (wyf, wxf), (M, N, Cc, modes), npre, lambda buf: check(
        lib.rpde_fspectral2d_prepare(ptr(wyf), ptr(wxf), M, N, Cc, modes, buf.data_ptr(), npre, stream_ptr()),
        "fspectral2d_prepare"), originals=(wy, wx))
    if prep is None:
        return None
    out = torch.empty_like(x)
    nws = lib.rpde_fspectral2d_eval_ws_bytes(B, M, N, Cc, modes)
    ws = workspace(nws, x.device)
    check(lib.rpde_fspectral2d_fwd_prepared(ptr(x), prep.data_ptr(), ptr(out), B, M, N, Cc, modes, ws.data_ptr(), nws,
                                            stream_ptr()), "fspectral2d_fwd_prepared")
    return out


def fspectral1d(x, w, modes: int, mode: str = "full", norm: str = "ortho", with_skip: bool = False):
    """FSpectralConv1d.forward_fourier: x [B,n,C], w [C,C,K,2].
    with_skip: returns (out, x') where x' aliases x -- use x' for a skip connection around the layer and
    its gradient is folded into the backward's last GEMM instead of a separate add."""
    if mode not in MODE:
        raise ValueError(f"Mode {mode} not recognized")
    return _FSpectral1d.apply(x, w if mode == "full" else None, int(modes), MODE[mode], NORM[norm], bool(with_skip))


def fspectral2d(x, wy, wx, modes: int, mode: str = "full", with_skip: bool = False):
    """FSpectralConv2d.forward_fourier: x [B,M,N,C], w_y/w_x [C,C,K,2].  with_skip: see fspectral1d."""
    if mode not in MODE:
        # the reference's 2-D layer has no else branch: both spectra stay zero
        return (torch.zeros_like(x), x) if with_skip else torch.zeros_like(x)
    full = mode == "full"
    if full and _FROZEN is not None and not torch.is_grad_enabled() and x.is_cuda:
        out = _fspectral2d_frozen(x, wy, wx, int(modes))
        if out is not None:
            return (out, x) if with_skip else out
    return _FSpectral2d.apply(x, wy if full else None, wx if full else None, int(modes), MODE[mode], bool(with_skip))


# ----------------------------------------------------------------------------
# dropout masks under a captured hipGraph: a replay repeats its launch arguments, so the seed the FeedForward draws on
# the host would freeze one mask for ever.  Every dropout kernel therefore also mixes a DEVICE counter into its seed
# (rpde_ff_params.seed_epoch); eager steps leave it at 0 (the host seed changes per call), a GraphedTrainStep advances it
# once per replay -- on the device, inside the graph -- before the forward, so forward and backward of one step agree.
# ----------------------------------------------------------------------------
_DROP_EPOCH: dict = {}


def drop_epoch(device) -> torch.Tensor:
    """the int64 [1] device counter of `device` (created on first use, 0)"""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _DROP_EPOCH.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        _DROP_EPOCH[key] = t
    return t


# ----------------------------------------------------------------------------
# FeedForward (+ residual / post-activation glue)
# ----------------------------------------------------------------------------
class _FeedForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, cfg, *params):
        """cfg = (n_layers, dim, factor, layer_norm, eps, dropout_p, seed, post_act)
        params = W0, b0, ..., W_{L-1}, b_{L-1} [, gamma, beta]"""
        lib = load()
        L, dim, factor, layer_norm, eps, p_drop, seed, post_act, grad_on = cfg
        shape = x.shape
        x2 = _f32c(x).reshape(-1, dim)
        P = x2.shape[0]
        res2 = _f32c(residual).reshape(-1, dim) if residual is not None else None
        ws_ = [_f32c(params[2 * l]) for l in range(L)]
        bs_ = [_f32c(params[2 * l + 1]) for l in range(L)]
        gamma = _f32c(params[2 * L]) if layer_norm else None
        beta = _f32c(params[2 * L + 1]) if layer_norm else None
        need_grad = grad_on and any(ctx.needs_input_grad)     # (grad mode is always off inside forward itself)
        hid = dim * factor
        # the fused kernel keeps the hidden activations on chip: in evaluation nothing but `out` is allocated
        # (the fused kernels' weight preparation reads 16 bytes at a time: a weight view at an odd storage offset takes the
        #  per-GEMM path inside the library, which needs the hidden buffers -- so it must not be "lean" here either)
        fused = bool(lib.rpde_feedforward_is_fused(dim, factor, L, P)) and all(w.data_ptr() % 16 == 0 for w in ws_)
        lean = (not need_grad) and fused
        # RPDE_FF_STASH=u: training through the fused kernels saves only u = dropout(z) of the hidden layers (in `hs`) and
        # the backward kernels re-evaluate gelu / gelu' from it -- half the saved-for-backward footprint, but measured
        # slower on MI355X (the erf-class vector work costs more than the 2 KB per point it keeps out of HBM:
        # DESIGN.md section 6).  Default: h and d = gelu'(u) * dropscale are stored once by the forward kernel.
        recompute = need_grad and fused and os.environ.get("RPDE_FF_STASH", "hd") == "u"
        hs = [None if lean else torch.empty(P, hid, dtype=torch.float32, device=x.device) for _ in range(L - 1)]
        ds = [torch.empty(P, hid, dtype=torch.float32, device=x.device) if (need_grad and not recompute) else None
              for _ in range(L - 1)]
        out = torch.empty(P, dim, dtype=torch.float32, device=x.device)
        # (the pre-LayerNorm tensor is saved for backward only: the evaluation kernel does not write it)
        z_last = out if lean else torch.empty(P, dim, dtype=torch.float32, device=x.device)
        wa, ba, ha, da = ptr_array(ws_), ptr_array(bs_), ptr_array(hs or [None]), ptr_array(ds or [None])
        epoch = drop_epoch(x.device).data_ptr() if p_drop > 0.0 else None
        fp = _lib.FFParams(L, dim, factor, int(layer_norm), eps, p_drop, seed, post_act,
                           C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p)), ptr(gamma), ptr(beta), epoch)
        nws = lib.rpde_feedforward_fwd_ws_bytes(dim, factor, L)
        if lean and p_drop == 0.0:
            held = tuple(ws_ + bs_ + ([gamma, beta] if layer_norm else []))
            orig = tuple([params[2 * l] for l in range(L)] + [params[2 * l + 1] for l in range(L)] +
                         ([params[2 * L], params[2 * L + 1]] if layer_norm else []))
            prep = _frozen_entry("ff", held, (L, dim, factor), nws, lambda buf: check(
                lib.rpde_feedforward_prepare(C.byref(fp), buf.data_ptr(), nws, stream_ptr()), "feedforward_prepare"),
                originals=orig)
            if prep is not None:
                check(lib.rpde_feedforward_fwd_prepared(C.byref(fp), ptr(x2), ptr(res2), ptr(out), P, prep.data_ptr(), nws,
                                                        stream_ptr()), "feedforward_fwd_prepared")
                return out.reshape(shape)
        ws = workspace(nws, x.device)
        check(lib.rpde_feedforward_fwd(C.byref(fp), ptr(x2), ptr(res2), C.cast(ha, C.POINTER(C.c_void_p)),
                                       C.cast(da, C.POINTER(C.c_void_p)), ptr(z_last), ptr(out), P, ws.data_ptr(), nws,
                                       stream_ptr()), "feedforward_fwd")
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        if need_grad:
            ctx.save_for_backward(x2, z_last, *hs, *ds, *ws_, *bs_, *([gamma, beta] if layer_norm else []))
        return out.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        lib = load()
        L, dim, factor, layer_norm, eps, p_drop, seed, post_act, _ = ctx.cfg
        saved = ctx.saved_tensors
        x2, z_last = saved[0], saved[1]
        o = 2
        hs, ds = list(saved[o:o + L - 1]), list(saved[o + L - 1:o + 2 * (L - 1)])
        o += 2 * (L - 1)
        ws_, bs_ = list(saved[o:o + L]), list(saved[o + L:o + 2 * L])
        gamma, beta = (saved[o + 2 * L], saved[o + 2 * L + 1]) if layer_norm else (None, None)
        P = x2.shape[0]
        g2 = _f32c(g).reshape(-1, dim)
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gws = [torch.empty_like(w) for w in ws_]
        gbs = [torch.empty_like(b) for b in bs_]
        ggamma = torch.empty_like(gamma) if layer_norm else None
        gbeta = torch.empty_like(beta) if layer_norm else None
        wa, ba, ha, da = ptr_array(ws_), ptr_array(bs_), ptr_array(hs or [None]), ptr_array(ds or [None])
        gwa, gba = ptr_array(gws), ptr_array(gbs)
        epoch = drop_epoch(g.device).data_ptr() if p_drop > 0.0 else None
        fp = _lib.FFParams(L, dim, factor, int(layer_norm), eps, p_drop, seed, post_act,
                           C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p)), ptr(gamma), ptr(beta), epoch)
        nws = lib.rpde_feedforward_ws_bytes(P, dim, factor, L)
        ws = workspace(nws, g.device)
        check(lib.rpde_feedforward_bwd(C.byref(fp), ptr(x2), C.cast(ha, C.POINTER(C.c_void_p)),
                                       C.cast(da, C.POINTER(C.c_void_p)), ptr(z_last), ptr(g2), ptr(gx),
                                       C.cast(gwa, C.POINTER(C.c_void_p)), C.cast(gba, C.POINTER(C.c_void_p)),
                                       ptr(ggamma), ptr(gbeta), P, ws.data_ptr(), nws, stream_ptr()), "feedforward_bwd")
        grads: List[Optional[torch.Tensor]] = []
        for l in range(L):
            grads += [gws[l], gbs[l]]
        if layer_norm:
            grads += [ggamma, gbeta]
        gres = g if ctx.has_res else None
        return (gx.reshape(g.shape) if gx is not None else None, gres, None, *grads)


def feedforward(x, residual, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], ln: Optional[Tuple],
                dim: int, factor: int, dropout_p: float, seed: int, post_act: str = "identity", eps: float = 1e-5):
    L = len(weights)
    params: List[torch.Tensor] = []
    for w, b in zip(weights, biases):
        params += [w, b]
    if ln is not None:
        params += [ln[0], ln[1]]
    cfg = (L, int(dim), int(factor), ln is not None, float(eps), float(dropout_p), int(seed) & (2 ** 64 - 1), ACT[post_act],
           torch.is_grad_enabled())
    return _FeedForward.apply(x, residual, cfg, *params)


# ----------------------------------------------------------------------------
# pointwise linear (channels-last)
# ----------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        lib = load()
        in_f, out_f = w.shape[1], w.shape[0]
        x2 = _f32c(x).reshape(-1, in_f)
        w = _f32c(w)
        b = _f32c(b) if b is not None else None
        P = x2.shape[0]
        out = torch.empty(P, out_f, dtype=torch.float32, device=x.device)
        check(lib.rpde_linear_fwd(ptr(x2), ptr(w), ptr(b), ptr(out), P, in_f, out_f, stream_ptr()), "linear_fwd")
        ctx.save_for_backward(x2, w)
        ctx.has_b = b is not None
        return out.reshape(*x.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, g):
        lib = load()
        x2, w = ctx.saved_tensors
        out_f, in_f = w.shape
        P = x2.shape[0]
        g2 = _f32c(g).reshape(-1, out_f)
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gb = torch.empty(out_f, dtype=torch.float32, device=g.device) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        nws = lib.rpde_linear_ws_bytes(P, in_f, out_f)
        ws = workspace(nws, g.device)
        check(lib.rpde_linear_bwd(ptr(x2), ptr(w), ptr(g2), ptr(gx), ptr(gw), ptr(gb), P, in_f, out_f, ws.data_ptr(), nws,
                                  stream_ptr()), "linear_bwd")
        return (gx.reshape(*g.shape[:-1], in_f) if gx is not None else None), gw, gb


def linear(x, w, b=None):
    return _Linear.apply(x, w, b)


class _WeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        lib = load()
        v = _f32c(v)
        g = _f32c(g)
        out_f, in_f = v.shape
        w = torch.empty_like(v)
        check(lib.rpde_weight_norm_fwd(ptr(v), ptr(g), ptr(w), out_f, in_f, stream_ptr()), "weight_norm_fwd")
        ctx.save_for_backward(v, g)
        return w

    @staticmethod
    def backward(ctx, gw):
        lib = load()
        v, g = ctx.saved_tensors
        out_f, in_f = v.shape
        gw = _f32c(gw)
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        if gv is None and gg is None:
            return None, None
        check(lib.rpde_weight_norm_bwd(ptr(v), ptr(g), ptr(gw), ptr(gv), ptr(gg), out_f, in_f, stream_ptr()), "weight_norm_bwd")
        return gv, gg


def weight_norm(v, g):
    """WNLinear's effective weight v * (g / |v|_row) (models/custom_layer.py:70-108), v [out,in], g [out,1]"""
    return _WeightNorm.apply(v, g)


# ----------------------------------------------------------------------------
# FNO: channels-first spectral conv, 1x1 conv, activation
# ----------------------------------------------------------------------------
class _Spectral1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, act_in: int):
        lib = load()
        x = _f32c(x)
        wf = _as_float_storage(w)
        B, Ci, n = x.shape
        Co, K = w.shape[1], w.shape[2]
        out = torch.empty(B, Co, n, dtype=torch.float32, device=x.device)
        kp = (K + 3) // 4 * 4
        spec = torch.empty(B * Ci * 2 * kp, dtype=torch.float32, device=x.device)
        nws = lib.rpde_spectral1d_ws_bytes(B, Ci, Co, n, K)
        ws = workspace(nws, x.device)
        check(lib.rpde_spectral1d_fwd(ptr(x), ptr(wf), ptr(out), ptr(spec), B, Ci, Co, n, K, act_in, ws.data_ptr(), nws,
                                      stream_ptr()), "spectral1d_fwd")
        ctx.save_for_backward(spec, w, x if act_in else x.new_empty(0))
        ctx.dims = (B, Ci, Co, n, K, act_in)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        spec, w, x = ctx.saved_tensors
        B, Ci, Co, n, K, act_in = ctx.dims
        g = _f32c(g)
        wf = _as_float_storage(w)
        gx = torch.empty(B, Ci, n, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        nws = lib.rpde_spectral1d_ws_bytes(B, Ci, Co, n, K)
        ws = workspace(nws, g.device)
        check(lib.rpde_spectral1d_bwd(ptr(g), ptr(spec), ptr(wf), ptr(x) if act_in else None, ptr(gx),
                                      ptr(_as_float_storage(gw)) if gw is not None else None, B, Ci, Co, n, K, act_in,
                                      ws.data_ptr(), nws, stream_ptr()), "spectral1d_bwd")
        return gx, gw, None


class _Spectral2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, act_in: int):
        lib = load()
        x = _f32c(x)
        B, Ci, M, N = x.shape
        Co, m1, m2 = w1.shape[1], w1.shape[2], w1.shape[3]
        out = torch.empty(B, Co, M, N, dtype=torch.float32, device=x.device)
        spec = torch.empty(lib.rpde_spectral2d_spec_elems(B, Ci, M, N, m1, m2), dtype=torch.float32, device=x.device)
        nws = lib.rpde_spectral2d_ws_bytes(B, Ci, Co, M, N, m1, m2)
        ws = workspace(nws, x.device)
        check(lib.rpde_spectral2d_fwd(ptr(x), ptr(_as_float_storage(w1)), ptr(_as_float_storage(w2)), ptr(out), ptr(spec),
                                      B, Ci, Co, M, N, m1, m2, act_in, ws.data_ptr(), nws, stream_ptr()), "spectral2d_fwd")
        ctx.save_for_backward(spec, w1, w2, x if act_in else x.new_empty(0))
        ctx.dims = (B, Ci, Co, M, N, m1, m2, act_in)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        spec, w1, w2, x = ctx.saved_tensors
        B, Ci, Co, M, N, m1, m2, act_in = ctx.dims
        g = _f32c(g)
        gx = torch.empty(B, Ci, M, N, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gw1 = torch.empty_like(w1) if need_w else None
        gw2 = torch.empty_like(w2) if need_w else None
        nws = lib.rpde_spectral2d_ws_bytes(B, Ci, Co, M, N, m1, m2)
        ws = workspace(nws, g.device)
        check(lib.rpde_spectral2d_bwd(ptr(g), ptr(spec), ptr(_as_float_storage(w1)), ptr(_as_float_storage(w2)),
                                      ptr(x) if act_in else None, ptr(gx),
                                      ptr(_as_float_storage(gw1)) if need_w else None,
                                      ptr(_as_float_storage(gw2)) if need_w else None,
                                      B, Ci, Co, M, N, m1, m2, act_in, ws.data_ptr(), nws, stream_ptr()), "spectral2d_bwd")
        return gx, gw1, gw2, None


def spectral1d(x, w, act_in: str = "identity"):
    if w.shape[2] > x.shape[-1] // 2 + 1:
        raise RuntimeError(f"SpectralConv1d: modes1={w.shape[2]} exceeds n//2+1={x.shape[-1] // 2 + 1}")
    return _Spectral1d.apply(x, w, ACT[act_in])


def spectral2d(x, w1, w2, act_in: str = "identity"):
    if w1.shape[3] > x.shape[-1] // 2 + 1 or w1.shape[2] > x.shape[-2]:
        raise RuntimeError("SpectralConv2d: modes exceed the available spectrum")
    return _Spectral2d.apply(x, w1, w2, ACT[act_in])


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act_in: int, acc, acc_owned: bool = False):
        """out = (acc +) W . act(x) + b ; x [B,Cin,*S] channels-first, w [Cout,Cin,1(,1)]
        acc_owned: the caller hands over `acc` (a temporary it will not read again): accumulate in place even
        when autograd cannot tell (no_grad / eval, where every tensor is a leaf)"""
        lib = load()
        x = _f32c(x)
        B, Ci = x.shape[0], x.shape[1]
        S = x[0, 0].numel()
        Co = w.shape[0]
        w2 = _f32c(w).reshape(Co, Ci)
        b = _f32c(b) if b is not None else None
        if acc is not None:
            if acc.dtype == torch.float32 and acc.is_contiguous() and (not acc.is_leaf or (acc_owned and not acc.requires_grad)):
                ctx.mark_dirty(acc)          # accumulate in place into the spectral branch's output
                out = acc
            else:
                out = _f32c(acc).clone()
        else:
            out = torch.empty(B, Co, *x.shape[2:], dtype=torch.float32, device=x.device)
        check(lib.rpde_conv1x1_fwd(ptr(x), ptr(w2), ptr(b), ptr(out), B, Ci, Co, S, act_in, int(acc is not None),
                                   stream_ptr()), "conv1x1_fwd")
        ctx.save_for_backward(x, w2)
        ctx.meta = (B, Ci, Co, S, act_in, b is not None, acc is not None, w.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        x, w2 = ctx.saved_tensors
        B, Ci, Co, S, act_in, has_b, has_acc, wshape = ctx.meta
        g = _f32c(g)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w2) if ctx.needs_input_grad[1] else None
        gb = torch.empty(Co, dtype=torch.float32, device=g.device) if (has_b and ctx.needs_input_grad[2]) else None
        nws = lib.rpde_conv1x1_ws_bytes(B, Ci, Co, S)
        ws = workspace(nws, g.device)
        check(lib.rpde_conv1x1_bwd(ptr(x), ptr(w2), ptr(g), ptr(gx), ptr(gw), ptr(gb), B, Ci, Co, S, act_in, 0,
                                   ws.data_ptr(), nws, stream_ptr()), "conv1x1_bwd")
        return gx, (gw.reshape(wshape) if gw is not None else None), gb, None, (g if has_acc else None), None


def conv1x1(x, w, b=None, act_in: str = "identity", acc=None, acc_owned: bool = False):
    return _Conv1x1.apply(x, w, b, ACT[act_in], acc, acc_owned)


def conv1x1_act_eval(x, w, b, acc, act_out: str):
    """evaluation only (no autograd): acc <- act_out(acc + W . x + b), in place in the temporary `acc`"""
    lib = load()
    x = _f32c(x)
    B, Ci = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    Co = w.shape[0]
    w2 = _f32c(w.detach()).reshape(Co, Ci)
    bb = _f32c(b.detach()) if b is not None else None
    out = acc if (acc.dtype == torch.float32 and acc.is_contiguous()) else _f32c(acc).clone()
    check(lib.rpde_conv1x1_act_fwd(ptr(x), ptr(w2), ptr(bb), ptr(out), B, Ci, Co, S, 0, 1, ACT[act_out], stream_ptr()),
          "conv1x1_act_fwd")
    return out


def fnoblock2d_eval(x, w1, w2, wc, bc, act_out: str):
    """evaluation only (no autograd): act_out(SpectralConv2d(x; w1, w2) + conv1x1(x; wc, bc)) with the spectral branch's
    last transform, the bypass convolution and the activation in one pass over x; None when the shape is not covered"""
    lib = load()
    if x.dim() != 4 or w1.shape[3] > x.shape[-1] // 2 + 1 or w1.shape[2] > x.shape[-2]:
        return None
    B, Ci, M, N = x.shape
    Co, m1, m2 = w1.shape[1], w1.shape[2], w1.shape[3]
    if not lib.rpde_fnoblock2d_eval_ok(Ci, Co, M, N, m2):
        return None
    x = _f32c(x)
    wcf = _f32c(wc.detach()).reshape(Co, Ci)
    bcf = _f32c(bc.detach()) if bc is not None else None
    out = torch.empty(B, Co, M, N, dtype=torch.float32, device=x.device)
    nws = lib.rpde_fnoblock2d_eval_ws_bytes(B, Ci, Co, M, N, m1, m2)
    ws = workspace(nws, x.device)
    check(lib.rpde_fnoblock2d_eval_fwd(ptr(x), ptr(_as_float_storage(w1.detach())), ptr(_as_float_storage(w2.detach())), ptr(wcf),
                                       ptr(bcf), ptr(out), B, Ci, Co, M, N, m1, m2, ACT[act_out], ws.data_ptr(), nws,
                                       stream_ptr()), "fnoblock2d_eval_fwd")
    return out


def fnoblock2d_proj_eval(x, w1, w2, wc, bc, act_out: str, pw1, pb1, pw2, pb2):
    """evaluation only (no autograd): the LAST FNO block and the projection MLP in one library call --
    mlp2(gelu(mlp1(act_out(SpectralConv2d(x; w1, w2) + conv1x1(x; wc, bc))))) -- without the block's output ever being
    written (rpde_fnoblock2d_proj_eval_fwd); None when the shape is not covered"""
    lib = load()
    if x.dim() != 4 or not x.is_cuda or w1.shape[3] > x.shape[-1] // 2 + 1 or w1.shape[2] > x.shape[-2]:
        return None
    B, Ci, M, N = x.shape
    Co, m1, m2 = w1.shape[1], w1.shape[2], w1.shape[3]
    Cm, Cq = pw1.shape[0], pw2.shape[0]
    if pw1[0].numel() != Co or pw2[0].numel() != Cm or not lib.rpde_fnoblock2d_proj_eval_ok(Ci, Co, M, N, m1, m2, Cm, Cq):
        return None
    x = _f32c(x)
    wcf = _f32c(wc.detach()).reshape(Co, Ci)
    bcf = _f32c(bc.detach()) if bc is not None else None
    p1, p2 = _f32c(pw1.detach()).reshape(Cm, Co), _f32c(pw2.detach()).reshape(Cq, Cm)
    q1 = _f32c(pb1.detach()) if pb1 is not None else None
    q2 = _f32c(pb2.detach()) if pb2 is not None else None
    out = torch.empty(B, Cq, M, N, dtype=torch.float32, device=x.device)
    nws = lib.rpde_fnoblock2d_eval_ws_bytes(B, Ci, Co, M, N, m1, m2)
    ws = workspace(nws, x.device)
    check(lib.rpde_fnoblock2d_proj_eval_fwd(ptr(x), ptr(_as_float_storage(w1.detach())), ptr(_as_float_storage(w2.detach())),
                                            ptr(wcf), ptr(bcf), ptr(p1), ptr(q1), ptr(p2), ptr(q2), ptr(out), B, Ci, Co, M, N, m1,
                                            m2, ACT[act_out], Cm, Cq, ws.data_ptr(), nws, stream_ptr()),
          "fnoblock2d_proj_eval_fwd")
    return out


def fno2d_lift_block_eval(u, gx, gy, wl, bl, w1, w2, wc, bc, act_out: str):
    """evaluation only (no autograd): act_out(SpectralConv2d(x0) + conv1x1(x0)) with x0 = lifting(cat(u, gx, gy)) formed on
    the fly (rpde_fno2d_lift_block_eval_fwd: the lifted field is never written); u [B,1,M,N], gx [M], gy [N] device
    arrays.  None when the shape is not covered."""
    lib = load()
    if u.dim() != 4 or u.shape[1] != 1 or not u.is_cuda:
        return None
    B, _, M, N = u.shape
    C, Co, m1, m2 = w1.shape[0], w1.shape[1], w1.shape[2], w1.shape[3]
    if wl.shape[0] != C or wl[0].numel() != 3 or not lib.rpde_fno2d_lift_block_eval_ok(1, C, Co, M, N, m1, m2):
        return None
    u = _f32c(u)
    gx, gy = _f32c(gx), _f32c(gy)
    wlf = _f32c(wl.detach()).reshape(C, 3)
    blf = _f32c(bl.detach()) if bl is not None else None
    wcf = _f32c(wc.detach()).reshape(Co, C)
    bcf = _f32c(bc.detach()) if bc is not None else None
    out = torch.empty(B, Co, M, N, dtype=torch.float32, device=u.device)
    nws = lib.rpde_fno2d_lift_block_eval_ws_bytes(B, C, Co, M, N, m1, m2)
    ws = workspace(nws, u.device)
    check(lib.rpde_fno2d_lift_block_eval_fwd(ptr(u), ptr(gx), ptr(gy), ptr(wlf), ptr(blf), ptr(_as_float_storage(w1.detach())),
                                             ptr(_as_float_storage(w2.detach())), ptr(wcf), ptr(bcf), ptr(out), B, C, Co, M, N,
                                             m1, m2, ACT[act_out], ws.data_ptr(), nws, stream_ptr()), "fno2d_lift_block_eval_fwd")
    return out


def conv_mlp_eval(x, w1, b1, w2, b2, act_in: str = "identity"):
    """evaluation only (no autograd): mlp2(gelu(mlp1(act_in(x)))) of the FNO projection in one pass, or None when the
    shape is not covered (the caller then runs the two convolutions)"""
    lib = load()
    B, Ci = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    Cm, Co = w1.shape[0], w2.shape[0]
    if not lib.rpde_conv_mlp_ok(Ci, Cm, Co, S):
        return None
    x = _f32c(x)
    w1f, w2f = _f32c(w1.detach()).reshape(Cm, Ci), _f32c(w2.detach()).reshape(Co, Cm)
    b1f = _f32c(b1.detach()) if b1 is not None else None
    b2f = _f32c(b2.detach()) if b2 is not None else None
    out = torch.empty(B, Co, *x.shape[2:], dtype=torch.float32, device=x.device)
    check(lib.rpde_conv_mlp_fwd(ptr(x), ptr(w1f), ptr(b1f), ptr(w2f), ptr(b2f), ptr(out), B, Ci, Cm, Co, S, ACT[act_in],
                                stream_ptr()), "conv_mlp_fwd")
    return out


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act: int):
        lib = load()
        x = _f32c(x)
        out = torch.empty_like(x)
        check(lib.rpde_act_fwd(ptr(x), ptr(out), x.numel(), act, stream_ptr()), "act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        (x,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(x)
        check(lib.rpde_act_bwd(ptr(x), ptr(g), ptr(dx), x.numel(), ctx.act, stream_ptr()), "act_bwd")
        return dx, None


def activation(x, act: str):
    return x if act == "identity" else _Act.apply(x, ACT[act])


# ----------------------------------------------------------------------------
# model-boundary layout helpers
# ----------------------------------------------------------------------------
class _ConcatGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid_dims: int, lo: float, hi: float, channels_last: bool, gx, gy):
        lib = load()
        x = _f32c(x)
        B, Ci = x.shape[0], x.shape[1]
        sp = tuple(x.shape[2:])
        M, N = (sp[0], 1) if len(sp) == 1 else sp
        Ct = Ci + grid_dims
        shape = (B, *sp, Ct) if channels_last else (B, Ct, *sp)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        check(lib.rpde_concat_grid(ptr(x), ptr(out), B, Ci, M, N, grid_dims, lo, hi, int(channels_last),
                                   ptr(gx), ptr(gy), stream_ptr()), "concat_grid")
        ctx.meta = (Ci, channels_last, len(sp))
        return out

    @staticmethod
    def backward(ctx, g):
        Ci, channels_last, nd = ctx.meta
        if channels_last:
            gx = g[..., :Ci]
            gx = gx.permute(0, nd + 1, *range(1, nd + 1))
        else:
            gx = g[:, :Ci]
        return gx.contiguous(), None, None, None, None, None, None


def concat_grid(x, grid_dims: int, lo: float = 0.0, hi: float = 1.0, channels_last: bool = True, gridx=None, gridy=None):
    return _ConcatGrid.apply(x, grid_dims, float(lo), float(hi), bool(channels_last), gridx, gridy)


class _TransposeCS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_channels_first: bool):
        """to_channels_first: [B,*S,C] -> [B,C,*S];  else the inverse"""
        lib = load()
        x = _f32c(x)
        B = x.shape[0]
        if to_channels_first:
            sp, Cc = tuple(x.shape[1:-1]), x.shape[-1]
            out = torch.empty(B, Cc, *sp, dtype=torch.float32, device=x.device)
        else:
            Cc, sp = x.shape[1], tuple(x.shape[2:])
            out = torch.empty(B, *sp, Cc, dtype=torch.float32, device=x.device)
        S = 1
        for s in sp:
            S *= s
        check(lib.rpde_transpose_cs(ptr(x), ptr(out), B, S, Cc, int(to_channels_first), stream_ptr()), "transpose_cs")
        ctx.tcf = to_channels_first
        return out

    @staticmethod
    def backward(ctx, g):
        return _TransposeCS.apply(g, not ctx.tcf), None


def to_channels_first(x):
    if x.shape[-1] == 1:
        return x.reshape(x.shape[0], 1, *x.shape[1:-1])
    return _TransposeCS.apply(x, True)


def to_channels_last(x):
    if x.shape[1] == 1:
        return x.reshape(x.shape[0], *x.shape[2:], 1)
    return _TransposeCS.apply(x, False)


# ----------------------------------------------------------------------------
# relative L2 loss
# ----------------------------------------------------------------------------
class _RelL2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, size_average: bool, reduction: bool):
        lib = load()
        x = _f32c(x)
        y = _f32c(y)
        B = x.shape[0]
        per = x.numel() // B
        if y.numel() != x.numel():
            raise RuntimeError(f"RelativeL2Loss: shapes {tuple(x.shape)} and {tuple(y.shape)} differ in size")
        stats = torch.empty(lib.rpde_rel_l2_stats_elems(B), dtype=torch.float32, device=x.device)
        rel = torch.empty(B, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device) if reduction else None
        check(lib.rpde_rel_l2_fwd(ptr(x), ptr(y), ptr(rel), ptr(loss), ptr(stats), B, per, int(size_average),
                                  stream_ptr()), "rel_l2_fwd")
        ctx.save_for_backward(x, y, stats)
        ctx.meta = (B, per, size_average, reduction)
        return loss if reduction else rel

    @staticmethod
    def backward(ctx, g):
        lib = load()
        x, y, stats = ctx.saved_tensors
        B, per, size_average, reduction = ctx.meta
        g = _f32c(g)
        gx = torch.empty_like(x)
        check(lib.rpde_rel_l2_bwd(ptr(x), ptr(y), ptr(stats), ptr(g) if reduction else None,
                                  None if reduction else ptr(g), ptr(gx), B, per, int(size_average), stream_ptr()),
              "rel_l2_bwd")
        return gx, None, None, None


def relative_l2(x, y, size_average: bool = True, reduction: bool = True):
    return _RelL2.apply(x, y, bool(size_average), bool(reduction))


# ----------------------------------------------------------------------------
# spectral resize (evaluation-time data path; no autograd)
# ----------------------------------------------------------------------------
def resize1d(x: torch.Tensor, out_size: int) -> torch.Tensor:
    """x [..., n] -> [..., out_size]: rfft, keep the shared bins, irfft(out_size), times out/in"""
    lib = load()
    x = _f32c(x.detach())
    n = x.shape[-1]
    rows = x.numel() // n
    out = torch.empty(*x.shape[:-1], int(out_size), dtype=torch.float32, device=x.device)
    nws = lib.rpde_resize1d_ws_bytes(rows, n, int(out_size))
    ws = workspace(nws, x.device)
    check(lib.rpde_resize1d(ptr(x), ptr(out), rows, n, int(out_size), ws.data_ptr(), nws, stream_ptr()), "resize1d")
    return out


def resize2d(x: torch.Tensor, out_size) -> torch.Tensor:
    """x [..., M, N] -> [..., Mo, No] (reference utils/res_utils.py `resize`)"""
    lib = load()
    x = _f32c(x.detach())
    M, N = x.shape[-2], x.shape[-1]
    Mo, No = int(out_size[0]), int(out_size[1])
    rows = x.numel() // (M * N)
    out = torch.empty(*x.shape[:-2], Mo, No, dtype=torch.float32, device=x.device)
    nws = lib.rpde_resize2d_ws_bytes(rows, M, N, Mo, No)
    ws = workspace(nws, x.device)
    check(lib.rpde_resize2d(ptr(x), ptr(out), rows, M, N, Mo, No, ws.data_ptr(), nws, stream_ptr()), "resize2d")
    return out


def warm_plans(model, resolutions, dims: int, in_channels: int = 1, device="cuda") -> None:
    """Build every DFT plan (tables, adjoint tables, operand images: hipMalloc + one stream sync each,
    csrc/core.hip get_plan) and size the workspaces the model needs at the given grid resolutions, with one
    throw-away forward per resolution -- so that no allocation or synchronisation happens inside a training
    step, and hipGraph capture never meets a first-use plan.  Leaves parameters and the train/eval flag alone."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for r in sorted({int(r) for r in resolutions}):
            model(torch.zeros((1, in_channels) + (r,) * dims, device=device))
    model.train(was_training)
    torch.cuda.synchronize(device)
