#!/usr/bin/env python3
"""Headline benchmark: training samples/s of FFNO2D on 256^2 Navier-Stokes-like
synthetic fields (BASELINE.json configs[2]), one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--batch B]

A step = forward + relative-L2 loss + backward + gradient all-reduce (N>1) +
AdamW on one batch of B samples per GPU, inputs resident in HBM.  With --gpus N
and no WORLD_SIZE in the environment the script spawns its N ranks itself
(rpde/launch.py); under torch.distributed.run it is one of them.  Rank 0 prints
ONE JSON line.  It also carries
  * roofline: the dominant kernel -- k_ff3_fwd_h2<1>, the fused FeedForward
    64->256->256->64 forward of one layer in training mode -- timed live with
    HIP events on the launch stream.  SURVEY 8(d) prices FeedForward against
    the MATRIX roof: flops = 2*P*(64*256 + 256*256 + 256*64) = 12.88 GFLOP
    per sample; every fp32 product is issued as three f16 MFMA products
    (csrc/h2.h), so achieved = 3 * flops / time against the 2.5 PFLOP/s dense
    f16 peak.  roofline_extra: the same kernel's HBM view (its saved-for-
    backward stash labelled as design traffic, not algorithmic bytes), the
    kernel in evaluation, the fused backward chain, the streaming
    weight-gradient kernel, the spectral backward and the BASELINE config-5
    SpectralConv2d forward;
  * roofline_spectral: the FSpectralConv2d.forward_fourier pipeline (fused h2
    analysis / synthesis kernels), algorithmic bytes (33.55 MB*B +
    1.31 MB per layer forward) against 8 TB/s;
  * roofline_step: all kernels of a step, PMC bytes (profiles/traffic.json)
    over this run's step time;
  * cpu_baseline: the CPU oracle's training step timed on the host cores
    (rank 0, at every N, bounded sample: batch 4, 2 warm-up + 5 timed steps,
    and one single-thread step);
  * gpu_aten_baseline: the SAME oracle (the reference's op sequence: permute,
    rfft, einsum, irfft ... on hipFFT/rocFFT + hipBLASLt through ATen) run on
    the GPU beside the HIP path -- the "why not rocFFT" A/B of DESIGN.md
    section 2 (checker code, never on the product path);
  * graphed_step: the same step captured once as a hipGraph and replayed (N = 1; N > 1 with RPDE_BENCH_GRAPH=1);
  * config4_mres: BASELINE configs[3] -- the same model on a {64,128,256}^2 stream through the real
    ResolutionGroupedDataLoader, pinned host batches, host-to-device copies inside the timed region (all ranks);
  * parity: the headline model at B = 8 (forward, loss, every gradient) against the digests the imported reference
    produced (tests/golden/ffno2d_cfg3_256_b8.npz).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG3 = dict(in_channels=1, out_channels=1, width=64, n_layers=4, n_modes=20, factor=4, ff_weight_norm=True,
            n_ff_layers=3, layer_norm=True, dropout=0.1)
RES = 256
PEAK_F32_MFMA_TF = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0         # HBM3E spec; 6290 measured-achievable
PEAK_BF16_MFMA_TF = 2500.0    # dense bf16 matrix peak; the split-bf16 fp32 GEMM spends 6 bf16 flops per fp32 flop


def synth_batch(b, res, seed, device):
    """periodic Gaussian random fields, spectrum (4 pi^2 |k|^2 + tau^2)^(-alpha/2), alpha=2.5, tau=7
    (data_generation/random_fields.py parameters), standardised to mean 0 / std 1; the target is the input
    advanced by a fixed linear spectral filter (utils/synthetic.py: diffusion + translation), so the
    relative-L2 training loss is meaningful and decreases"""
    from utils.synthetic import advance, random_fields
    x = random_fields(b, res, 2, seed)
    y = advance(x, 2)
    return [x.to(device), y.to(device)]


def _time_gemm(desc, iters):
    from rpde import _lib
    lib = _lib.load()
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(lib.rpde_gemm_f32(C.byref(desc), st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.check(lib.rpde_gemm_f32(C.byref(desc), st), "gemm")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _ev_time(fn, iters, warm=3):
    """average milliseconds per call, HIP events on the current (= launch) stream"""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_feedforward(B, device, iters=10):
    """the FeedForward(64 -> 256 -> 256 -> 64) of one FFNO2D layer on P = B*65536 points, through the C ABI:
      fwd_train  rpde_feedforward_fwd with the hidden buffers given: weight preparation (one small block) + the fused
                 kernel k_ff3_fwd_h2<true> that also stores h1, d1, h2, d2, z3 -- the DOMINANT kernel of the step
      fwd_eval   the same call without hidden buffers (k_ff3_fwd_h2<false>: nothing but the output is written)
      bwd_chain  rpde_feedforward_bwd with every gradient pointer NULL except nothing: preparation + k_ff3_bwd_h2
                 (LayerNorm / dropout adjoint and the data-gradient chain; writes dz3, du2, du1)
      wgrad      the 256 x 256 weight gradient (k_wgrad_h2: both operands streamed once, + the slab reduction)
    algorithmic bytes: every tensor that has to cross HBM once, 4 B per element"""
    from rpde import _lib
    lib = _lib.load()
    P, dim, hid = B * RES * RES, 64, 256
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(device)          # noqa: E731
    ws_ = [mk(hid, dim, s=0.12), mk(hid, hid, s=0.06), mk(dim, hid, s=0.06)]
    bs_ = [mk(hid, s=0.1), mk(hid, s=0.1), mk(dim, s=0.1)]
    gamma, beta = 1.0 + mk(dim, s=0.1), mk(dim, s=0.1)
    x, res, gout = mk(P, dim), mk(P, dim), mk(P, dim)
    hs = [torch.empty(P, hid, device=device) for _ in range(2)]
    ds = [torch.empty(P, hid, device=device) for _ in range(2)]
    z3, out = torch.empty(P, dim, device=device), torch.empty(P, dim, device=device)
    wa, ba = _lib.ptr_array(ws_), _lib.ptr_array(bs_)
    fp = _lib.FFParams(3, dim, 4, 1, 1e-5, 0.1, 12345, 0, C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p)),
                       gamma.data_ptr(), beta.data_ptr())
    fp_eval = _lib.FFParams(3, dim, 4, 1, 1e-5, 0.0, 0, 0, C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p)),
                            gamma.data_ptr(), beta.data_ptr())
    nfw = lib.rpde_feedforward_fwd_ws_bytes(dim, 4, 3)
    wsf = _lib.workspace(nfw, device)
    nbw = lib.rpde_feedforward_ws_bytes(P, dim, 4, 3)
    wsb = _lib.workspace(nbw, device)
    ha, da = _lib.ptr_array(hs), _lib.ptr_array(ds)
    none2 = _lib.ptr_array([None, None])
    PP = C.POINTER(C.c_void_p)
    st = _lib.stream_ptr()

    def fwd_train():
        _lib.check(lib.rpde_feedforward_fwd(C.byref(fp), x.data_ptr(), res.data_ptr(), C.cast(ha, PP), C.cast(da, PP), z3.data_ptr(),
                                            out.data_ptr(), P, wsf.data_ptr(), nfw, st), "ff fwd")

    def fwd_eval():
        _lib.check(lib.rpde_feedforward_fwd(C.byref(fp_eval), x.data_ptr(), res.data_ptr(), C.cast(none2, PP), C.cast(none2, PP),
                                            z3.data_ptr(), out.data_ptr(), P, wsf.data_ptr(), nfw, st), "ff fwd eval")

    def bwd_chain():
        _lib.check(lib.rpde_feedforward_bwd(C.byref(fp), x.data_ptr(), C.cast(ha, PP), C.cast(da, PP), z3.data_ptr(), gout.data_ptr(),
                                            None, None, None, None, None, P, wsb.data_ptr(), nbw, st), "ff bwd")

    # the complete backward as a training step runs it: chain + three weight gradients (+ dx) + bias / LayerNorm sums
    gx = torch.empty(P, dim, device=device)
    gws = [torch.empty_like(w) for w in ws_]
    gbs = [torch.empty_like(b) for b in bs_]
    ggam, gbet = torch.empty(dim, device=device), torch.empty(dim, device=device)
    gwa, gba = _lib.ptr_array(gws), _lib.ptr_array(gbs)

    def bwd_full():
        _lib.check(lib.rpde_feedforward_bwd(C.byref(fp), x.data_ptr(), C.cast(ha, PP), C.cast(da, PP), z3.data_ptr(), gout.data_ptr(),
                                            gx.data_ptr(), C.cast(gwa, PP), C.cast(gba, PP), ggam.data_ptr(), gbet.data_ptr(), P,
                                            wsb.data_ptr(), nbw, st), "ff bwd (all gradients)")

    assert lib.rpde_feedforward_is_fused(dim, 4, 3, P) == 1, "the fused FeedForward kernel does not cover the headline shape"
    t_ft = _ev_time(fwd_train, iters)
    t_bf = _ev_time(bwd_full, iters)
    t_fe = _ev_time(fwd_eval, iters)
    t_bc = _ev_time(bwd_chain, iters)
    # the 256 x 256 weight gradient, as rpde_feedforward_bwd launches it: k_wgrad_h2 + the fixed-order slab reduction
    gw = torch.empty(hid, hid, device=device)
    nlw = lib.rpde_linear_ws_bytes(P, hid, hid)
    wsl = _lib.workspace(nlw, device)

    def wgrad():
        _lib.check(lib.rpde_linear_bwd(hs[0].data_ptr(), ws_[1].data_ptr(), ds[1].data_ptr(), None, gw.data_ptr(), None, P, hid, hid,
                                       wsl.data_ptr(), nlw, st), "linear bwd (weight gradient)")
    t_wg = _ev_time(wgrad, iters)
    flops_fwd = 2.0 * P * (dim * hid + hid * hid + hid * dim)
    by_train = 4.0 * P * (4 * dim + 4 * hid)            # x, residual, out, z3 + h1, d1, h2, d2
    by_eval = 4.0 * P * 3 * dim                         # x, residual, out
    by_chain = 4.0 * P * (3 * dim + 4 * hid)            # g, z3, dz3 + d2, d1, du2, du1
    return {"fwd_train_ms": t_ft, "fwd_eval_ms": t_fe, "bwd_chain_ms": t_bc, "bwd_full_ms": t_bf, "wgrad_ms": t_wg, "flops_fwd": flops_fwd,
            "bytes_train": by_train, "bytes_eval": by_eval, "bytes_chain": by_chain, "flops_wgrad": 2.0 * P * hid * hid,
            "bytes_wgrad": 8.0 * P * hid}


def time_spectral(B, device, iters=10):
    """FSpectralConv2d.forward_fourier and its backward at [B,256,256,64], 20 modes (fused h2 path)"""
    from rpde import ops
    x = torch.randn(B, RES, RES, 64, device=device)
    wy = torch.randn(64, 64, 20, 2, device=device) * 0.1
    wx = torch.randn(64, 64, 20, 2, device=device) * 0.1
    with torch.no_grad():
        ms = _ev_time(lambda: ops.fspectral2d(x, wy, wx, 20), iters, warm=2)
    xg = x.clone().requires_grad_(True)
    wyg, wxg = wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    gg = torch.randn_like(x)

    def fb():
        ops.fspectral2d(xg, wyg, wxg, 20).backward(gg)
        xg.grad = None

    ms_fb = _ev_time(fb, iters, warm=2)
    alg_bytes = 4.0 * B * RES * RES * 2 * 64 + 2 * 8.0 * 64 * 64 * 20
    return ms, alg_bytes / (ms * 1e-3) / 1e9, alg_bytes, ms_fb - ms, ms_fb


def time_cfg5(device, B=8, iters=10):
    """BASELINE config 5: SpectralConv2d(32,32,12,12) forward at [B,32,512,512] against SURVEY 8(d)'s
    67.11 MB*B + 2.36 MB per layer, and the whole FNO2d(1,1,12,12,32) evaluation forward / rollout step"""
    from models.fno import FNO2d
    from rpde import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, 32, 512, 512, generator=g).to(device)
    w1 = (torch.rand(32, 32, 12, 12, 2, generator=g) / 1024).to(device)
    w2 = (torch.rand(32, 32, 12, 12, 2, generator=g) / 1024).to(device)
    w1, w2 = torch.view_as_complex(w1), torch.view_as_complex(w2)
    with torch.no_grad():
        ms = _ev_time(lambda: ops.spectral2d(x, w1, w2), iters)
        del x
        torch.manual_seed(0)
        model = FNO2d(1, 1, modes1=12, modes2=12, width=32).to(device).eval()
        u = torch.randn(B, 1, 512, 512, generator=g).to(device)
        ms_model = _ev_time(lambda: model(u), iters)
        u16 = torch.randn(16, 1, 512, 512, generator=g).to(device)
        ms_model16 = _ev_time(lambda: model(u16), iters)
    return ms, 67.11e6 * B + 2.36e6, ms_model, ms_model16


def time_shipped_yamls(B, device):
    """The reference's SHIPPED model yamls beside the BASELINE configurations (they differ: conf/model/ffno_2d/ffno_2d.yaml
    has n_modes 64, which is off the fused spectral path; conf/model/ffno_1d/ffno_1d.yaml is BASELINE configs[1]'s model):
      * one FSpectralConv2d.forward_fourier at K = 64, [B,256,256,64] (pack + per-mode GEMM + unpack around truncated-DFT
        GEMMs) and the FFNO2D(n_modes 64, dropout 0.1) training step at batch B;
      * FFNO1D(width 128, 64 modes, 3 FeedForward layers, LayerNorm, dropout 0.2) training step at [16,1,512], eager and as
        one hipGraph replay (the configuration is launch-bound)."""
    from models.ffno import FFNO1D, FFNO2D
    from rpde import ops
    from rpde.graph import GraphedTrainStep
    from rpde.optim import FlatAdamW
    from utils.loss import RelativeL2Loss
    K = 64
    g = torch.Generator(device="cpu").manual_seed(23)
    x = torch.randn(B, RES, RES, 64, generator=g).to(device)
    wy = (torch.randn(64, 64, K, 2, generator=g) * 0.1).to(device)
    wx = (torch.randn(64, 64, K, 2, generator=g) * 0.1).to(device)
    with torch.no_grad():
        spec_ms = _ev_time(lambda: ops.fspectral2d(x, wy, wx, K), 10)
    del x
    loss_fn = RelativeL2Loss(size_average=True)

    def stepper(model, xb, yb, opt):
        def step():
            opt.zero_grad()
            loss_fn(model(xb), yb).backward()
            opt.step()
        return step
    torch.manual_seed(0)
    m2 = FFNO2D(**dict(CFG3, n_modes=K)).to(device).train()
    xb = torch.randn(B, 1, RES, RES, generator=g).to(device)
    yb = torch.randn(B, 1, RES, RES, generator=g).to(device)
    step2_ms = _ev_time(stepper(m2, xb, yb, FlatAdamW(m2.parameters(), lr=1e-3)), 5, warm=2)
    del m2, xb, yb
    torch.manual_seed(0)
    m1 = FFNO1D(1, 1, width=128, n_layers=4, n_modes=64, factor=4, ff_weight_norm=True, n_ff_layers=3, layer_norm=True,
                dropout=0.2).to(device).train()
    x1 = torch.randn(16, 1, 512, generator=g).to(device)
    y1 = torch.randn(16, 1, 512, generator=g).to(device)
    opt1 = FlatAdamW(m1.parameters(), lr=1e-3, capturable=True)
    eager1_ms = _ev_time(stepper(m1, x1, y1, opt1), 30, warm=5)
    gstep = GraphedTrainStep(m1, loss_fn, opt1, x1, y1)
    graph1_ms = _ev_time(lambda: gstep(x1, y1), 50, warm=5)
    return {"spec_ms": spec_ms, "spec_bytes": 4.0 * B * RES * RES * 2 * 64 + 2 * 8.0 * 64 * 64 * K, "step2_ms": step2_ms,
            "ffno1d_eager_ms": eager1_ms, "ffno1d_graph_ms": graph1_ms}


def gpu_aten_baseline(B, device, hip_spec_ms, hip_spec_fb_ms, hip_ff_fwd_ms, hip_ff_fb_ms, hip_step_ms):
    """The reference's own op sequence on this GPU through stock ATen (hipFFT/rocFFT for rfft/irfft, hipBLASLt/rocBLAS
    for einsum/linear): oracle/reference_path.py on cuda tensors.  Baseline leg only (like cpu_baseline); the product
    path never touches it.  Times the spectral pipeline (models/spectral_convolution.py:256-318) and the FeedForward
    (models/custom_layer.py:49-68) of one FFNO2D layer at [B,256,256,64], forward and forward+backward, and the whole
    training step at batch B."""
    from models.ffno import FFNO2D
    from oracle import reference_path as R
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.randn(B, RES, RES, 64, generator=g).to(device)
    wy = (torch.randn(64, 64, 20, 2, generator=g) * 0.1).to(device)
    wx = (torch.randn(64, 64, 20, 2, generator=g) * 0.1).to(device)
    alg = 4.0 * B * RES * RES * 2 * 64 + 2 * 8.0 * 64 * 64 * 20
    out = {}
    with torch.no_grad():
        ms = _ev_time(lambda: R.fspectral2d_fourier(x, wy, wx, 20), 5, warm=2)
    xg, wyg, wxg = x.clone().requires_grad_(True), wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    cot = torch.randn_like(x)

    def fb():
        R.fspectral2d_fourier(xg, wyg, wxg, 20).backward(cot)
        xg.grad = wyg.grad = wxg.grad = None
    ms_fb = _ev_time(fb, 5, warm=2)
    out["spectral_fwd"] = {"aten_ms": round(ms, 3), "hip_ms": round(hip_spec_ms, 3), "aten_algorithmic_GBps": round(alg / ms / 1e6, 1),
                           "hip_algorithmic_GBps": round(alg / hip_spec_ms / 1e6, 1), "speedup": round(ms / hip_spec_ms, 2)}
    out["spectral_fwd_bwd"] = {"aten_ms": round(ms_fb, 3), "hip_ms": round(hip_spec_fb_ms, 3), "speedup": round(ms_fb / hip_spec_fb_ms, 2)}
    del xg, cot
    torch.manual_seed(3)
    model = FFNO2D(**CFG3).to(device)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    pfx = "fourier_layers.0.backcast_ff."
    x2 = x.reshape(-1, 64)
    with torch.no_grad():
        ms = _ev_time(lambda: R.feedforward(x2, sd, pfx, 3, True, 0.1, True), 5, warm=2)
    params = R.make_params({k: v for k, v in sd.items() if k.startswith(pfx)})
    x2g = x2.clone().requires_grad_(True)
    cot2 = torch.randn_like(x2)

    def ffb():
        R.feedforward(x2g, params, pfx, 3, True, 0.1, True).backward(cot2)
        x2g.grad = None
        for p_ in params.values():
            p_.grad = None
    ms_fb = _ev_time(ffb, 5, warm=2)
    out["feedforward_fwd_train"] = {"aten_ms": round(ms, 3), "hip_ms": round(hip_ff_fwd_ms, 3), "speedup": round(ms / hip_ff_fwd_ms, 2)}
    out["feedforward_fwd_bwd"] = {"aten_ms": round(ms_fb, 3), "hip_ms": round(hip_ff_fb_ms, 3), "speedup": round(ms_fb / hip_ff_fb_ms, 2)}
    del x2g, cot2, x, x2
    # the whole training step (fwd + rel-L2 + bwd + AdamW, dropout 0.1) of the oracle on the GPU
    params = R.make_params(sd)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3)
    xb, yb = synth_batch(B, RES, 99, device)
    fwd = lambda p, xx: R.ffno2d_forward(p, xx, CFG3["n_layers"], CFG3["n_modes"], CFG3["n_ff_layers"],  # noqa: E731
                                         CFG3["layer_norm"], CFG3["dropout"], training=True)

    def step():
        opt.zero_grad()
        R.relative_l2(fwd(params, xb), yb).backward()
        opt.step()
    ms = _ev_time(step, 3, warm=2)
    out["train_step"] = {"aten_ms": round(ms, 2), "hip_ms": round(hip_step_ms, 2), "aten_samples_per_s": round(B / ms * 1e3, 1),
                         "speedup": round(ms / hip_step_ms, 2)}
    out["what"] = ("oracle/reference_path.py (the reference's op sequence) on cuda tensors: ATen -> hipFFT/rocFFT, hipBLASLt/rocBLAS, "
                   f"ATen elementwise kernels; fp32, batch {B}, 256^2; hip = this repository's kernels, same shapes, same run")
    return out


def mres_leg(model, bucket, opt, loss_fn, B, world, rank, device, batches_per_res=4):
    """BASELINE configs[3]: the cfg3 model trained on a resolution-grouped stream -- one third of the batches each at 64^2,
    128^2 and 256^2, batch order shuffled with seed 0 -- THROUGH train/mres_training.ResolutionGroupedDataLoader
    (the reference's train/mres_training.py:87-166 with a seed and rank slices), from page-locked host memory with the
    host-to-device copies inside the timed region (non_blocking, one batch ahead on a side stream).  Every rank builds
    the same dataset and takes its slice of each global batch, so all ranks run the same resolution in the same step.
    Epochs 0 and 1 are the warm-up (DFT plans and allocator pools of the three grids), epochs 2 and 3 are timed."""
    from train.mres_training import ResolutionGroupedDataLoader, SimpleDataset
    from utils.synthetic import markov_pairs
    per_res = batches_per_res * B * world
    samples = markov_pairs({64: per_res, 128: per_res, 256: per_res}, 2, 4321)
    loader = ResolutionGroupedDataLoader(SimpleDataset(samples), B, shuffle=True, seed=0, rank=rank, world_size=world,
                                         verbose=False, pin_memory=True)
    copy_stream = torch.cuda.Stream(device=device)
    main_stream = torch.cuda.current_stream(device)

    def fetch(it):
        try:
            xb, yb = next(it)
        except StopIteration:
            return None
        with torch.cuda.stream(copy_stream):
            xd, yd = xb.to(device, non_blocking=True), yb.to(device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(copy_stream)
        return xd, yd, ready

    host = []

    def epoch(events):
        it = iter(loader)
        nxt = fetch(it)
        n = 0
        while nxt is not None:
            h0 = time.perf_counter()
            xd, yd, ready = nxt
            main_stream.wait_event(ready)
            xd.record_stream(main_stream)
            yd.record_stream(main_stream)
            bucket.zero()
            loss_fn(model(xd), yd).backward()
            bucket.all_reduce_mean()
            bucket.detach_untouched()
            opt.step()
            h1 = time.perf_counter()
            nxt = fetch(it)                          # host: stack the next batch + start its copies while the GPU runs this step
            if events is not None:
                host.append([int(xd.shape[-1]), round((h1 - h0) * 1e3, 2), round((time.perf_counter() - h1) * 1e3, 2)])
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append((int(xd.shape[-1]), e))
            n += 1
        return n

    # two warm-up epochs: the caching allocator still asks the driver for new segments in the second one (every epoch
    # draws another batch order; a segment request stalls the GPU for ~100 ms -- profiles/mres_host_probe.py)
    epoch(None)
    epoch(None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    events = []
    t0 = time.perf_counter()
    e0.record()
    steps = epoch(events) + epoch(events)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    by_res, prev = {}, e0
    for res, e in events:
        by_res.setdefault(res, []).append(prev.elapsed_time(e))
        prev = e
    return {"workload": "FFNO2D cfg3 on a ResolutionGroupedDataLoader stream {64,128,256}^2 in equal thirds (BASELINE configs[3]), "
                        "pinned host batches, H2D inside the timed region",
            "value": round(steps * B * world / elapsed, 2), "unit": "samples/s", "steps": steps, "batch_per_gpu": B,
            "batches_per_resolution": batches_per_res, "epochs": 2, "ms_per_epoch": round(elapsed * 1e3 / 2, 3),
            "ms_per_step_by_resolution": {str(r): round(sorted(v)[len(v) // 2], 3) for r, v in sorted(by_res.items())},
            "ms_all_steps_in_order": [[r, round(prev_e.elapsed_time(e), 3)] for (r, e), prev_e in
                                      zip(events, [e0] + [e for _, e in events[:-1]])],
            "host_ms_in_order": host, "host_ms_what": "[resolution, host time to launch the step, host time to draw + copy-start the next batch]"}


def graph_leg(model, bucket, opt, loss_fn, x, y, world, device, steps):
    """The same training step (incl. the gradient all-reduce) captured ONCE as a hipGraph and replayed
    (rpde/graph.py): what a rank's host has to do per step shrinks from ~250 kernel launches to one graph launch --
    the thing that matters when 8 ranks share one host's cores.  Dropout masks change per replay through the
    device-side counter (rpde_ff_params.seed_epoch)."""
    from rpde.graph import GraphedTrainStep
    try:
        gs = GraphedTrainStep(model, loss_fn, opt, x, y, warmup=1, after_backward=bucket.all_reduce_mean)
    except Exception as e:                                   # a leg, not the product: report and go on
        return {"error": repr(e)[:300]}
    for _ in range(3):
        gs(x, y)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs(x, y)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"ms_per_step": round(elapsed / steps * 1e3, 3), "samples_per_s": round(x.shape[0] * world * steps / elapsed, 2),
            "steps": steps, "what": "forward + rel-L2 + backward + all_reduce_mean + FlatAdamW(capturable) as ONE hipGraph replay per step"}


def host_cores() -> int:
    from rpde.launch import host_cores as _hc
    return _hc()


def _cpu_steps(batch, warm, timed, threads):
    from models.ffno import FFNO2D
    from oracle import reference_path as R
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in FFNO2D(**CFG3).state_dict().items()}
    torch.set_num_threads(threads)
    params = R.make_params(sd)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3)
    x, y = synth_batch(batch, RES, 99, "cpu")
    fwd = lambda p, xx: R.ffno2d_forward(p, xx, CFG3["n_layers"], CFG3["n_modes"], CFG3["n_ff_layers"],  # noqa: E731
                                         CFG3["layer_norm"], CFG3["dropout"], training=True)
    for _ in range(warm):
        R.train_step(fwd, params, opt, x, y)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        R.train_step(fwd, params, opt, x, y)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], torch.get_num_threads()


def cpu_baseline(batch=4, warm=2, timed=5):
    """the CPU oracle (pinned restatement of the reference) doing the same training step on the host cores
    (2 warm + 5 timed, median; plus one 1-thread step for a per-core figure)"""
    cores = host_cores()
    med, used = _cpu_steps(batch, warm, timed, cores)
    one, _ = _cpu_steps(1, 0, 1, 1)
    torch.set_num_threads(cores)
    return {"value": round(batch / med, 4), "unit": "samples/s", "cores": used, "kind": "port",
            "sample": f"median of {timed} timed (+{warm} warm-up) training steps (fwd + rel-L2 + bwd + AdamW, dropout 0.1) "
                      f"of the CPU oracle at batch {batch}, 256^2, fp32, {used} threads",
            "one_thread": {"value": round(1.0 / one, 4), "unit": "samples/s", "cores": 1,
                           "sample": "1 training step at batch 1, 1 thread, no warm-up"}}


def parity_check(device):
    """The headline model at B = 8, 256^2 -- forward, loss, input gradient and every parameter gradient -- against the
    digests the IMPORTED REFERENCE produced for the same numpy-seeded weights and inputs (tests/golden/
    ffno2d_cfg3_256_b8.npz, generated by tests/golden/make_golden.py in the build container; sampled scalars + norm +
    full-tensor projections).  The fixture is data; nothing of the reference or the oracle runs here."""
    import types
    from models.custom_layer import FeedForward, WNLinear                      # noqa: F401
    from models.ffno import FFNO1D, FFNO2D                                     # noqa: F401
    from utils.loss import RelativeL2Loss                                      # noqa: F401
    from tests.conftest import load_fixture
    from tests.golden import synth
    from tests.golden.runner import ModuleBackend, run_case
    ns = types.SimpleNamespace(**{k: v for k, v in locals().items() if k[0].isupper()})
    case, spec, digests = load_fixture("ffno2d_cfg3_256_b8")
    sd = synth.fill_state_dict(spec, case["seed"])
    res = run_case(case, ModuleBackend(ns, str(device)), sd)
    torch.cuda.synchronize()
    errs = {k: synth.compare(v, digests[k]) for k, v in res.items()}
    gmax = max(float(digests[k]["norm"]) for k in digests if k.startswith("grad/") or k == "dx")
    fwd = max(errs[k] for k in ("out", "loss"))
    # (gradients that vanish identically in the reference -- weight_v of one-column rows, SURVEY quirk Q1 -- are compared
    #  by magnitude in tests/golden/synth.check_results; here they are left out of the relative figure)
    grad = max(v for k, v in errs.items() if k not in ("out", "loss") and float(digests[k]["norm"]) > 1e-5 * gmax)
    return {"case": "ffno2d_cfg3_256_b8: FFNO2D cfg3, [8,1,256,256], forward + rel-L2 + backward, vs digests generated by the "
                    "imported reference", "fwd_rel_l2_vs_reference": fwd, "grad_rel_l2_vs_reference": grad,
            "tolerance": {"fwd": 1e-5, "grad": 2e-5}, "tensors_compared": len(errs)}


_T0 = time.time()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 10 warm-up + 50 timed
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aten-baseline", action="store_true", help="skip the stock-ATen-on-GPU A/B leg")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph-replayed step leg")
    ap.add_argument("--no-mres", action="store_true", help="skip the BASELINE configs[3] leg (mixed resolutions, grouped loader)")
    ap.add_argument("--steps-only", action="store_true",
                    help="profiling aid: stop after the timed training steps (no kernel microbenchmarks, no parity leg)")
    args = ap.parse_args()

    # --gpus N without a launcher: this (GPU-free) parent starts the N ranks itself and returns their exit code
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        from rpde.launch import spawn_ranks
        raise SystemExit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values", file=sys.stderr)
        raise SystemExit(2)
    if os.environ.get("RPDE_BENCH_DRYRUN") == "1":
        # launcher plumbing test (tests/test_bench_launcher_cpu.py): no GPU, no process group
        print(json.dumps({"dryrun": True, "rank": rank, "local_rank": local, "world": world,
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}), flush=True)
        raise SystemExit(int(os.environ.get("RPDE_BENCH_FAIL_CODE", "3"))
                         if os.environ.get("RPDE_BENCH_FAIL_RANK") == str(rank) else 0)
    # rehearsal aid: RPDE_DIST_BACKEND=gloo RPDE_SHARE_GPU=1 runs all ranks on cuda:0 (1-GPU box)
    backend = os.environ.get("RPDE_DIST_BACKEND", "nccl")
    if os.environ.get("RPDE_SHARE_GPU") == "1":
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    from rpde.launch import limit_host_threads
    limit_host_threads(world)                  # torch's intra-op pool: this rank's share of the cores it may really use

    from models.ffno import FFNO2D
    from rpde.parallel import FlatGradBucket
    from utils.loss import RelativeL2Loss

    torch.manual_seed(0)                       # identical initial weights on every rank
    model = FFNO2D(**CFG3).to(device).train()
    from rpde.optim import FlatAdamW
    bucket = FlatGradBucket(model.parameters())
    # torch.optim.AdamW's rule, one kernel per step; capturable: the step counter lives on the device, so the same
    # optimizer also serves the hipGraph leg below
    opt = FlatAdamW(model.parameters(), lr=1e-3, bucket=bucket, capturable=True)
    loss_fn = RelativeL2Loss(size_average=True)
    B = args.batch
    x, y = synth_batch(B, RES, 1234 + rank, device)
    torch.manual_seed(100 + rank)              # dropout seeds differ per rank
    losses = torch.zeros(args.warmup + args.steps, device=device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ar_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i, timed=-1):
        bucket.zero()
        loss = loss_fn(model(x), y)
        loss.backward()
        if world > 1 and timed >= 0:
            ar_ev[timed][0].record()
        bucket.all_reduce_mean()
        if world > 1 and timed >= 0:
            ar_ev[timed][1].record()
        bucket.detach_untouched()
        opt.step()
        losses[i].copy_(loss.detach())         # device-side bookkeeping, no per-step host sync

    log(f"model + data ready (B={B}/gpu, world={world})")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    from rpde.launch import freeze_setup_garbage
    freeze_setup_garbage()                     # no full garbage collection over the setup's objects inside a timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step(args.warmup + i, i)
        ev[i + 1].record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]       # noqa: E731
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_ev) / len(ar_ev) if world > 1 else 0.0
    lh = losses.cpu().tolist()
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps; rel-L2 {lh[0]:.4f} -> {lh[-1]:.4f}")
    # FSpectralConv2d.forward_fourier timed WHERE IT RUNS: four more training steps (every rank: the all-reduce is in the
    # step) with an event pair around each of the layer's forward calls.  The stand-alone loop of time_spectral() runs
    # the same four launches back to back, and every one of them slows by 10-20 % within its first 5 ms (the clock
    # settling under sustained matrix + memory load); inside a step they run at the step's clock.
    spec_in_step = ff_in_step = None
    if not args.steps_only:
        from rpde import ops as _ops
        pairs = {"fspectral2d": [], "feedforward": []}
        real = {k: getattr(_ops, k) for k in pairs}

        def timed_call(name):
            def call(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = real[name](*a, **k)
                e1.record()
                pairs[name].append((e0, e1))
                return r
            return call
        for name in pairs:
            setattr(_ops, name, timed_call(name))
        try:
            for _ in range(4):
                step(args.warmup + args.steps - 1)
        finally:
            for name in pairs:
                setattr(_ops, name, real[name])
        torch.cuda.synchronize()
        mean = lambda ps: sum(a.elapsed_time(b) for a, b in ps) / len(ps)      # noqa: E731
        spec_in_step, ff_in_step = mean(pairs["fspectral2d"]), mean(pairs["feedforward"])
        log(f"FeedForward forward (training) inside the training step: {ff_in_step:.4f} ms (mean of {len(pairs['feedforward'])} calls)")
        pairs = pairs["fspectral2d"]
        log(f"spectral forward inside the training step: {spec_in_step:.4f} ms (mean of {len(pairs)} calls)")
    graphed = None
    # (at N > 1 only on request: a capture that RCCL refused on one box would hang the other ranks, and N > 1 cannot be
    #  rehearsed on the 1-GPU boxes this is built on -- tests/test_gpu_rccl.py captures the collective in a world of one)
    if not args.steps_only and not args.no_graph and (world == 1 or os.environ.get("RPDE_BENCH_GRAPH") == "1"):
        graphed = graph_leg(model, bucket, opt, loss_fn, x, y, world, device, min(args.steps, 20))
        log(f"graphed step: {json.dumps(graphed)}")
    mres = None
    if not args.steps_only and not args.no_mres:
        mres = mres_leg(model, bucket, opt, loss_fn, B, world, rank, device)          # every rank takes part
        log(f"config 4 (mixed resolutions through the grouped loader): {json.dumps(mres)}")

    if rank == 0 and args.steps_only:
        print(json.dumps({"steps_only": True, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "samples_per_s": round(B * world * args.steps / elapsed, 3)}), flush=True)
    elif rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        ff = time_feedforward(B, device)
        log(f"FeedForward: fwd(train) {ff['fwd_train_ms']:.3f} ms, fwd(eval) {ff['fwd_eval_ms']:.3f} ms, "
            f"bwd chain {ff['bwd_chain_ms']:.3f} ms, wgrad {ff['wgrad_ms']:.3f} ms")
        s_ms, s_gbs, s_bytes, s_bwd_ms, s_fb_ms = time_spectral(B, device)
        log(f"spectral fwd {s_ms:.3f} ms = {s_gbs:.0f} GB/s algorithmic; bwd {s_bwd_ms:.3f} ms")
        c5_ms, c5_bytes, c5_model_ms, c5_model16_ms = time_cfg5(device)
        log(f"config 5: SpectralConv2d 512^2 forward {c5_ms:.3f} ms, FNO2d eval forward {c5_model_ms:.3f} ms (B=8), "
            f"{c5_model16_ms:.3f} ms (B=16)")
        sy = time_shipped_yamls(B, device)
        log(f"shipped yamls: FFNO2D n_modes 64: layer forward {sy['spec_ms']:.3f} ms, step {sy['step2_ms']:.3f} ms; FFNO1D step "
            f"{sy['ffno1d_eager_ms']:.3f} ms eager, {sy['ffno1d_graph_ms']:.3f} ms as a hipGraph")
        traffic = step_traffic = traffic_src = spec_traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                blob = json.load(open(tpath))
                traffic = blob.get("dominant_kernel", {}).get(f"B{B}")
                spec_traffic = blob.get("spectral_forward", {}).get(f"B{B}")
                step_traffic = blob.get("train_step", {}).get(f"B{B}", {}).get("hbm_bytes_per_step")
                traffic_src = blob.get("source")
            except Exception:
                traffic = step_traffic = spec_traffic = None

        def hbm(kernel, byt, ms, **extra):
            gbs = byt / (ms * 1e-3) / 1e9
            d = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                 "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": byt, "ms_per_launch": round(ms, 4)}
            d.update(extra)
            return d

        tf = lambda fl, ms: round(fl / (ms * 1e-3) / 1e12, 2)                                  # noqa: E731
        ff_ms = ff_in_step if ff_in_step else ff["fwd_train_ms"]
        line = {
            "metric": "training samples/sec, FFNO2D NS 256^2", "value": round(value, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FFNO2D(1,1,width=64,n_layers=4,n_modes=20,factor=4,ff_weight_norm,n_ff_layers=3,"
                                   "layer_norm,dropout=0.1) train step on [B,1,256,256] Gaussian random fields "
                                   "(BASELINE configs[2])",
                       "batch_per_gpu": B, "global_batch": B * world, "grid": [RES, RES], "optimizer": "AdamW lr=1e-3 (rpde.optim.FlatAdamW: torch.optim.AdamW's rule, one kernel)",
                       "parallelism": f"dp{world}" if world > 1 else "single", "grad_bucket_bytes": bucket.nbytes,
                       "allreduce_ms_per_step": round(ar_ms, 4),
                       "step_ms_p10_p50_p90": [round(pct(0.1), 3), round(pct(0.5), 3), round(pct(0.9), 3)],
                       "train_rel_l2_first_last": [round(lh[0], 6), round(lh[-1], 6)],
                       "target": "input advanced by a fixed spectral filter (utils/synthetic.py:advance)"},
            # fp32 in / out / accumulate everywhere.  Products run on the f16 matrix pipe by two-piece splitting with
            # dynamic power-of-two scaling (3 MFMA terms, csrc/h2.h) in the fused kernels and on the bf16 pipe by
            # three-piece splitting (6 terms) in the remaining GEMMs.
            # FeedForward is priced against the matrix roof.  flops_per_launch = 2*P*(64*256 + 256*256 +
            # 256*64) = 12.88 GFLOP * B; the kernel issues every fp32 product as THREE f16 MFMA products (h2.h), so the
            # matrix pipe executes 3x that: achieved = issued flops / time, peak = 2.5 PFLOP/s dense f16
            "roofline": {"kernel": "k_ff3_fwd_h2<train>: fused FeedForward 64->256->256->64 forward of one layer (+ its weight "
                                   "preparation launch)",
                         "bound": "mfma", "achieved": tf(3.0 * ff["flops_fwd"], ff_ms), "peak": PEAK_BF16_MFMA_TF,
                         "unit": "TFLOP/s", "frac": round(tf(3.0 * ff["flops_fwd"], ff_ms) / PEAK_BF16_MFMA_TF, 4),
                         "flops_per_launch": ff["flops_fwd"], "issued_flops_per_launch": 3.0 * ff["flops_fwd"],
                         "issued_flops": "3x: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate",
                         "fp32_equiv_tflops": tf(ff["flops_fwd"], ff_ms),
                         "frac_of_fp32_mfma_peak": round(tf(ff["flops_fwd"], ff_ms) / PEAK_F32_MFMA_TF, 4),
                         "ms_per_launch": round(ff_ms, 4),
                         "timed": ("HIP event pairs around the layer's forward call inside 4 training steps (16 calls: preparation "
                                   "launch + the kernel), i.e. over the timed region's own launches" if ff_in_step
                                   else "stand-alone loop of 10 launches"),
                         "ms_per_launch_back_to_back_loop": round(ff["fwd_train_ms"], 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "note": "fp32 in / out / accumulate; hidden activations never leave the CU.  Neither roof is near: "
                                 "the kernel is bound by VALU issue (bias, dropout, GELU + GELU', f16 splitting, LayerNorm) -- "
                                 "SQ counters in profiles/"},
            "roofline_extra": [
                hbm("k_ff3_fwd_h2<train>, HBM view: reads x, residual; writes out, z3 (algorithmic: 4*P*2C + weights = "
                    f"{(4.0 * B * RES * RES * 2 * 64 + 4 * 98304) / 1e9:.2f} GB) plus the saved-for-backward stash h1, d1, h2, d2 "
                    "(DESIGN TRAFFIC, not algorithmic: 4 x [P,256] fp32)", ff["bytes_train"], ff["fwd_train_ms"],
                    algorithmic_bytes_8d=4.0 * B * RES * RES * 2 * 64 + 4 * 98304, design_stash_bytes=4.0 * B * RES * RES * 4 * 256,
                    traffic=traffic),
                hbm("k_ff3_fwd_h2<eval>: the same forward in evaluation (writes nothing but the output)", ff["bytes_eval"],
                    ff["fwd_eval_ms"], fp32_equiv_tflops=tf(ff["flops_fwd"], ff["fwd_eval_ms"])),
                hbm("k_ff3_bwd_h2: LayerNorm/dropout adjoint + data-gradient chain (reads g, z3, d2, d1; writes dz3, du2, du1)",
                    ff["bytes_chain"], ff["bwd_chain_ms"]),
                hbm("k_wgrad_h2 + reduce_slabs: weight gradient [256,P]x[P,256] (reads du2 and h1 once; 256 slabs of 256 KB)",
                    ff["bytes_wgrad"], ff["wgrad_ms"], fp32_equiv_tflops=tf(ff["flops_wgrad"], ff["wgrad_ms"])),
                hbm("FSpectralConv2d backward (adjoint analysis, mode mix^T + weight gradients, adjoint synthesis + skip)",
                    2 * s_bytes, s_bwd_ms),
                hbm("BASELINE config 5: SpectralConv2d(32,32,12,12) forward at [8,32,512,512] (k_cf_analysis_h2, row DFT, "
                    "mode mix, row DFT, k_cf_synthesis_h2), SURVEY 8(d): 67.11 MB*B + 2.36 MB", c5_bytes, c5_ms,
                    fno2d_512_eval_forward_ms_B8=round(c5_model_ms, 3),
                    fno2d_512_eval_samples_per_s=round(8 / c5_model_ms * 1e3, 1)),
                # the whole evaluation forward of BASELINE config 5 against ITS ideal traffic: per sample 4 blocks x 67.11 MB
                # (read x, write the activated block output) + lifting (write 33.55 MB) + projection (read 33.55 MB) + the
                # 1-channel input and output (2 x 1.05 MB) = 337.6 MB
                hbm("BASELINE config 5: FNO2d(1,1,12,12,32) evaluation forward at [16,1,512,512] (per block: k_cf_analysis_h2, two row "
                    "DFTs + mode mix on the small spectra, then ONE pass for inverse DFT + bypass conv + GELU; fused projection MLP)",
                    16 * 337.6e6, c5_model16_ms, samples_per_s=round(16 / c5_model16_ms * 1e3, 1)),
                hbm("the reference's SHIPPED ffno_2d.yaml (n_modes 64, off the fused spectral path): FSpectralConv2d.forward_fourier "
                    f"at [{B},256,256,64], K = 64 (truncated-DFT GEMMs + per-mode GEMM), SURVEY 8(d) bytes", sy["spec_bytes"], sy["spec_ms"],
                    ffno2d_n_modes64_train_step_ms=round(sy["step2_ms"], 3),
                    ffno2d_n_modes64_samples_per_s=round(B / sy["step2_ms"] * 1e3, 1)),
            ],
            "config2_ffno1d": {"workload": "BASELINE configs[1] with the reference's ffno_1d.yaml: FFNO1D(width 128, 4 layers, 64 modes, "
                               "3-layer FeedForward, LayerNorm, weight norm, dropout 0.2) training step at [16,1,512]",
                               "eager_ms_per_step": round(sy["ffno1d_eager_ms"], 3), "graphed_ms_per_step": round(sy["ffno1d_graph_ms"], 3),
                               "samples_per_s_graphed": round(16 / sy["ffno1d_graph_ms"] * 1e3, 1),
                               "note": "launch-bound: ~100 dispatches per step (profiles/r04_ffno1d_sequence_after.txt)"},
            "roofline_spectral": hbm("FSpectralConv2d.forward_fourier: k_mix_prep + k_dft_analysis_rr_h2 (both axes, the field read "
                                     "from HBM once, no cross-wave sums) + k_mix_h2 (mode mix of both axes, writes the synthesis "
                                     "operands) + k_dft_synthesis4_h2 (field written once, whole 128-byte lines)",
                                     s_bytes, spec_in_step if spec_in_step else s_ms, traffic=spec_traffic,
                                     timed="HIP event pairs around the layer's forward call inside 4 training steps (16 calls): the "
                                     "four launches at the clock the step runs at" if spec_in_step else "stand-alone loop",
                                     ms_per_launch_back_to_back_loop=round(s_ms, 4),
                                     frac_back_to_back_loop=round(s_bytes / (s_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                     traffic_over_algorithmic=(round(spec_traffic / s_bytes, 3) if spec_traffic else None),
                                     backward_ms=round(s_bwd_ms, 4), backward_frac=round(2 * s_bytes / (s_bwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                     note="HBM traffic of the four launches (PMC, profiles/): field 1x in + 1x out, spectra "
                                     "written and read once as fp32 and once as operand fragments; the second read of the field "
                                     "(the other axis) is served by the L2 of the XCD that read it first.  What bounds the two big "
                                     "kernels is the rate at which a CU's vector-memory pipeline moves bytes (~10 B/cycle from HBM, "
                                     "~23 from L2): every field byte is LOADED twice by the analysis, 2.25 fragment bytes per byte "
                                     "stored by the synthesis -- DESIGN.md section 4.1"),
        }
        if step_traffic:
            # whole training step against the HBM roof: PMC-measured bytes of one step / this run's step time
            gbs = step_traffic / (ms_step * 1e-3) / 1e9
            line["roofline_step"] = {"bound": "hbm", "traffic_bytes_per_step": step_traffic, "traffic_source": traffic_src,
                                     "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": round(gbs / PEAK_HBM_GBS, 4),
                                     "note": "all kernels of one step; PMC bytes from the committed profile, time from this run"}
        if graphed is not None:
            graphed["eager_ms_per_step"] = round(ms_step, 3)
            line["graphed_step"] = graphed
        if mres is not None:
            line["config4_mres"] = mres
        # (the other ranks wait at the final barrier meanwhile: their host threads sleep, so the CPU baseline has the
        #  box's cores to itself at every N)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            log(f"cpu baseline {line['cpu_baseline']['value']} samples/s")
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        if world == 1:
            line["parity"] = parity_check(device)
            log(f"parity {line['parity']}")
            if not args.no_aten_baseline:
                line["gpu_aten_baseline"] = gpu_aten_baseline(B, device, s_ms, s_fb_ms, ff["fwd_train_ms"],
                                                              ff["fwd_train_ms"] + ff["bwd_full_ms"], ms_step)
                log(f"stock ATen on this GPU: {json.dumps(line['gpu_aten_baseline'])}")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
