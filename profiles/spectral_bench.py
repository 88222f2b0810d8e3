"""FSpectralConv2d.forward_fourier and its backward at the headline shape (B x 256 x 256 x 64, 20 modes):
time per call with HIP events on the launch stream, algorithmic GB/s (33.55 MB*B + 1.31 MB per
forward, twice that per backward), and the error of the HIP path against torch.fft in float64 on one sample.
    python profiles/spectral_bench.py [B] [iters]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from rpde import ops  # noqa: E402


def ref_forward(x, wy, wx, K):
    """float64 restatement of spectral_convolution.py:256-318"""
    x = x.double().permute(0, 3, 1, 2)          # b i m n
    B, C, M, N = x.shape
    fy = torch.fft.rfft(x, dim=-1, norm="ortho")
    oy = fy.new_zeros(B, C, M, N // 2 + 1)
    oy[..., :K] = torch.einsum("bixy,ioy->boxy", fy[..., :K], torch.view_as_complex(wy.double().contiguous()))
    xy = torch.fft.irfft(oy, n=N, dim=-1, norm="ortho")
    fx = torch.fft.rfft(x, dim=-2, norm="ortho")
    ox = fx.new_zeros(B, C, M // 2 + 1, N)
    ox[:, :, :K] = torch.einsum("bixy,iox->boxy", fx[:, :, :K], torch.view_as_complex(wx.double().contiguous()))
    xx = torch.fft.irfft(ox, n=M, dim=-2, norm="ortho")
    return (xx + xy).permute(0, 2, 3, 1)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    R, C, K = 256, 64, 20
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.randn(B, R, R, C, device=dev)
    wy = torch.randn(C, C, K, 2, device=dev) * 0.1
    wx = torch.randn(C, C, K, 2, device=dev) * 0.1
    # accuracy, forward and backward, one sample
    xs = x[:1].clone().requires_grad_(True)
    wys, wxs = wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    out = ops.fspectral2d(xs, wys, wxs, K)
    g = torch.randn_like(out)
    out.backward(g)
    xr = x[:1].detach().cpu().double().requires_grad_(True)
    wyr, wxr = wy.detach().cpu().double().requires_grad_(True), wx.detach().cpu().double().requires_grad_(True)
    ref = ref_forward(xr, wyr, wxr, K)
    ref.backward(g.cpu().double())
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).norm() / b.detach().norm())  # noqa: E731
    print(f"rel-L2 vs float64: out {rel(out, ref):.2e}  dx {rel(xs.grad, xr.grad):.2e}  dWy {rel(wys.grad, wyr.grad):.2e}  "
          f"dWx {rel(wxs.grad, wxr.grad):.2e}", flush=True)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    with torch.no_grad():
        ms_f = timed(lambda: ops.fspectral2d(x, wy, wx, K))
    alg = 4.0 * B * R * R * 2 * C + 2 * 8.0 * C * C * K
    print(f"forward  B={B}: {ms_f:.4f} ms  {alg / ms_f / 1e6:.0f} GB/s algorithmic  frac of 8 TB/s {alg / ms_f / 1e6 / 8000:.3f}", flush=True)
    xg = x.clone().requires_grad_(True)
    wyg, wxg = wy.clone().requires_grad_(True), wx.clone().requires_grad_(True)
    gg = torch.randn_like(x)

    def fb():
        o = ops.fspectral2d(xg, wyg, wxg, K)
        o.backward(gg)
        xg.grad = None

    ms_fb = timed(fb)
    print(f"fwd+bwd  B={B}: {ms_fb:.4f} ms  (backward ~{ms_fb - ms_f:.4f} ms, {2 * alg / (ms_fb - ms_f) / 1e6:.0f} GB/s algorithmic)",
          flush=True)


if __name__ == "__main__":
    main()
