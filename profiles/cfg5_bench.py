"""BASELINE config 5: FNO2d(1,1,12,12,32) at 512^2, evaluation forward and a 16-step rollout; plus SpectralConv2d
forward alone against SURVEY 8(d): 67.11 MB*B + 2.36 MB per layer.
    python profiles/cfg5_bench.py [B] [iters]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from models.fno import FNO2d  # noqa: E402
from rpde import ops  # noqa: E402
from utils.autoregressive_step import perform_rollout_2d  # noqa: E402


def ref_spectral2d(x, w1, w2, m1, m2):
    x = x.double()
    B, C, M, N = x.shape
    xf = torch.fft.rfft2(x)
    out = torch.zeros(B, w1.shape[1], M, N // 2 + 1, dtype=torch.complex128)
    out[:, :, :m1, :m2] = torch.einsum("bixy,ioxy->boxy", xf[:, :, :m1, :m2], w1.to(torch.complex128))
    out[:, :, -m1:, :m2] = torch.einsum("bixy,ioxy->boxy", xf[:, :, -m1:, :m2], w2.to(torch.complex128))
    return torch.fft.irfft2(out, s=(M, N))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.randn(B, 32, 512, 512, device=dev)
    w1 = (torch.rand(32, 32, 12, 12, dtype=torch.cfloat) / 1024).to(dev)
    w2 = (torch.rand(32, 32, 12, 12, dtype=torch.cfloat) / 1024).to(dev)

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
        out = ops.spectral2d(x[:1], w1, w2)
        ref = ref_spectral2d(x[:1].cpu(), w1.cpu(), w2.cpu(), 12, 12)
        print(f"SpectralConv2d 512^2 rel-L2 vs float64: {float((out.cpu().double() - ref).norm() / ref.norm()):.2e}", flush=True)
        ms = timed(lambda: ops.spectral2d(x, w1, w2))
        alg = 67.11e6 * B + 2.36e6
        print(f"SpectralConv2d(32,32,12,12) forward [B={B},32,512,512]: {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s algorithmic  "
              f"frac of 8 TB/s {alg / ms / 1e6 / 8000:.3f}", flush=True)
        model = FNO2d(1, 1, modes1=12, modes2=12, width=32).to(dev).eval()
        u = torch.randn(B, 1, 512, 512, device=dev)
        ms_f = timed(lambda: model(u))
        print(f"FNO2d(1,1,12,12,32) eval forward [B={B},1,512,512]: {ms_f:.3f} ms  {B / ms_f * 1e3:.0f} samples/s", flush=True)
        ms_r = timed(lambda: perform_rollout_2d(model, u[:, 0], 16, device=dev))
        print(f"16-step rollout: {ms_r:.3f} ms  ({ms_r / 16:.3f} ms per step)", flush=True)


if __name__ == "__main__":
    main()
