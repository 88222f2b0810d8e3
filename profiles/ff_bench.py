"""FeedForward(64, 4, n_layers=3, layer_norm) on P = B*256*256 points: forward (training: stores h, d; evaluation),
forward + backward, with the fused kernel and with the per-GEMM path (RPDE_FUSED_FF=0 in a second run).
    python profiles/ff_bench.py [B] [iters]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "resolution-pde_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from models.custom_layer import FeedForward  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
torch.manual_seed(0)
ff = FeedForward(64, 4, n_layers=3, layer_norm=True, dropout=float(os.environ.get("FF_DROPOUT", "0.1"))).to(dev)
x = torch.randn(B, 256, 256, 64, device=dev)
res = torch.randn(B, 256, 256, 64, device=dev)


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


P = B * 65536
flops = 2.0 * P * (64 * 256 + 256 * 256 + 256 * 64)
ff.eval()
with torch.no_grad():
    ms = timed(lambda: ff(x, residual=res))
print(f"fused={os.environ.get('RPDE_FUSED_FF', '1')} eval forward      {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF fp32-equivalent", flush=True)
ff.train()
xg = x.clone().requires_grad_(True)


def fwd_train():
    return ff(xg, residual=res)


ms_f = timed(fwd_train)
print(f"fused={os.environ.get('RPDE_FUSED_FF', '1')} training forward  {ms_f:8.3f} ms  {flops / ms_f / 1e9:7.1f} TF", flush=True)
g = torch.randn_like(x)


def fb():
    o = ff(xg, residual=res)
    o.backward(g)
    xg.grad = None
    for p_ in ff.parameters():
        p_.grad = None


ms_fb = timed(fb)
print(f"fused={os.environ.get('RPDE_FUSED_FF', '1')} forward+backward  {ms_fb:8.3f} ms  (backward ~{ms_fb - ms_f:.3f} ms)", flush=True)
