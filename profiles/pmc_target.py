#!/usr/bin/env python3
"""Target of the separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950,
MI355X_MICROARCH.md 'PMC slots'): the fused FeedForward kernels, the weight-gradient GEMM and the spectral
forward / backward a few times each, at B*65536 points."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "resolution-pde_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print("feedforward", bench.time_feedforward(B, dev, iters=3))
print("spectral", bench.time_spectral(B, dev, iters=2))
print("cfg5", bench.time_cfg5(dev, iters=2))
