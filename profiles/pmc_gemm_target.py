#!/usr/bin/env python3
"""the dominant FeedForward GEMMs only (target of SQ counter passes)"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "profiles"))
import torch  # noqa: E402
import kernel_bench as kb  # noqa: E402
P = 16 * 65536
bias = torch.randn(256, device="cuda:0")
hout = torch.empty(P * 256, device="cuda:0")
kb.run("fwd NT 256->256 +bias", P, 256, 256, 1, 1, iters=3, bias=bias, bias_mode=1)
kb.run("fwd NT 256->256 -> h,d", P, 256, 256, 1, 1, iters=3, bias=bias, bias_mode=1, write_act=1, aux_out=hout, drop_p=0.1,
       drop_seed=7, drop_ld=256, drop_where=4)
kb.run("wgrad TN 256x256 split128 plain", 256, 256, P, 0, 0, iters=3, ksplit=128)
