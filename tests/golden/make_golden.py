#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the IMPORTED reference (build container only).

Run:  python tests/golden/make_golden.py [case-name ...]

The reference tree (/root/reference, read-only) is put on sys.path with an
in-process stub for its unused ``import h5py``; nothing
from it is copied: the fixtures hold only the parameter layout (names, shapes,
dtypes), digests of outputs / gradients (full tensors when small, 8192 sampled
scalars + norm + sum when large) for the numpy-seeded inputs and weights of
``synth.py``.  The GPU box never sees the reference; it reads these files.
"""
from __future__ import annotations

import json
import os
import sys
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _reference_namespace():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    from models.spectral_convolution import (SpectralConv1d, SpectralConv2d, FSpectralConv1d,
                                             FSpectralConv2d)
    from models.custom_layer import FeedForward, WNLinear
    from models.fno import FNO1d, FNO2d
    from models.fno_blocks import FNOBlock1d, FNOBlock2d, MLP1d, MLP2d
    from models.ffno import FFNO1D, FFNO2D
    from utils.loss import RelativeL2Loss
    from utils.res_utils import resize, resize_1d
    sys.path.remove(REF)
    # drop the reference's top-level package names so they cannot shadow ours
    ns = types.SimpleNamespace(**{k: v for k, v in locals().items() if k[0].isupper() or k.startswith("resize")})
    for name in [m for m in sys.modules if m.split(".")[0] in ("models", "utils")]:
        del sys.modules[name]
    return ns


def main(argv):
    ns = _reference_namespace()
    sys.path.insert(0, REPO)
    import torch
    from tests.golden import synth
    from tests.golden.cases import CASES
    from tests.golden.runner import ModuleBackend, run_case

    torch.set_num_threads(8)
    backend = ModuleBackend(ns, "cpu")

    class Float64Backend(ModuleBackend):
        """same reference modules with every parameter and input promoted to
        float64 / complex128 (``.double()`` leaves complex weights alone, quirk Q17)"""

        def make(self, case, sd):
            inst = super().make(case, sd)
            mod = inst.module.double()
            for p_ in mod.parameters():
                if p_.is_complex():
                    p_.data = p_.data.to(torch.complex128)
            inner = inst.call
            inst.call = lambda x: inner(x.double())
            inst.leaves = dict(mod.named_parameters())
            return inst
    want = set(argv)
    for case in CASES:
        if want and case["name"] not in want:
            continue
        t0 = time.time()
        spec = {} if case["kind"] in ("RelativeL2Loss", "Resize1d", "Resize2d") else backend.spec(case)
        sd = synth.fill_state_dict(spec, case["seed"])
        res = run_case(case, backend, sd)
        floors = {}
        if case.get("floor"):
            # the reference's own fp32-vs-fp64 disagreement per result tensor
            res64 = run_case(case, Float64Backend(ns, "cpu"), sd)
            for name, t in res.items():
                r64 = res64[name].to(torch.float64)
                floors[name] = float((t.to(torch.float64) - r64).norm() / (r64.norm() + 1e-300))
        blob = {"meta": np.array(json.dumps({"case": case, "spec": spec,
                                             "torch": torch.__version__}))}
        for name, t in res.items():
            for k, v in synth.digest(t).items():
                blob[f"{name}|{k}"] = v
            if name in floors:
                blob[f"{name}|floor"] = np.array(floors[name])
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{case['name']:28s} {len(res):3d} tensors  {os.path.getsize(path) / 1024:8.1f} KiB"
              f"  {time.time() - t0:6.1f}s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
