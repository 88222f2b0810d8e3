"""pytest configuration: `gpu` marker, import paths, fixture loading."""
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, "resolution-pde_amd")
for p in (REPO, DROPIN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_fixture(name):
    """-> (case dict, spec dict, {result-name: digest dict})"""
    z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    digests = {}
    for key in z.files:
        if key == "meta":
            continue
        res, field = key.split("|")
        digests.setdefault(res, {})[field] = z[key]
    case = meta["case"]
    case["x"] = tuple(case["x"])
    spec = {k: (tuple(v[0]), v[1]) for k, v in meta["spec"].items()}
    return case, spec, digests


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return "cuda:0"
