"""CPU: the data layer against vectors produced by the imported reference
(tests/golden/make_golden_data.py -> tests/golden/data_layer.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_data import NS_CASES, synthetic_u  # noqa: E402  (data only: case list + seeded generator)


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "data_layer.npz")))


@pytest.fixture(scope="module")
def mat_dir(tmp_path_factory):
    from scipy.io import savemat
    d = tmp_path_factory.mktemp("ns")
    savemat(os.path.join(d, "ns_16_synth.mat"), {"u": synthetic_u()})
    np.savez(os.path.join(d, "ns_16_synth.npz"), u=synthetic_u())                               # [N,H,W,T]: heuristic path
    np.save(os.path.join(d, "ns_16_time_major.npy"), np.transpose(synthetic_u(), (0, 3, 1, 2)))  # [N,T,H,W]
    return str(d)


@pytest.mark.parametrize("name", list(NS_CASES))
@pytest.mark.parametrize("fname", ["ns_16_synth.mat", "ns_16_synth.npz", "ns_16_time_major.npy"])
def test_ns_markov_dataset_matches_reference(golden, mat_dir, name, fname):
    from dataloaders.ns_naive_markov import extract_ns_test_trajectories_for_rollout_single, ns_markov_dataset
    kw = NS_CASES[name]
    train, val, test, xn, yn = ns_markov_dataset(fname, mat_dir, **kw)
    rk = {k: v for k, v in kw.items() if k in ("reduced_batch", "reduced_resolution", "reduced_resolution_t",
                                               "use_low_pass_filter", "lowpass_cutoff_ratio", "num_samples_max")}
    trajs, info = extract_ns_test_trajectories_for_rollout_single(fname, mat_dir, **rk)
    assert [len(train), len(val), len(test), len(trajs)] == golden[f"{name}/sizes"].tolist()
    tol = dict(rtol=2e-6, atol=2e-6)
    for split, ds in (("train", train), ("val", val), ("test", test)):
        for tag, idx in (("first", 0), ("last", len(ds) - 1)):
            x, y = ds[idx]
            np.testing.assert_allclose(np.asarray(x), golden[f"{name}/{split}_{tag}_x"], **tol)
            np.testing.assert_allclose(np.asarray(y), golden[f"{name}/{split}_{tag}_y"], **tol)
    if kw.get("data_normalizer", True):
        for tag, nrm in (("x", xn), ("y", yn)):
            np.testing.assert_allclose(np.asarray(nrm.mean, dtype=np.float32), golden[f"{name}/{tag}_mean"], **tol)
            np.testing.assert_allclose(np.asarray(nrm.std, dtype=np.float32), golden[f"{name}/{tag}_std"], **tol)
        if f"{name}/decode_of_encode" in golden:
            probe = torch.from_numpy(synthetic_u(seed=5, n=1)[0, :, :, :1].transpose(2, 0, 1).copy())
            np.testing.assert_allclose(np.asarray(yn.decode(xn.encode(probe), device="cpu")),
                                       golden[f"{name}/decode_of_encode"], rtol=1e-5, atol=1e-5)
    else:
        assert xn is None and yn is None
    np.testing.assert_allclose(np.asarray(trajs[0]), golden[f"{name}/traj0"], **tol)
    np.testing.assert_allclose(np.asarray(trajs[-1]), golden[f"{name}/traj_last"], **tol)
    assert info[0] == {"original_index": 0, "source": "single_resolution_file"}


def test_lowpass_filters_match_reference(golden):
    from utils.low_pass_filter import lowpass_filter_1d, lowpass_filter_2d
    for c in (0.25, 0.5, 1.0):
        for key, fn, src in ((f"1d_{c}", lowpass_filter_1d, "in1"), (f"1db_{c}", lowpass_filter_1d, "in1b"),
                             (f"2d_{c}", lowpass_filter_2d, "in2"), (f"2db_{c}", lowpass_filter_2d, "in2b")):
            got = fn(torch.from_numpy(golden[f"lp/{src}"].copy()), cutoff_ratio=c).numpy()
            np.testing.assert_allclose(got, golden[f"lp/{key}"], rtol=1e-5, atol=1e-6)
            assert got.shape == golden[f"lp/{src}"].shape          # filters keep the grid


def test_layout_heuristic_and_errors(tmp_path):
    from dataloaders.ns_naive_markov import NSMarkovDataset, _to_time_major
    a = np.zeros((3, 64, 64, 10), np.float32)
    assert _to_time_major(a).shape == (3, 10, 64, 64)              # short last axis -> time
    b = np.zeros((3, 10, 64, 64), np.float32)
    assert _to_time_major(b).shape == (3, 10, 64, 64)              # already time-major
    c = np.zeros((3, 200, 8, 8), np.float32)
    assert _to_time_major(c).shape == (3, 200, 8, 8)
    with pytest.raises(FileNotFoundError):
        NSMarkovDataset("missing.mat", str(tmp_path))
    np.save(tmp_path / "bad.npy", np.zeros((4, 4), np.float32))
    with pytest.raises(ValueError, match="4D"):
        NSMarkovDataset("bad.npy", str(tmp_path))
    (tmp_path / "x.txt").write_text("")
    with pytest.raises(ValueError, match="Unsupported file extension"):
        NSMarkovDataset("x.txt", str(tmp_path))
    (tmp_path / "x.h5").write_bytes(b"")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            NSMarkovDataset("x.h5", str(tmp_path))
