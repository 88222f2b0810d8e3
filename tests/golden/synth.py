"""Deterministic synthetic weights / inputs / digests for the golden fixtures.

Everything is drawn from ``numpy.random.RandomState`` (a frozen, version-stable
generator), so the fixture files only need to hold *digests* of large tensors:
any host (the build container, the GPU box) regenerates bit-identical inputs
and state_dicts from a case's seed.  Own code; no reference content.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence, Tuple

import numpy as np
import torch

FULL_LIMIT = 20000      # tensors up to this many scalars are stored whole
N_SAMPLE = 8192         # larger ones: this many pseudo-randomly chosen scalars


def _seed_for(seed: int, name: str) -> int:
    return (seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1)


def rand_tensor(shape: Sequence[int], seed: int, name: str = "x", scale: float = 1.0,
                dtype: torch.dtype = torch.float32) -> torch.Tensor:
    rs = np.random.RandomState(_seed_for(seed, name))
    if dtype == torch.complex64:
        a = rs.standard_normal(tuple(shape)) + 1j * rs.standard_normal(tuple(shape))
        return torch.from_numpy((scale * a).astype(np.complex64))
    a = rs.standard_normal(tuple(shape)) * scale
    return torch.from_numpy(a.astype(np.float32)).to(dtype)


def smooth_field(shape: Sequence[int], seed: int, name: str = "field") -> torch.Tensor:
    """Band-limited periodic random field over the trailing 1 or 2 dims of a
    channels-first tensor [B,C,n] / [B,C,M,N], standardised to mean 0, std 1."""
    rs = np.random.RandomState(_seed_for(seed, name))
    shape = tuple(shape)
    nd = len(shape) - 2
    white = rs.standard_normal(shape)
    if nd == 1:
        k = np.fft.rfftfreq(shape[-1], 1.0 / shape[-1])
        filt = (4 * np.pi ** 2 * k ** 2 + 49.0) ** (-1.25)
        f = np.fft.irfft(np.fft.rfft(white, axis=-1) * filt, n=shape[-1], axis=-1)
    else:
        kx = np.fft.fftfreq(shape[-2], 1.0 / shape[-2])[:, None]
        ky = np.fft.rfftfreq(shape[-1], 1.0 / shape[-1])[None, :]
        filt = (4 * np.pi ** 2 * (kx ** 2 + ky ** 2) + 49.0) ** (-1.25)
        f = np.fft.irfft2(np.fft.rfft2(white, axes=(-2, -1)) * filt, s=shape[-2:], axes=(-2, -1))
    f = (f - f.mean()) / f.std()
    return torch.from_numpy(f.astype(np.float32))


def fill_state_dict(spec: Mapping[str, Tuple[Tuple[int, ...], str]], seed: int) -> Dict[str, torch.Tensor]:
    """spec: name -> (shape, dtype-string).  Scales keep activations O(1)."""
    out: Dict[str, torch.Tensor] = {}
    for name in sorted(spec):
        shape, dt = spec[name]
        shape = tuple(int(s) for s in shape)
        dtype = getattr(torch, dt)
        leaf = name.rsplit(".", 1)[-1]
        if dtype == torch.complex64:                       # SpectralConv*.weights1/2
            t = rand_tensor(shape, seed, name, 0.7 / shape[0], dtype)
        elif "fourier_weight" in name:                     # [C,C,K,2]
            t = rand_tensor(shape, seed, name, 1.0 / np.sqrt(shape[0]))
        elif leaf == "weight_g":
            t = 1.0 + 0.25 * rand_tensor(shape, seed, name)
        elif leaf == "bias":
            t = 0.1 * rand_tensor(shape, seed, name)
        elif leaf == "weight" and len(shape) == 1:         # LayerNorm gamma
            t = 1.0 + 0.1 * rand_tensor(shape, seed, name)
        else:                                              # Linear / conv1x1 / weight_v
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = rand_tensor(shape, seed, name, 1.0 / np.sqrt(max(fan_in, 1)))
        out[name] = t
    return out


def spec_of(sd: Mapping[str, torch.Tensor]) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    return {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()}


# ----------------------------------------------------------------------------
# digests
# ----------------------------------------------------------------------------
def _flat64(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu()
    if t.is_complex():
        t = torch.view_as_real(t.resolve_conj().contiguous())
    return t.contiguous().reshape(-1).to(torch.float64).numpy()


def sample_index(numel: int) -> np.ndarray:
    return np.random.RandomState(numel % (2 ** 31 - 1)).randint(0, numel, N_SAMPLE)


N_PROJ = 4              # full-tensor checksums: projections onto seeded Gaussian vectors


def projections(f: np.ndarray) -> np.ndarray:
    """f . r_j for N_PROJ seeded standard-normal vectors r_j (float64).  An error e anywhere in the tensor moves a
    projection by about |e|_2, so -- unlike the sampled scalars -- no tile can be wrong unseen: the sampled scalars
    say WHERE the values agree, the projections that nothing else differs."""
    out = np.empty(N_PROJ)
    for j in range(N_PROJ):
        rs = np.random.RandomState((f.size * 7919 + 104729 * j + 13) % (2 ** 31 - 1))
        acc, step = 0.0, 1 << 22
        for lo in range(0, f.size, step):                       # (chunked: the generator's draws are sequential)
            acc += float(f[lo:lo + step] @ rs.standard_normal(min(step, f.size - lo)))
        out[j] = acc
    return out


def digest(t: torch.Tensor) -> Dict[str, np.ndarray]:
    f = _flat64(t)
    d = {"norm": np.array(np.sqrt((f * f).sum())), "sum": np.array(f.sum()),
         "numel": np.array(f.size)}
    if f.size > FULL_LIMIT:
        d["proj"] = projections(f)
    if f.size <= FULL_LIMIT:
        d["full"] = f.astype(np.float32)
    else:
        d["sample"] = f[sample_index(f.size)].astype(np.float32)
    return d


def compare(t: torch.Tensor, d: Mapping[str, np.ndarray]) -> float:
    """rel-L2 distance between tensor ``t`` and a stored digest (on the stored
    scalars), also checking the global norm.  Returns the larger of the two."""
    f = _flat64(t)
    assert f.size == int(d["numel"]), (f.size, int(d["numel"]))
    ref = d["full"].astype(np.float64) if "full" in d else d["sample"].astype(np.float64)
    got = f if "full" in d else f[sample_index(f.size)]
    den = np.sqrt((ref * ref).sum()) + 1e-30
    e_samp = float(np.sqrt(((got - ref) ** 2).sum()) / den)
    n_ref = float(d["norm"])
    e_norm = abs(float(np.sqrt((f * f).sum())) - n_ref) / (n_ref + 1e-30)
    e_proj = 0.0
    if "proj" in d:
        e_proj = float(np.sqrt(((projections(f) - d["proj"].astype(np.float64)) ** 2).mean())) / (n_ref + 1e-30)
    return max(e_samp, e_norm, e_proj)


def check_results(res: Mapping[str, torch.Tensor], digests: Mapping[str, Mapping[str, np.ndarray]],
                  tol: float, grad_tol: float = None, label: str = "") -> Dict[str, float]:
    """Assert every result matches its digest.  Gradients whose reference norm is
    below 1e-5 x the case's largest gradient norm are mathematically zero (e.g.
    d/dv of g*v/|v| for one-element rows) and only have to be
    equally negligible."""
    grad_tol = tol if grad_tol is None else grad_tol
    assert set(res) == set(digests), (sorted(res), sorted(digests))
    gkeys = [k for k in digests if k.startswith("grad/") or k == "dx"]
    gmax = max([float(digests[k]["norm"]) for k in gkeys], default=0.0)
    errs = {}
    for k, t in res.items():
        ref_norm = float(digests[k]["norm"])
        if k in gkeys and ref_norm < 1e-5 * gmax:
            got = float(np.sqrt((_flat64(t) ** 2).sum()))
            assert got < 1e-4 * gmax, f"{label}:{k} should vanish, norm {got:.3e} (scale {gmax:.3e})"
            errs[k] = 0.0
            continue
        e = compare(t, digests[k])
        lim = grad_tol if (k in gkeys or k.startswith("param/")) else tol
        if "floor" in digests[k]:      # the reference's own fp32-vs-fp64 disagreement (ReLU kinks)
            lim = max(lim, 3.0 * float(digests[k]["floor"]))
        assert e <= lim, f"{label}:{k} rel-L2 {e:.3e} > {lim:.1e}"
        errs[k] = e
    return errs
