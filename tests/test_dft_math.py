"""CPU: the truncated real-DFT matrix formulation the HIP path uses
(oracle/dft_math.py restates it in float64) equals torch.fft as the reference
calls it, including Hermitian weights, odd n, Nyquist and the dropped Im(DC)."""
import numpy as np
import pytest
import torch

from oracle import dft_math as D
from oracle import reference_path as R


@pytest.mark.parametrize("n,k,norm", [(16, 5, "ortho"), (16, 9, "backward"), (33, 7, "ortho"),
                                      (33, 17, "backward"), (256, 20, "ortho"), (12, 7, "forward")])
def test_analysis_equals_rfft(n, k, norm):
    x = torch.randn(3, n, dtype=torch.float64)
    ref = torch.fft.rfft(x, norm=norm)[:, :k]
    got = torch.from_numpy(D.analysis(n, k, norm)) @ x.T          # [2k, 3]
    got = torch.complex(got[0::2], got[1::2]).T
    assert torch.allclose(got, ref, atol=1e-12)


@pytest.mark.parametrize("n,k,norm", [(16, 5, "ortho"), (16, 9, "backward"), (33, 7, "ortho"),
                                      (33, 17, "backward"), (256, 20, "ortho"), (12, 7, "forward")])
def test_synthesis_equals_irfft_with_nonhermitian_input(n, k, norm):
    spec = torch.randn(3, k, dtype=torch.complex128)             # Im(DC), Im(Nyquist) nonzero
    full = torch.zeros(3, n // 2 + 1, dtype=torch.complex128)
    full[:, :k] = spec
    ref = torch.fft.irfft(full, n=n, norm=norm)
    coef = torch.view_as_real(spec).reshape(3, 2 * k)            # interleaved
    got = coef @ torch.from_numpy(D.synthesis(n, k, norm)).T
    assert torch.allclose(got, ref, atol=1e-12)


def test_fspectral2d_as_matrices_equals_oracle():
    torch.manual_seed(0)
    b, m, n, c, k = 2, 12, 20, 4, 5
    x = torch.randn(b, m, n, c, dtype=torch.float64)
    wy = torch.randn(c, c, k, 2, dtype=torch.float64)
    wx = torch.randn(c, c, k, 2, dtype=torch.float64)
    ref = R.fspectral2d_fourier(x, wy, wx, k)
    fa_n, fs_n = map(torch.from_numpy, (D.analysis(n, k, "ortho"), D.synthesis(n, k, "ortho")))
    fa_m, fs_m = map(torch.from_numpy, (D.analysis(m, k, "ortho"), D.synthesis(m, k, "ortho")))

    def mix(a, w):          # a [..., 2k, c] interleaved -> same
        ac = torch.complex(a[..., 0::2, :], a[..., 1::2, :])
        wc = torch.view_as_complex(w.contiguous())               # [i,o,k]
        oc = torch.einsum("...ki,iok->...ko", ac, wc)
        out = torch.empty_like(a)
        out[..., 0::2, :], out[..., 1::2, :] = oc.real, oc.imag
        return out

    ay = torch.einsum("ky,bmyc->bmkc", fa_n, x)
    xy = torch.einsum("yk,bmkc->bmyc", fs_n, mix(ay, wy))
    ax = torch.einsum("km,bmnc->bnkc", fa_m, x)
    xx = torch.einsum("mk,bnkc->bmnc", fs_m, mix(ax, wx))
    assert torch.allclose(xx + xy, ref, atol=1e-11)


def test_spectral_conv2d_as_matrices_equals_oracle():
    torch.manual_seed(1)
    b, ci, co, m, n, m1, m2 = 2, 3, 4, 12, 20, 4, 11
    x = torch.randn(b, ci, m, n, dtype=torch.float64)
    w1 = torch.randn(ci, co, m1, m2, dtype=torch.complex128)
    w2 = torch.randn(ci, co, m1, m2, dtype=torch.complex128)
    ref = R.spectral_conv2d(x, w1, w2)
    rows = np.concatenate([np.arange(m1), np.arange(m - m1, m)])
    fa_n = torch.from_numpy(D.analysis(n, m2, "backward"))
    a1 = torch.einsum("ky,bcmy->bcmk", fa_n, x)
    a1 = torch.complex(a1[..., 0::2], a1[..., 1::2])                        # [b,c,m,m2]
    a2 = torch.einsum("rm,bcmk->bcrk", torch.from_numpy(D.complex_analysis(m, rows, "backward")), a1)
    o = torch.cat([torch.einsum("bixy,ioxy->boxy", a2[:, :, :m1], w1),
                   torch.einsum("bixy,ioxy->boxy", a2[:, :, m1:], w2)], dim=2)
    s1 = torch.einsum("mr,borK->bomK", torch.from_numpy(D.complex_synthesis(m, rows, "backward")), o)
    coef = torch.view_as_real(s1).reshape(b, co, m, 2 * m2)
    got = torch.einsum("yk,bomk->bomy", torch.from_numpy(D.synthesis(n, m2, "backward")), coef)
    assert torch.allclose(got, ref, atol=1e-11)
