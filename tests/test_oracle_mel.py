"""CPU tests pinning the feature ORACLE (oracle/mel_oracle.py): analytic known answers,
float32-vs-float64 agreement, filterbank structure and the committed golden slices.
The reference ships no tests for this path, so these are the pins."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import mel_oracle as mo

CASES = [(800, 80), (800, 128), (1600, 80), (1600, 128), (1024, 80), (1024, 128)]


def _power_spec_f64(x, n_fft):
    """|X|^2 (C, T, n_freq) via the oracle's framing with an identity 'filterbank'."""
    n_freq = n_fft // 2 + 1
    return np.transpose(mo.mel_power_f64(x, n_fft, n_freq, fb=np.eye(n_freq)), (0, 2, 1))


@pytest.mark.parametrize("n_fft,F", CASES)
def test_zeros_give_minus_100_db(n_fft, F):
    x = torch.zeros(1, 8000)
    assert torch.all(mo.mel_spectrogram_torch(x, n_fft, F) == -100.0)
    assert np.all(mo.mel_spectrogram_f64(x.numpy(), n_fft, F) == -100.0)


@pytest.mark.parametrize("n_fft", [800, 1600, 1024])
def test_bin_centre_sinusoid(n_fft):
    # unit sinusoid at bin k: centre frame |X[k]|^2 = (n_fft/4)^2, |X[k+-1]|^2 = (n_fft/8)^2
    k = 37
    n = np.arange(16000)
    x = np.cos(2 * np.pi * k * n / n_fft)[None, :]
    P = _power_spec_f64(x, n_fft)
    t = P.shape[1] // 2
    assert P[0, t, k] == pytest.approx((n_fft / 4) ** 2, rel=1e-9)
    assert P[0, t, k - 1] == pytest.approx((n_fft / 8) ** 2, rel=1e-9)
    assert P[0, t, k + 1] == pytest.approx((n_fft / 8) ** 2, rel=1e-9)
    assert P[0, t, k + 5] < 1e-12 * (n_fft / 4) ** 2


@pytest.mark.parametrize("n_fft", [800, 1600, 1024])
def test_constant_and_impulse(n_fft):
    x = np.ones((1, 16000))
    P = _power_spec_f64(x, n_fft)
    t = P.shape[1] // 2
    assert P[0, t, 0] == pytest.approx((n_fft / 2) ** 2, rel=1e-9)
    assert P[0, t, 1] == pytest.approx((n_fft / 4) ** 2, rel=1e-9)
    # unit impulse at the centre of frame t: flat spectrum w[n_fft/2]^2 = 1
    x = np.zeros((1, 16000))
    x[0, 160 * 40] = 1.0
    P = _power_spec_f64(x, n_fft)
    np.testing.assert_allclose(P[0, 40], 1.0, rtol=1e-9)


@pytest.mark.parametrize("n_fft,F", CASES)
def test_fp32_oracle_matches_f64(n_fft, F):
    torch.manual_seed(8)
    x = torch.randn(2, 16000) * 0.1
    a = mo.mel_spectrogram_torch(x, n_fft, F).numpy()
    b = mo.mel_spectrogram_f64(x.numpy(), n_fft, F)
    assert a.shape == (2, F, 101)
    pa, pb = 10 ** (a / 10), 10 ** (b / 10)
    assert np.max(np.abs(pa - pb) / pb) < 1e-4   # north-star tolerance: 1e-4 rel on mel power
    assert np.max(np.abs(a - b)) < 1e-3


@pytest.mark.parametrize("n_fft,F", CASES)
def test_filterbank_structure(n_fft, F, golden_dir):
    g = np.load(os.path.join(golden_dir, "mel_golden.npz"))
    fb = mo.melscale_fbanks_htk(n_fft // 2 + 1, F).numpy()
    key = f"n{n_fft}_f{F}"
    assert (fb > 0).sum() == g[key + "_fb_nnz"]
    assert ((fb > 0).sum(1) <= 2).all()          # each bin feeds at most two filters
    assert ((fb > 0).sum(0) > 0).all()           # no empty filter at the sizes in scope
    np.testing.assert_allclose(fb.sum(0), g[key + "_fb_colsum"], rtol=1e-6)
    for m in range(F):                           # each filter is one contiguous run
        nz = np.nonzero(fb[:, m])[0]
        assert nz[-1] - nz[0] + 1 == len(nz)
        assert nz[0] == g[key + "_fb_first"][m]


@pytest.mark.parametrize("n_fft,F", CASES)
def test_golden_slices(n_fft, F, golden_dir):
    g = np.load(os.path.join(golden_dir, "mel_golden.npz"))
    torch.manual_seed(8)
    x = torch.randn(4, 80000) * 0.1
    key = f"n{n_fft}_f{F}"
    frames = list(g["frames"])
    b = mo.mel_spectrogram_f64(x.numpy(), n_fft, F)
    np.testing.assert_allclose(b[:, :, frames], g[key + "_slices"], rtol=1e-9, atol=1e-9)
    assert b.sum() == pytest.approx(float(g[key + "_sum"]), rel=1e-9)
    a = mo.mel_spectrogram_torch(x, n_fft, F).numpy()
    assert a.shape == (4, F, 501)
    np.testing.assert_allclose(a[:, :, frames], g[key + "_slices"], rtol=1e-4, atol=1e-3)


def test_product_tables_match_oracle_tables():
    """The product builds its own window / filterbank (sept_amd.melscale); it must be the
    same table the oracle restates."""
    from sept_amd import melscale
    for n_fft, F in CASES + [(400, 128)]:
        assert torch.equal(melscale.melscale_fbanks_htk(n_fft // 2 + 1, F),
                           mo.melscale_fbanks_htk(n_fft // 2 + 1, F))
        assert torch.equal(melscale.hann_window(n_fft), torch.hann_window(n_fft))


def test_mfcc_oracle_identities():
    """MFCC restatement (audio_feature_extraction.py:15-26): the ortho DCT-II has orthonormal
    columns, coefficient 0 is sum(mel_db)/sqrt(n_mels), the top_db clamp holds per clip, and the
    120-row layout is [x, gradient(x), gradient(x, 2)]."""
    dct = mo.create_dct_ortho(40, 128).double().numpy()
    np.testing.assert_allclose(dct.T @ dct, np.eye(40), atol=5e-6)      # float32 table
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 8000)) * 0.05
    x[:, 4000:] = 0.0                                   # silence -> -100 dB before the clamp
    db = mo.mel_spectrogram_f64(x, 400, 128, hop=200)
    m = mo.mfcc_f64(x)
    assert m.shape == (1, 40, 41)
    clamped = np.maximum(db, db.max() - 80.0)
    assert clamped.min() >= db.max() - 80.0 - 1e-9 and db.min() < db.max() - 80.0
    np.testing.assert_allclose(m[:, 0], clamped.sum(axis=1) / np.sqrt(128.0), rtol=1e-6)   # float32 DCT table
    full = mo.mfcc_with_deltas_f64(x)
    assert full.shape == (1, 120, 41)
    np.testing.assert_allclose(full[:, :40], m)
    np.testing.assert_allclose(full[:, 80:], mo.mfcc_f64(np.gradient(x[0], 2)[None]))


def test_resample_oracle_properties():
    """Resample restatement (audio_feature_extraction.py:139-141): length rule, DC gain ~ 1, an in-band
    tone keeps its frequency and amplitude, a tone above the new Nyquist is removed."""
    sr, new = 44100, 16000
    n = 44100
    t = torch.arange(n, dtype=torch.float64) / sr
    y = mo.resample_torch(torch.ones(1, n), sr, new)
    assert y.shape == (1, 16000) and abs(float(y[0, 200:-200].mean()) - 1.0) < 2e-3
    tone = torch.sin(2 * math.pi * 1000.0 * t).float()[None]
    y = mo.resample_torch(tone, sr, new)[0, 400:-400]
    want = torch.sin(2 * math.pi * 1000.0 * torch.arange(16000, dtype=torch.float64) / new)[400:-400]
    assert float((y.double() - want).abs().max()) < 2e-2
    high = torch.sin(2 * math.pi * 12000.0 * t).float()[None]
    assert float(mo.resample_torch(high, sr, new)[0, 400:-400].abs().max()) < 2e-2
    assert mo.resample_torch(torch.ones(2, 3, 301), 48000, 16000).shape == (2, 3, 101)
