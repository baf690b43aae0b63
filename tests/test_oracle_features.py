"""CPU: the oracle's feature functions against the golden vectors made from the imported reference."""
import os

import numpy as np
import pytest

from oracle import features as F

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'features.npz'))


def test_band_energy_matches_reference():
    # reference squares float32 scalars with glibc powf (not correctly rounded): <= 2 ulp
    np.testing.assert_allclose(F.compute_band_E(G['bandE_in']), G['bandE_out'], rtol=2.5e-7, atol=0)


def test_interp_band_gain_bit_exact():
    for t in range(G['gain_in'].shape[0]):
        assert np.array_equal(F.interp_band_gain(G['gain_in'][t]), G['gain_out'][t])
    assert np.array_equal(F.interp_band_gain_batch(G['gain_in']), G['gain_out'].T)


@pytest.mark.parametrize('k', ['a', 'b', 'c'])
def test_imcra_bit_exact(k):
    out = F.imcra_noise_psd(G['imcra_in_' + k])
    assert out.dtype == np.float32
    assert np.array_equal(out, G['imcra_out_' + k])


def test_noise_band_feature():
    b = F.compute_band_E(np.sqrt(G['imcra_out_c'].T)) ** (1 / 6)
    np.testing.assert_allclose(b, G['noise_band_c'], rtol=2.5e-7, atol=0)


def test_rms():
    assert F.rms(G['rms_in']) == pytest.approx(float(G['rms_out']), rel=1e-7)


def test_stft_framing_and_roundtrip():
    rs = np.random.RandomState(0)
    for L in (257, 1000, 33536, 34048):
        x = rs.randn(L).astype(np.float32) * 0.03
        X = F.stft(x)
        assert X.shape == (257, 1 + L // 256) and X.dtype == np.complex64
        y = F.istft(X)
        assert y.shape == (256 * (X.shape[1] - 1),) and y.dtype == np.float32
        n = min(L, y.shape[0])
        np.testing.assert_allclose(y[:n], x[:n], atol=2e-7)


def test_stft_frame_index_bit_exact():
    # frame t = padded[256 t : 256 t + 512]: an impulse at sample 1000 (padded 1256) lands in frames 3 and 4
    L = 4096
    x = np.zeros(L, np.float32)
    x[1000] = 1.0
    E = (np.abs(F.stft(x)) ** 2).sum(0)
    nz = np.nonzero(E > 1e-12)[0]
    assert list(nz) == [3, 4]


def test_toy_file_identity():
    # iSTFT length identity on the toy file (33536 = 256 * 131)
    assert 256 * (F.n_frames(33536) - 1) == 33536
