"""GPU: HIP feature kernels (through the C ABI) against the oracle and the reference goldens."""
import os

import numpy as np
import pytest
from conftest import ab_env

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'features.npz'))


@pytest.fixture(scope='module')
def au():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from nele_gan_amd import audio_util
    return audio_util


def _batch(n, L, start=0):
    from nele_gan_amd import synth
    return synth.batch(n, L, start)


@pytest.mark.parametrize('L', [257, 1000, 33536, 64000])
def test_stft_and_band_vs_oracle(au, L):
    from oracle import features as F
    c, _ = _batch(3, L)
    spec, band = au.stft_band(torch.from_numpy(c).cuda())
    T = 1 + L // 256
    assert spec.shape == (3, T, 257) and band.shape == (3, T, 64)      # framing: bit-exact frame count
    spec = spec.cpu().numpy()
    band = band.cpu().numpy()
    for b in range(3):
        X = F.stft(c[b])                                               # [257, T]
        scale = np.abs(X).max()
        assert np.abs(spec[b].T - X).max() <= 2e-6 * scale             # float64 FFT rounded to complex64
        ref = F.compute_band_E(np.abs(X).T) ** (1 / 6)
        np.testing.assert_allclose(band[b], ref, rtol=1e-5)


def test_stft_impulse_lands_in_the_right_frames(au):
    x = np.zeros((1, 4096), np.float32)
    x[0, 1000] = 1.0
    spec, _ = au.stft_band(torch.from_numpy(x).cuda())
    E = (spec[0].abs() ** 2).sum(1).cpu().numpy()
    assert list(np.nonzero(E > 1e-12)[0]) == [3, 4]


def test_band_energy_from_reference_spectrum(au):
    # feed the golden IMCRA input spectrum through nele_imcra_band: PSD and noise band vs the reference
    for k in ('a', 'b', 'c'):
        Y = G['imcra_in_' + k]                                         # [257, T]
        spec = torch.from_numpy(np.ascontiguousarray(Y.T)).cuda().unsqueeze(0)
        psd, band = au.imcra_band(spec, want_psd=True)
        psd = psd[0].cpu().numpy().T
        ref = G['imcra_out_' + k]
        rel = np.abs(psd - ref) / np.abs(ref)
        assert rel.max() < 1e-5, "IMCRA PSD max rel err %g" % rel.max()
        # frames 0..14 (pure float32 recursion) are bit-exact
        assert np.array_equal(psd[:, :15], ref[:, :15])
        if k == 'c':
            np.testing.assert_allclose(band[0].cpu().numpy(), G['noise_band_c'], rtol=1e-5)


def test_imcra_batched_matches_oracle(au):
    from oracle import features as F
    _, v = _batch(4, 20000, start=10)
    spec, _ = au.stft_band(torch.from_numpy(v).cuda(), want_band=False)
    psd, band = au.imcra_band(spec, want_psd=True)
    for b in range(4):
        Xo = np.ascontiguousarray(spec[b].cpu().numpy().T)
        ref = F.imcra_noise_psd(Xo)
        np.testing.assert_allclose(psd[b].cpu().numpy().T, ref, rtol=1e-5)
        refb = F.compute_band_E(np.sqrt(ref.T)) ** (1 / 6)
        np.testing.assert_allclose(band[b].cpu().numpy(), refb, rtol=1e-5)


def test_gain_golden(au):
    # interp_band_gain golden: unit spectrum -> the iSTFT of sqrt(g) is checked through the oracle path
    from oracle import features as F
    a = G['gain_in']                                                   # [5, 64]
    T = a.shape[0]
    rs = np.random.RandomState(1)
    X = (rs.randn(257, T) + 1j * rs.randn(257, T)).astype(np.complex64)
    wav = au.gain_istft(torch.from_numpy(a).cuda().unsqueeze(0),
                        torch.from_numpy(np.ascontiguousarray(X.T)).cuda().unsqueeze(0))[0].cpu().numpy()
    ref = F.istft(np.sqrt(G['gain_out'].T) * X)                        # gains from the *reference* function
    assert wav.shape == ref.shape == (256 * (T - 1),)
    np.testing.assert_allclose(wav, ref, atol=2e-6 * np.abs(ref).max())


@pytest.mark.parametrize('L', [1000, 33536, 64000])
def test_gain_istft_vs_oracle_and_roundtrip(au, L):
    from oracle import features as F
    c, _ = _batch(2, L, start=3)
    x = torch.from_numpy(c).cuda()
    spec, band = au.stft_band(x)
    T = spec.shape[1]
    rs = np.random.RandomState(5)
    alpha = np.exp(0.5 * rs.randn(2, T, 64)).astype(np.float32)
    wav = au.gain_istft(torch.from_numpy(alpha).cuda(), spec).cpu().numpy()
    assert wav.shape == (2, 256 * (T - 1))                             # bit-exact length
    for b in range(2):
        X = spec[b].cpu().numpy().T
        ref = F.resyn(X, alpha[b])
        np.testing.assert_allclose(wav[b], ref, atol=3e-6 * np.abs(ref).max())
    # unit gains except the forced low/high bins: STFT -> iSTFT returns the input above ~100 Hz; with the
    # oracle the same property holds, so compare the two rather than the raw input
    ones = np.ones((2, T, 64), np.float32)
    w1 = au.gain_istft(torch.from_numpy(ones).cuda(), spec).cpu().numpy()
    r1 = F.resyn(spec[0].cpu().numpy().T, ones[0])
    np.testing.assert_allclose(w1[0], r1, atol=3e-6 * np.abs(r1).max())


def test_reference_shaped_wrappers_on_toy_file(au):
    import wave
    from oracle import features as F
    p = os.path.join(os.path.dirname(__file__), 'golden', 'toy', 'Train_Clean.wav')
    w = wave.open(p)
    x = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
    bandE, mag, phase = au.Sp_and_phase_Speech(x, 1 / 6)
    assert bandE.shape == (132, 64) and mag.shape == (257, 132) and phase.shape == (257, 132)
    rb, rm, rp = F.sp_and_phase_speech(x, 1 / 6)
    np.testing.assert_allclose(bandE.cpu().numpy(), rb, rtol=1e-5)
    np.testing.assert_allclose(mag.cpu().numpy(), rm, atol=2e-6 * rm.max())
    nb, _, _ = au.Sp_and_phase_Noise(x, 1 / 6)
    rnb, _, _ = F.sp_and_phase_noise(x, 1 / 6)
    np.testing.assert_allclose(nb.cpu().numpy(), rnb, rtol=2e-5)
    alpha = np.ones((132, 64), np.float32) * 1.5
    y = au.SP_to_wav(alpha, mag, phase).cpu().numpy()
    ry = F.sp_to_wav(alpha, rm, rp)
    assert y.shape == (33536,)
    np.testing.assert_allclose(y, ry, atol=5e-6 * np.abs(ry).max())


def test_wav_post_rms_and_pcm16(au):
    rs = np.random.RandomState(2)
    x = (0.1 * rs.randn(2, 5000)).astype(np.float32)
    t = torch.from_numpy(x.copy()).cuda()
    from nele_gan_amd._lib import call, ptr, stream
    from nele_gan_amd import _lib
    ws = torch.empty(int(_lib.lib.nele_wav_post_workspace_doubles(2, 5000)), dtype=torch.float64, device='cuda')
    call('nele_wav_post', ptr(t), 2, 5000, 0.03, 1, ptr(ws), stream())
    with pytest.raises(ValueError):                                      # the rms normalisation without its workspace is an argument error
        call('nele_wav_post', ptr(t), 2, 5000, 0.03, 1, None, stream())
    y = t.cpu().numpy()
    for b in range(2):
        ref = x[b] / np.sqrt(np.mean(x[b] ** 2)) * np.float32(0.03)
        ref = np.clip(np.rint(ref * 32767.0), -32768, 32767) / 32768.0
        assert np.abs(y[b] - ref).max() <= 1.0 / 32768 + 1e-9          # at most one LSB (rounding ties)
        assert np.mean(np.abs(y[b] - ref) > 1e-9) < 1e-3


def test_wav_post_rms_is_batch_and_padding_invariant(au):
    """The rms of an utterance is summed in chunk order from per-chunk partials: the same bits whether the utterance is scored alone,
    inside a batch, or inside a longer padded batch (inference.py:109 is per file)."""
    rs = np.random.RandomState(3)
    x = (0.05 * rs.randn(3, 128000)).astype(np.float32)
    full = au.gain_istft.__globals__['torch'].from_numpy(x.copy()).cuda()
    from nele_gan_amd import _lib
    from nele_gan_amd._lib import call, ptr, stream

    def post(t, frames=None):
        B, N = t.shape
        ws = torch.empty(int(_lib.lib.nele_wav_post_workspace_doubles(B, N)), dtype=torch.float64, device='cuda')
        call('nele_wav_post_var', ptr(t), ptr(frames), B, N, 0.03, 1, ptr(ws), stream())
        return t.cpu().numpy()

    yb = post(full.clone())
    for b in range(3):
        assert np.array_equal(post(full[b:b + 1].clone())[0], yb[b])
    # utterance 1 cut to 300 frames (76 544 samples) inside a padded batch == the same samples scored at their own length
    frames = torch.tensor([501, 300, 501], dtype=torch.int32, device='cuda')
    cut = full.clone()
    cut[1, 256 * 299:] = 0
    yp = post(cut.clone(), frames)
    alone = post(full[1:2, :256 * 299].clone())
    assert np.array_equal(yp[1, :256 * 299], alone[0]) and not yp[1, 256 * 299:].any()
    assert np.array_equal(yp[0], yb[0]) and np.array_equal(yp[2], yb[2])


def test_errors_surface_as_exceptions(au):
    with pytest.raises(ValueError):
        au.stft_band(torch.zeros(1, 100).cuda())


_STFT_AB_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import audio_util as au, synth
out = []
for L, B in ((257, 2), (1000, 3), (33536, 2), (64000, 5), (63871, 2), (128000, 2)):
    c, _ = synth.batch(B, L, start=21)
    x = torch.from_numpy(c).cuda()
    spec, band = au.stft_band(x)
    T = spec.shape[1]
    alpha = torch.from_numpy(np.exp(0.5 * np.random.RandomState(L).randn(B, T, 64)).astype(np.float32)).cuda()
    out += [torch.view_as_real(spec).cpu().numpy().ravel(), band.cpu().numpy().ravel(), au.gain_istft(alpha, spec).cpu().numpy().ravel(),
            au.ISTFT(spec[0].transpose(0, 1)).cpu().numpy().ravel() if L == 1000 else np.zeros(0, np.float32)]
# a padded batch of different lengths (frames behind a short row's end are zeros; the last pair of a row may hold one frame only)
c, _ = synth.batch(4, 40000, start=22)
lens = torch.tensor([40000, 25601, 511, 30000], dtype=torch.int32)
x = torch.from_numpy(c).cuda()
spec, band = au.stft_band(x, lengths=lens.cuda())
alpha = torch.ones(4, spec.shape[1], 64, device='cuda')
out += [torch.view_as_real(spec).cpu().numpy().ravel(), band.cpu().numpy().ravel(),
        au.gain_istft(alpha, spec, frames=au.frames_of(lens.cuda())).cpu().numpy().ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_wave_per_transform_stft_istft_are_bit_identical_to_the_workgroup_kernels(tmp_path):
    """STFT / iSTFT run one wave per 512-point transform with the data in registers (fft512_wave: three butterfly stages per LDS
    exchange); every butterfly is the same float64 operation on the same operands as in the workgroup-per-transform kernels
    (NELE_STFT_WAVE=0, read once per process), so spectra, band features and resynthesised signals must be equal bit for bit -
    single frames, odd frame counts, 4 s / 8 s utterances and a padded batch of different lengths."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('stft_wave_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _STFT_AB_CHILD, os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out], check=True,
                       env=ab_env(NELE_STFT_WAVE=flag), timeout=240)
        res.append(np.load(out))
    assert res[0].shape == res[1].shape and np.all(np.isfinite(res[0]))
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))


def test_band_power_law_is_the_nearest_float32_to_the_float64_power():
    """`compute_band_E(...) ** p_power` (audio_util.py:434, 451): the kernels' x ** float32(1/6) fast path (float32 hardware log2 / exp2 +
    one float64 Newton step, features.hip pow_f32) against numpy's float64 power of the SAME band energies (power = 1 returns them raw),
    rounded to float32: equal except where a tie falls the other way (1 ulp, a few in 10^7); general exponents take the float64 pow."""
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import synth
    c, v = synth.batch(16, 64000, start=40)
    x = torch.from_numpy(np.concatenate([c, v * 30.0, c * 1e-4])).cuda()                   # band energies over ~14 decades
    lens = torch.full((x.shape[0],), 64000, dtype=torch.int32)
    lens[3] = 40000                                                                          # zero frames behind a short row: 0 ** p = 0
    _, raw = au.stft_band(x, power=1.0, want_spec=False, lengths=lens)
    _, got = au.stft_band(x, power=1.0 / 6, want_spec=False, lengths=lens)
    e = raw.cpu().numpy()
    p32 = np.float32(1.0 / 6)
    want = np.power(e.astype(np.float64), np.float64(p32)).astype(np.float32)
    g = got.cpu().numpy()
    assert (e == 0).any() and (e > 0).sum() > 700000 and e[e > 0].max() / e[e > 0].min() > 1e12
    diff = g != want
    assert diff.mean() < 5e-6
    if diff.any():
        assert np.max(np.abs(g[diff].astype(np.float64) - want[diff]) / np.spacing(want[diff])) <= 1.0
    assert np.all(g[e == 0] == 0)
    # a general exponent: the float64 pow
    _, g3 = au.stft_band(x[:4], power=0.3, want_spec=False)
    _, r3 = au.stft_band(x[:4], power=1.0, want_spec=False)
    w3 = np.power(r3.cpu().numpy().astype(np.float64), np.float64(np.float32(0.3))).astype(np.float32)
    assert np.mean(g3.cpu().numpy() != w3) < 1e-4
    # the noise path's band feature goes through the same function (band_from_psd_kernel)
    spec, _ = au.stft_band(x[:8], want_band=False)
    psd, nb = au.imcra_band(spec, want_psd=True)
    _, nb1 = au.imcra_band(spec, power=1.0, want_psd=True)
    w = np.power(nb1.cpu().numpy().astype(np.float64), np.float64(p32)).astype(np.float32)
    assert np.mean(nb.cpu().numpy() != w) < 5e-6


@pytest.mark.parametrize('B,L,ragged', [(3, 40000, False), (5, 64000, True), (2, 128000, True), (1, 5000, False), (4, 3800, True)])
def test_two_kernel_imcra_is_bit_identical_to_the_one_kernel_form(B, L, ragged):
    """nele_imcra_band_ws (|Y|^2 at once, then the indicator and the prior + tracker as one thread per utterance and bin)
    against nele_imcra_band_var (everything in one serial kernel): the same PSD and band feature, bit for bit - frames 0..14 (float32
    lambda_D), the hand-over at frames 15 / 16, minima-store rotations, short rows of a padded batch."""
    import ctypes
    from nele_gan_amd import _lib
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import synth
    c, v = synth.batch(B, L, start=70)
    x = torch.from_numpy(v + 0.3 * c).cuda()
    T = 1 + L // 256
    lens = None
    if ragged:
        lens = torch.tensor([L - 777 * k for k in range(B)], dtype=torch.int32)
    spec, _ = au.stft_band(x, want_band=False, lengths=lens)
    frames = None if lens is None else (1 + lens // 256).to(torch.int32).cuda()
    out = []
    for two in (False, True):
        psd = torch.full((B, T, 257), -1.0, dtype=torch.float32, device='cuda')
        band = torch.full((B, T, 64), -1.0, dtype=torch.float32, device='cuda')
        if two:
            nb = int(_lib.lib.nele_imcra_workspace_bytes(B, T))
            assert nb >= B * T * 257 * 5
            ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device='cuda')
            _lib.call('nele_imcra_band_ws', _lib.ptr(spec), _lib.ptr(frames), B, T, 1.0 / 6, _lib.ptr(psd), _lib.ptr(band), _lib.ptr(ws), nb, _lib.stream())
            with pytest.raises(Exception):
                _lib.call('nele_imcra_band_ws', _lib.ptr(spec), _lib.ptr(frames), B, T, 1.0 / 6, _lib.ptr(psd), _lib.ptr(band), _lib.ptr(ws), nb - 256, _lib.stream())
        else:
            _lib.call('nele_imcra_band_var', _lib.ptr(spec), _lib.ptr(frames), B, T, 1.0 / 6, _lib.ptr(psd), _lib.ptr(band), _lib.stream())
        torch.cuda.synchronize()
        out.append((psd.cpu().numpy(), band.cpu().numpy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.isfinite(out[1][0]).all() and (out[1][0] >= 0).all()
    if T > 40:
        assert (out[1][0][0, 20:40] > 0).all()


def test_noise_band_from_the_power_spectrum_equals_the_route_through_the_spectrum():
    """audio_util.noise_band (nele_stft_pow_var: the STFT kernel writes |Y|^2 itself; nele_imcra_band_pw: IMCRA from it) against
    imcra_band(stft_band(noise)) through the ONE-kernel recursion: PSD and band feature bit for bit, short rows of a padded batch included."""
    from nele_gan_amd import _lib
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import synth
    B, L = 6, 50000
    c, v = synth.batch(B, L, start=90)
    x = torch.from_numpy(v).cuda()
    for lens in (None, torch.tensor([50000, 49000, 31000, 50000, 4100, 26000], dtype=torch.int32)):
        T = 1 + L // 256
        spec, _ = au.stft_band(x, want_band=False, lengths=lens)
        frames = au.frames_of(au._i32(lens, x.device))
        psd0 = torch.empty((B, T, 257), dtype=torch.float32, device='cuda')
        band0 = torch.empty((B, T, 64), dtype=torch.float32, device='cuda')
        _lib.call('nele_imcra_band_var', _lib.ptr(spec), _lib.ptr(frames), B, T, 1.0 / 6, _lib.ptr(psd0), _lib.ptr(band0), _lib.stream())
        psd1, band1 = au.noise_band(x, lengths=lens, want_psd=True)
        assert torch.equal(psd0, psd1) and torch.equal(band0, band1)
        assert torch.equal(au.noise_band(x, lengths=lens), band0)
