"""GPU: batched metric kernels (through the C ABI) against the oracle."""
import os
import wave

import numpy as np
import pytest
from conftest import ab_env

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
HERE = os.path.dirname(__file__)


def toy(name):
    w = wave.open(os.path.join(HERE, 'golden', 'toy', name))
    return np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0


@pytest.fixture(scope='module')
def mt():
    assert torch.cuda.is_available()
    from nele_gan_amd import metrics
    return metrics


def test_estoi_toy_file_vs_oracle(mt):
    from oracle import estoi
    x, v = toy('Train_Clean.wav'), toy('Train_Noise.wav')
    ys = np.stack([x + v, x + np.float32(0.25) * v, x, np.float32(3.0) * (x + v)])
    xs = np.stack([x] * 4)
    raw, mapped = mt.batch_estoi(xs, ys)
    raw, mapped = raw.cpu().numpy(), mapped.cpu().numpy()
    for b in range(4):
        ref = estoi.estoi(xs[b], ys[b])
        assert raw[b] == pytest.approx(ref, rel=1e-4, abs=1e-6)
        assert mapped[b] == pytest.approx(1 / (1 + np.exp(-8 * (ref - 0.25))), rel=1e-4)
    assert raw[2] == pytest.approx(1.0, abs=1e-6)
    assert mt.ESTOI_Wrapper_raw_harvard(x, x + v, 16000) == pytest.approx(float(raw[0]), rel=1e-6)


@pytest.mark.parametrize('L', [64000, 20001])
def test_estoi_synthetic_batch_vs_oracle(mt, L):
    from nele_gan_amd import synth
    from oracle import estoi
    c, v = synth.batch(4, L, start=40)
    y = c + v
    raw, _ = mt.batch_estoi(c, y)
    raw = raw.cpu().numpy()
    for b in range(4):
        assert raw[b] == pytest.approx(estoi.estoi(c[b], y[b]), rel=1e-4, abs=1e-6)


def test_estoi_too_short_returns_pystoi_constant(mt):
    x = toy('Train_Clean.wav')[:3000]
    raw, _ = mt.batch_estoi(x, x)
    assert float(raw[0]) == pytest.approx(1e-5, rel=1e-6)


def test_siib_toy_file_vs_oracle(mt):
    from oracle import siib
    x, v = toy('Train_Clean.wav'), toy('Train_Noise.wav')
    ys = np.stack([x + v, x + np.float32(0.25) * v, x])
    xs = np.stack([x] * 3)
    raw, mapped, info = mt.batch_siib(xs, ys, return_info=True)
    raw, mapped, info = raw.cpu().numpy(), mapped.cpu().numpy(), info.cpu().numpy()
    assert list(info[0][:1]) == [14]                                   # replication factor pinned by the reference (intel.npz)
    for b in range(3):
        ref = siib.siib_wrapper(xs[b], ys[b], norm=False)
        assert raw[b] == pytest.approx(ref, rel=1e-4, abs=1e-3)
        assert mapped[b] == pytest.approx(1 / (1 + np.exp(-0.06 * (ref - 32))), rel=1e-4)
    assert raw[2] == pytest.approx(80 / 15 * 420 * (-0.5 * np.log2(1 - 0.5625)), rel=1e-5)


@pytest.mark.parametrize('L', [64000, 47777, 48200])
def test_siib_synthetic_batch_vs_oracle(mt, L):
    from nele_gan_amd import synth
    from oracle import siib
    c, v = synth.batch(3, L, start=60)
    y = c + v
    raw, _, info = mt.batch_siib(c, y, return_info=True)
    raw, info = raw.cpu().numpy(), info.cpu().numpy()
    for b in range(3):
        ref, parts = None, None
        M, _ = __import__('oracle.intel', fromlist=['x']).siib_replication(c[b])
        assert info[b][0] == M
        ref = siib.siib_wrapper(c[b], y[b], norm=False)
        assert raw[b] == pytest.approx(ref, rel=1e-4, abs=1e-3)


def test_haspi_matches_reference_golden_24k(mt):
    import sys
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    from make_golden_haspi import golden_dither
    G = np.load(os.path.join(HERE, 'golden', 'haspi.npz'))
    x, y = G['x'], G['y']
    dx, dy = golden_dither(int(G['seed']), len(x), int(G['n_active']))
    nsub = int(G['n_sub'])
    d = np.zeros((1, 2, nsub, 32))
    d[0, 0, :dx.shape[0]] = dx
    d[0, 1, :dy.shape[0]] = dy
    raw, mapped, info = mt.batch_haspi(x, y, fs=24000, dither=torch.from_numpy(d), return_info=True)
    assert int(info[0, 0]) == int(G['n_active']) and int(info[0, 1]) == 0     # silence gate: integer-exact
    assert float(raw[0]) == pytest.approx(float(G['intel']), rel=1e-4)
    assert float(mapped[0]) == pytest.approx(1 / (1 + np.exp(-0.95 * (float(G['intel']) - 2.8))), rel=1e-4)


def test_haspi_16k_batch_vs_oracle(mt):
    from nele_gan_amd import synth
    from oracle import haspi as H
    c, v = synth.batch(3, 24000, start=80)
    y = c + v
    y[2] = c[2]                                                        # identical pair: every |rho| = 1
    raw, mapped = mt.batch_haspi(c, y, fs=16000, dither=None)
    raw = raw.cpu().numpy()
    for b in range(3):
        ref, _ = H.haspi_v2(c[b], 16000, y[b], 16000)
        assert raw[b] == pytest.approx(ref, rel=1e-4)
    assert raw[2] == pytest.approx(float(np.sum(H.WEIGHTS)), rel=1e-6)


@pytest.mark.parametrize('fs,L', [(22050, 33075), (8000, 12000), (16000, 24001), (11025, 16001)])
def test_haspi_other_sampling_rates_vs_oracle(mt, fs, L):
    """pyhaspi2.py:810-821: any rate below 24 kHz goes through librosa.resample (the reference's own demo pair is 22.05 kHz);
    odd lengths exercise librosa's fix_length padding (ceil(L * ratio) samples, the last one zero)."""
    from nele_gan_amd import synth
    from oracle import haspi as H
    c, v = synth.batch(2, L, start=85)
    y = c + v
    raw, mapped, info = mt.batch_haspi(c, y, fs=fs, dither=None, return_info=True)
    raw = raw.cpu().numpy()
    for b in range(2):
        ref, parts = H.haspi_v2(c[b], fs, y[b], fs, return_parts=True)
        assert int(info[b, 0]) == len(parts['index'])                 # active sub-frames: integer-exact
        assert raw[b] == pytest.approx(ref, rel=1e-4)
    q = mt.batch_haspi_quality(c, y, fs=fs, noise=False).cpu().numpy()
    for b in range(2):
        assert q[b, 0] == pytest.approx(H.haspi_v1(c[b], fs, y[b], fs)[0], rel=1e-4)


def test_haspi_above_24k_is_refused_like_the_reference(mt):
    from nele_gan_amd import synth
    c, v = synth.batch(1, 44100, start=86)
    with pytest.raises(Exception):                                     # NotImplementedError at pyhaspi2.py:819-820
        mt.batch_haspi(c, c + v, fs=44100, dither=None)


def test_haspi_random_dither_is_small_and_seeded(mt):
    from nele_gan_amd import synth
    c, v = synth.batch(1, 24000, start=90)
    r0, _ = mt.batch_haspi(c, c + v, dither=None)
    r1, _ = mt.batch_haspi(c, c + v, dither=True, seed=7)
    r2, _ = mt.batch_haspi(c, c + v, dither=True, seed=7)
    assert float(r1[0]) == float(r2[0])
    assert abs(float(r1[0]) - float(r0[0])) < 0.02 * abs(float(r0[0]))    # N(0, 0.1 dB) jitter: per-mille level effect


# 17 / 33: ragged cluster launches; 300 / 227 / 448: other hand-over points of the cluster -> register kernel chain (227 = one cluster step),
# 448 / 512: the 32-tile variant of the blocked back-transformation
@pytest.mark.parametrize('n,B', [(420, 3), (97, 2), (16, 4), (420, 17), (420, 33), (512, 9), (300, 5), (227, 3), (226, 2), (448, 2)])
def test_batched_eigensolver_vs_numpy(mt, n, B):
    rs = np.random.RandomState(n)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]       # covariance-like, decaying spectrum
        A[b] = G @ G.T / (3 * n)
    lam, U = mt.eigh_batched(torch.from_numpy(A).cuda())
    lam, U = lam.cpu().numpy(), U.cpu().numpy()
    for b in range(B):
        ref = np.linalg.eigvalsh(A[b])
        np.testing.assert_allclose(lam[b], ref, rtol=0, atol=1e-13 * ref.max())
        V = U[b].T                                                             # columns = eigenvectors
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-8
        assert np.abs(A[b] @ V - V * lam[b][None, :]).max() < 1e-11 * ref.max()


def test_siib_split_by_data_dependence_equals_one_shot(mt):
    """clean_part() + degraded_part(y) (the order GanTrainer uses) must give exactly the one-shot result."""
    from nele_gan_amd import synth
    c, v = synth.batch(3, 40000, start=11)
    x = torch.from_numpy(c).cuda()
    y = torch.from_numpy(c + v).cuda()
    raw0, map0 = mt.batch_siib(x, y)
    raw0, map0 = raw0.clone(), map0.clone()
    sp = mt.SiibSplit(x)
    sp.clean_part()
    raw1, map1 = sp.degraded_part(y)
    assert torch.equal(raw0, raw1) and torch.equal(map0, map1)
    y2 = torch.from_numpy(c + 2.0 * v).cuda()           # a second degraded signal against the same clean part
    raw2, _ = sp.degraded_part(y2)
    raw2 = raw2.clone()
    raw3, _ = mt.batch_siib(x, y2)
    assert torch.equal(raw2, raw3)


_AB_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for L in (64000, 48200, 32000):
    c, v = synth.batch(3, L, start=60)
    raw, _, info = mt.batch_siib(c, c + v, return_info=True)
    out += [raw.cpu().numpy().view(np.uint32), info.cpu().numpy().astype(np.uint32).ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_siib_period_shortcut_is_bit_identical_to_computing_every_frame(tmp_path):
    """L a multiple of 200: the tiled signal is frame-periodic; spectra / masking of the repeats are copied, not recomputed.
    A/B against the same library with the shortcut switched off (separate processes: the switch is read once)."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('siib_%s.npy' % flag))
        env = dict(os.environ, NELE_SIIB_DEDUP=flag)
        subprocess.run([sys.executable, '-c', _AB_CHILD, os.path.dirname(HERE), out], check=True, env=env, timeout=240)
        res.append(np.load(out))
    assert res[0].tobytes() == res[1].tobytes()


_LAG_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for L, B in ((64000, 3), (63871, 3), (33536, 2), (18000, 2)):
    c, v = synth.batch(B, L, start=70)
    raw, _, info = mt.batch_siib(c, 0.8 * c + v, return_info=True)
    out += [raw.double().cpu().numpy(), info.cpu().numpy().astype(np.float64).ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_siib_lag_products_equal_the_stacked_frame_gemms(tmp_path):
    """Round 3: covariance and the per-component sums come from lag products of the 28 band rows (the stacked frame is 15 consecutive
    frames: every entry of Xs Xs^T / Ys Ys^T / Xs Ys^T is a lag product) and quadratic forms u^T S u instead of 420 x 420 x n GEMMs
    over the stacked frames.  Same mathematics in another summation order: A/B against the round-2 kernels of the same library
    (NELE_SIIB_LAG=0, read once per process), frame-periodic and aperiodic lengths, a short file with a large replication factor."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('siib_lag_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _LAG_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_SIIB_LAG=flag), timeout=240)
        res.append(np.load(out))
    assert np.all(np.isfinite(res[0])) and res[0].shape == res[1].shape
    np.testing.assert_allclose(res[0], res[1], rtol=3e-7, atol=0)          # float32 outputs: at most a couple of ulps apart


_SPECW_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for L, B in ((64000, 3), (63871, 3), (33536, 2), (18000, 2), (4001, 2)):
    c, v = synth.batch(B, L, start=77)
    raw, _, info = mt.batch_siib(c, 0.8 * c + v, return_info=True)
    out += [raw.double().cpu().numpy(), info.cpu().numpy().astype(np.float64).ravel()]
c, v = synth.batch(4, 48000, start=78)
raw, _, info = mt.batch_siib(c, c + v, lengths=torch.tensor([48000, 30001, 9999, 41234], dtype=torch.int32), return_info=True)
out += [raw.double().cpu().numpy(), info.cpu().numpy().astype(np.float64).ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_siib_wave_autonomous_spectrum_kernel_matches_the_workgroup_kernel(tmp_path):
    """The 400-point spectra + band energies run one wave per three frames (samples straight into registers, radix-2 split of the two
    20-point stages, wave-level ordering only).  Same transform in another association: scores against the kernel it replaces
    (NELE_SIIB_SPECW=0, read once per process) - periodic and aperiodic lengths, short files with large replication factors (frames
    that run over the end of the tiled signal), a padded batch of different lengths; VAD / frame counts must be identical."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('siib_specw_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _SPECW_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_SIIB_SPECW=flag), timeout=240)
        res.append(np.load(out))
    assert np.all(np.isfinite(res[0])) and res[0].shape == res[1].shape
    np.testing.assert_allclose(res[0], res[1], rtol=3e-7, atol=0)          # float32 outputs: at most a couple of ulps apart


_ESTOI_AB_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for L, B in ((64000, 3), (63871, 2), (20011, 2)):
    c, v = synth.batch(B, L, start=75)
    raw, _ = mt.batch_estoi(c, 0.8 * c + v)
    out.append(raw.double().cpu().numpy())
c, v = synth.batch(4, 48000, start=76)
raw, _ = mt.batch_estoi(c, c + v, lengths=torch.tensor([48000, 30001, 9999, 41234], dtype=torch.int32))
out.append(raw.double().cpu().numpy())
np.save(sys.argv[2], np.concatenate(out))
'''


def test_estoi_five_output_resampler_is_bit_identical_to_the_output_per_thread_one(tmp_path):
    """The 16 -> 10 kHz resampler computes five consecutive outputs per thread (inputs read once, tap weights by scalar loads); every
    output's own sum is unchanged, so the scores must equal those of the kernel it replaces (NELE_ESTOI_RS5=0, read once per process)
    exactly - whole and ragged batches, lengths that are no multiple of 8 (the last group of five is partial)."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('estoi_rs5_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _ESTOI_AB_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_ESTOI_RS5=flag), timeout=240)
        res.append(np.load(out))
    assert np.all(np.isfinite(res[0])) and res[0].shape == (11,)
    assert np.array_equal(res[0], res[1])


def test_estoi_wave_per_frame_third_octave_kernel_matches_the_workgroup_kernel(tmp_path):
    """The third-octave analysis runs one wave per kept frame with the 512-point transform in registers (fft512_wave).  Same transform,
    but estoi.hip is compiled with FMA contraction, so the two kernels may round differently: scores against the workgroup-per-frame
    kernel (NELE_ESTOI_TOBW=0, read once per process) to float32 resolution - whole and ragged batches, short files."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('estoi_tobw_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _ESTOI_AB_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_ESTOI_TOBW=flag), timeout=240)
        res.append(np.load(out))
    assert np.all(np.isfinite(res[0])) and res[0].shape == (11,)
    np.testing.assert_allclose(res[0], res[1], rtol=3e-7, atol=1e-7)


_HASPI_AB_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
c, v = synth.batch(4, 64000, start=33)
raw, mapped = mt.batch_haspi(c, c + v, dither=None)
np.save(sys.argv[2], raw.double().cpu().numpy())
'''


def test_haspi_chunk_parallel_filters_equal_the_serial_ones(tmp_path):
    """The linear recurrences (middle ear, gammatone banks, gain low-pass) run parallel over chunks with a warm-up that makes the
    neglected history < 1e-20 of the signal; A/B against the serial kernels of the same library (switches are read once per process)."""
    import subprocess
    import sys
    res = []
    for env in ({}, {'NELE_HASPI_PAR_IIR': '0', 'NELE_HASPI_FUSED_GAIN': '0'}):
        out = str(tmp_path / ('haspi_%d.npy' % len(res)))
        subprocess.run([sys.executable, '-c', _HASPI_AB_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(**env), timeout=240)
        res.append(np.load(out))
    assert res[0].shape == (4,) and np.all(np.isfinite(res[0]))
    np.testing.assert_allclose(res[0], res[1], rtol=3e-7, atol=0)          # float32 outputs: at most a couple of ulps apart


def test_haspi_sliding_modulation_filters_equal_the_direct_fir(tmp_path):
    """The modulation filter bank runs as three sliding sums per Hann window (O(1) per output); A/B against the direct-form FIR
    kernel of the same library (NELE_HASPI_MOD_DIRECT=1, read once per process)."""
    import subprocess
    import sys
    res = []
    for env in ({}, {'NELE_HASPI_MOD_DIRECT': '1'}):
        out = str(tmp_path / ('haspi_mod_%d.npy' % len(res)))
        subprocess.run([sys.executable, '-c', _HASPI_AB_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(**env), timeout=240)
        res.append(np.load(out))
    assert res[0].shape == (4,) and np.all(np.isfinite(res[0]))
    np.testing.assert_allclose(res[0], res[1], rtol=3e-7, atol=0)          # float32 outputs: at most a couple of ulps apart


_HASPI_RAGGED_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
c, v = synth.batch(5, 63871, start=21)
lens = torch.tensor([63871, 20011, 48000, 5120, 33333], dtype=torch.int32)
raw, mapped = mt.batch_haspi(c, c * 0.8 + v, dither=None, lengths=lens)
np.save(sys.argv[2], raw.double().cpu().numpy())
'''


@pytest.mark.parametrize('env', [{'NELE_HASPI_FIR9': '0'}, {'NELE_HASPI_CEP_SERIAL': '0'}, {'NELE_HASPI_TAIL1': '0'}, {'NELE_HASPI_MOD_DIRECT': '1'}, {'NELE_HASPI_RS3': '0'}],
                         ids=['fir8-frame-rows', 'parallel-cepstra', 'whole-chunk-pass1', 'direct-modulation-fir', 'resampler-output-per-thread'])
def test_haspi_round3_kernels_equal_the_ones_they_replace(tmp_path, env):
    """Round 3 restructured HASPI's envelope filter (groups of nine samples, outputs stored as whole rows per GROUP and read back
    through the per-channel frame offset), the cepstrum stage (LDS-staged tiles), pass 1 of the filter banks (chunk heads skipped) and the
    modulation filters (LDS ring, rotating frame): each against the kernel it replaced (switches are read once per process), on a padded
    batch of utterances of different lengths - the row ends, the first frames (windows that start before the signal) and the
    frame-offset addressing all differ per utterance and channel."""
    import subprocess
    import sys
    res = []
    for e in ({}, env):
        out = str(tmp_path / ('haspi_r3_%d.npy' % len(res)))
        subprocess.run([sys.executable, '-c', _HASPI_RAGGED_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(**e), timeout=240)
        res.append(np.load(out))
    assert res[0].shape == (5,) and np.all(np.isfinite(res[0]))
    np.testing.assert_allclose(res[0], res[1], rtol=1e-6, atol=0)           # float32 outputs; multiply-add chains differ in the last bits


def test_eigensolver_on_matrices_that_put_exact_zeros_into_the_sturm_recurrence(mt):
    """Diagonal, repeated and block-diagonal matrices: the Sturm sequence hits exact zeros (an evaluation point equal to an eigenvalue
    of a leading block); the bisection kernel does not repair them on its critical chain and must still converge."""
    n = 64
    A = np.zeros((4, n, n))
    A[0] = np.diag(np.arange(1, n + 1, dtype=float))
    A[1] = np.eye(n); A[1][3, 3] = 2.0
    A[2] = np.diag(np.arange(1, n + 1, dtype=float)); A[2][:8, :8] += 0.5
    rs = np.random.RandomState(1)
    G = rs.randn(n, n)
    A[3] = np.kron(np.eye(2), (G @ G.T)[:32, :32])
    lam, U = mt.eigh_batched(torch.from_numpy(A).cuda())
    lam, U = lam.cpu().numpy(), U.cpu().numpy()
    for b in range(4):
        ref = np.linalg.eigvalsh(A[b])
        V = U[b].T
        np.testing.assert_allclose(lam[b], ref, rtol=0, atol=1e-13 * ref.max())
        assert np.abs(A[b] @ V - V * lam[b][None, :]).max() < 1e-11 * ref.max()


def test_siib_is_batch_invariant_across_cluster_launches(mt):
    """72 utterances = two launches of the 4-workgroup tridiagonalisation (64 + 8) and one tail launch: every utterance's score must equal
    the score it gets in a small batch (no cross-talk between matrices, hand-over state per matrix)."""
    from nele_gan_amd import synth
    c, v = synth.batch(72, 40000, start=200)
    y = c + v
    raw, _ = mt.batch_siib(c, y)
    raw = raw.cpu().numpy()
    assert np.isfinite(raw).all()
    sub, _ = mt.batch_siib(c[62:68], y[62:68])
    assert np.array_equal(raw[62:68], sub.cpu().numpy())


def test_haspi_split_by_data_dependence_equals_one_shot(mt):
    """clean_part() + degraded_part(y) (the order GanTrainer uses) must give exactly the one-shot result; a second degraded signal
    against the same clean part too; and the dither rows of x / y are consumed by their own halves."""
    from nele_gan_amd import synth
    c, v = synth.batch(3, 40000, start=21)
    x = torch.from_numpy(c).cuda()
    y = torch.from_numpy(c + v).cuda()
    raw0, map0, info0 = mt.batch_haspi(x, y, return_info=True)
    raw0, map0 = raw0.clone(), map0.clone()
    sp = mt.HaspiSplit(x)
    sp.clean_part()
    raw1, map1 = sp.degraded_part(y)
    assert torch.equal(raw0, raw1) and torch.equal(map0, map1) and torch.equal(info0, sp.info)
    y2 = torch.from_numpy(c + 2.0 * v).cuda()
    raw2 = sp.degraded_part(y2)[0].clone()
    raw2_ref = mt.batch_haspi(x, y2)[0]
    assert torch.equal(raw2, raw2_ref) and not torch.equal(raw2, raw0)
    nsub = mt._lib.lib.nele_metric_haspi_nsub(40000, 16000)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    d = torch.randn((3, 2, nsub, 32), dtype=torch.float64, device='cuda', generator=g)
    raw3 = mt.batch_haspi(x, y, dither=d)[0].clone()
    sp.clean_part(dither=d)
    raw4 = sp.degraded_part(y, dither=d)[0]
    assert torch.equal(raw3, raw4) and not torch.equal(raw3, raw0)


def test_haspi_per_utterance_lengths_in_one_padded_batch(mt):
    """Three utterances of different lengths side by side in one padded [B, Lmax] launch: every score equals the score of the same
    utterance launched alone at its own length (the reference scores files one at a time, audio_util.py:134-141)."""
    from nele_gan_amd import synth
    lens = [40000, 33536, 26001]
    c, v = synth.batch(3, 40000, start=51)
    y = c + v
    xp, yp = c.copy(), y.copy()
    for b, n in enumerate(lens):
        xp[b, n:] = 0.0
        yp[b, n:] = 7.0                                                     # garbage behind the end must not matter
    raw, mapped, info = mt.batch_haspi(xp, yp, lengths=lens, return_info=True)
    raw, info = raw.cpu().numpy(), info.cpu().numpy()
    for b, n in enumerate(lens):
        r1, _, i1 = mt.batch_haspi(c[b:b + 1, :n], y[b:b + 1, :n], return_info=True)
        assert raw[b] == pytest.approx(float(r1[0]), rel=1e-6), (b, raw[b], float(r1[0]))   # chunk seams sit elsewhere: warm-up residue 1e-20
        assert info[b, 0] == int(i1[0, 0])


def test_siib_score_of_an_utterance_across_batch_size_classes(mt):
    """csrc/siib.hip sums the lag products of an utterance as four frame segments whose boundaries come from the utterance's own
    active-frame count, combined as (s0 + s1) + (s2 + s3); the batch size only decides how many workgroups share the segments (4, 2 or 1).
    Round 5: an utterance's raw score in a batch of 1, of 50 and of 130 is the SAME float32 (r04: 1e-6 apart, the segment boundaries
    followed the batch-size class); M / frame counts are identical."""
    from nele_gan_amd import synth
    L = 24000
    c, v = synth.batch(2, L, start=900)
    y = (0.8 * c + v).astype(np.float32)
    ref = None
    for B in (1, 50, 130):
        x_ = torch.from_numpy(np.repeat(c[:1], B, axis=0)).cuda()
        y_ = torch.from_numpy(np.repeat(y[:1], B, axis=0)).cuda()
        x_[B - 1], y_[B - 1] = torch.from_numpy(c[1]).cuda(), torch.from_numpy(y[1]).cuda()     # not all rows alike
        if B == 1:
            x_, y_ = torch.from_numpy(c[:1]).cuda(), torch.from_numpy(y[:1]).cuda()
        raw, _, info = mt.batch_siib(x_, y_, return_info=True)
        r0, i0 = float(raw[0].double()), info[0].cpu().numpy()
        if ref is None:
            ref = (r0, i0)
        assert r0 == ref[0], (B, r0, ref[0])                      # bit-identical across batch-size classes
        np.testing.assert_array_equal(i0[:3], ref[1][:3])


_REPAIR_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for n, B in ((420, 8), (300, 5)):
    rs = np.random.RandomState(n)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]
        A[b] = G @ G.T / (3 * n)
    lam, U, rep = mt.eigh_batched(torch.from_numpy(A).cuda(), return_repaired=True)
    lam, U = lam.cpu().numpy(), U.cpu().numpy()
    worst = [0.0, 0.0, 0.0]
    for b in range(B):
        ref = np.linalg.eigvalsh(A[b])
        V = U[b].T
        worst[0] = max(worst[0], np.abs(lam[b] - ref).max() / ref.max())
        worst[1] = max(worst[1], np.abs(V.T @ V - np.eye(n)).max())
        worst[2] = max(worst[2], np.abs(A[b] @ V - V * lam[b][None, :]).max() / ref.max())
    out += [float(rep)] + worst
c, v = synth.batch(4, 32000, start=91)
raw, _, info = mt.batch_siib(c, 0.8 * c + v, return_info=True)
out += list(raw.double().cpu().numpy()) + list(info[:, 3].double().cpu().numpy())
np.save(sys.argv[2], np.array(out))
'''


def test_cluster_tridiagonalisation_give_up_is_repaired_not_poisoned(tmp_path):
    """Round 4: when the workgroups of a matrix are not co-resident within the spin limit the cluster kernel used to poison the matrix
    with NaN (the optimiser step was then masked).  Now it flags the matrix and one workgroup redoes it from the untouched lower
    triangle (eigh_tridiag_repair_kernel).  NELE_EIGH_FAIL_EVERY=3 (read once per process) sends every third matrix down that path:
    same accuracy against numpy as the fast path, the repair count is reported, and SIIB scores equal the unforced run's."""
    import subprocess
    import sys
    res = []
    for k in ('3', '0', '-3'):
        out = str(tmp_path / ('repair_%s.npy' % k))
        subprocess.run([sys.executable, '-c', _REPAIR_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_EIGH_FAIL_EVERY=k), timeout=240)
        res.append(np.load(out))
    forced, plain, forced2 = res
    assert forced[0] == 3 and forced[4] == 2 and plain[0] == 0 and plain[4] == 0           # matrices 0, 3, 6 of 8; 0, 3 of 5
    # -3: the give-up happens in the SECOND cluster stage (two workgroups per matrix, from 320 rows on; n = 300 has no such stage): the
    # matrix restarts from the first stage's hand-over (eigh_tridiag_repair2_kernel)
    assert forced2[0] == 3 and forced2[4] == 0
    np.testing.assert_allclose(forced2[8:12], plain[8:12], rtol=1e-6)
    for r in (forced, plain, forced2):
        for o in (0, 4):
            assert r[o + 1] < 1e-13 and r[o + 2] < 1e-8 and r[o + 3] < 1e-11
    np.testing.assert_allclose(forced[8:12], plain[8:12], rtol=1e-6)                       # SIIB raw scores (float32 outputs)
    # status words: the forced run marks the utterances whose covariance went through the repair path (bit 32: matrices 0 and 3 of 4) -
    # what GanTrainer.check_status() counts as 'eigh_repaired' - and is otherwise the unforced run's
    assert np.all(forced[12:] == plain[12:] + np.array([32, 0, 0, 32])) and np.all(plain[12:].astype(int) & 32 == 0) and np.all(np.isfinite(forced))


def test_eigensolver_beside_resident_workgroups_is_correct_and_reports_its_repairs():
    """What a collective library's resident kernels (or another process) do to the cluster tridiagonalisation: N CUs are held by idle
    workgroups that claim the whole LDS while 64 matrices of SIIB's size are decomposed on another stream.  The cluster kernels need their
    2 workgroups per matrix co-resident; when the launch no longer fits they wait, and past the spin limit they give up and the matrix is
    redone by one workgroup (eigh_repair1 inside eigh_tridiag_midx_kernel).  Whatever happens: finite, accurate results, no hang, and the
    number of repairs is reported.  Times and counts are printed (pytest -s) - DESIGN 4.3 quotes them."""
    import time
    from nele_gan_amd import metrics as mt
    from nele_gan_amd._lib import call
    import ctypes
    n, B = 420, 64
    rs = np.random.RandomState(5)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]
        A[b] = G @ G.T / (3 * n)
    At = torch.from_numpy(A).cuda()
    ref = [np.linalg.eigvalsh(A[b]) for b in (0, 17, 63)]
    mt.eigh_batched(At)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    rows = []
    for ncu in (0, 32, 64, 128, 192, 240):
        torch.cuda.synchronize()
        if ncu:
            call('nele_stream_occupy', ncu, 160 * 1024, 30000.0, ctypes.c_void_p(side.cuda_stream))     # 30 ms of resident workgroups
            time.sleep(0.002)                                             # let them start
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lam, U = mt.eigh_batched(At)
        e1.record()
        e1.synchronize()                                                  # the eigensolver's stream only: the resident workgroups are still there
        dt = e0.elapsed_time(e1)
        from nele_gan_amd import _lib
        ws = mt._workspace('eigh', _lib.lib.nele_eigh_workspace_bytes(B, n), At.device)
        rep = int(_lib.lib.nele_eigh_repaired(mt.ptr(ws), B, n))        # (synchronises the device: waits for the occupiers too)
        torch.cuda.synchronize()
        rows.append((ncu, dt, rep))
        lam_h = lam.cpu().numpy()
        assert np.isfinite(lam_h).all() and torch.isfinite(U).all()
        for k, b in enumerate((0, 17, 63)):
            assert np.abs(lam_h[b] - ref[k]).max() <= 1e-12 * ref[k].max()
        assert 0 <= rep <= B
    print('eigh_batched(64 x 420) beside N occupied CUs: ' + ', '.join('N=%d: %.1f ms, %d repaired' % r for r in rows))
    assert rows[0][2] == 0                                                # nothing else on the GPU: the launch fits, no give-ups
    assert max(r[1] for r in rows) < 2000.0                               # and never a hang


_INVIT_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt
out = []
for n, B in ((420, 3), (257, 2), (64, 2), (17, 2)):
    rs = np.random.RandomState(1000 + n)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]
        A[b] = G @ G.T / (3 * n)
    lam, U = mt.eigh_batched(torch.from_numpy(A).cuda())
    out += [lam.cpu().numpy().ravel(), U.cpu().numpy().ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_inverse_iteration_with_recomputed_factors_is_bit_identical_to_the_stored_form(tmp_path):
    """Round 4: eigh_invit2_kernel keeps no LU factors (b, multipliers, 1 / pivot: 57 KB of traffic per eigenvector, the kernel was
    HBM-bound on them) but a checkpoint of the three-value forward recurrence every 16 steps, recomputes the factors in lock-step with
    the forward elimination and chunk by chunk ahead of the back substitution: the same operations on the same operands in the same
    order.  A/B against eigh_invit_kernel (NELE_EIGH_INVIT_STORE=1, test library) at n with partial last chunks / pairs: eigenvalues and
    eigenvectors bit for bit."""
    import subprocess
    import sys
    res = []
    for flag in ('0', '1'):
        out = str(tmp_path / ('invit_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _INVIT_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_EIGH_INVIT_STORE=flag), timeout=240)
        res.append(np.load(out))
    assert np.all(np.isfinite(res[0])) and res[0].tobytes() == res[1].tobytes()


_SYM_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
out = []
for n, B in ((420, 5), (432, 2), (400, 3), (340, 2)):
    rs = np.random.RandomState(2000 + n)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]
        A[b] = G @ G.T / (3 * n)
    lam, U = mt.eigh_batched(torch.from_numpy(A).cuda())
    lam, U = lam.cpu().numpy(), U.cpu().numpy()
    worst = [0.0, 0.0]
    for b in range(B):
        V = U[b].T
        worst[0] = max(worst[0], np.abs(V.T @ V - np.eye(n)).max())
        worst[1] = max(worst[1], np.abs(A[b] @ V - V * lam[b][None, :]).max() / np.abs(lam[b]).max())
    out += [lam.ravel() / np.abs(lam).max(), np.array(worst)]
c, v = synth.batch(3, 40000, start=95)
raw, _, info = mt.batch_siib(c, 0.8 * c + v, return_info=True)
out += [raw.double().cpu().numpy(), info.double().cpu().numpy().ravel()]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_symmetric_first_cluster_stage_agrees_with_the_full_storage_kernel(tmp_path):
    """Round 4: the first tridiagonalisation stage holds only the LOWER triangle of the trailing matrix (two workgroups per matrix instead
    of four: column sums as before, row sums by a reduce-scatter over the 32 column classes of a half wave, four kinds of exchange
    slots).  Same Householder recurrences in another summation order: eigenvalues agree with the four-workgroup full-storage kernel
    (NELE_EIGH_SYM=0, test library) to 1e-13 of the largest, orthogonality / residual stay at their level, SIIB scores to float32
    precision; n = 432 is the largest order the layout takes (27 row slots of 16), 340 has only 18 first-stage steps."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('sym_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _SYM_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_EIGH_SYM=flag), timeout=240)
        res.append(np.load(out))
    a, b = res
    assert a.shape == b.shape and np.all(np.isfinite(a))
    nl = 420 * 5 + 2 + 432 * 2 + 2 + 400 * 3 + 2 + 340 * 2 + 2
    np.testing.assert_allclose(a[:nl], b[:nl], rtol=0, atol=1e-8)          # (includes the orthogonality / residual pairs, all < 1e-8)
    pos = 0
    for n, B in ((420, 5), (432, 2), (400, 3), (340, 2)):
        np.testing.assert_allclose(a[pos:pos + n * B], b[pos:pos + n * B], rtol=0, atol=1e-13)
        assert a[pos + n * B] < 1e-8 and a[pos + n * B + 1] < 1e-11
        pos += n * B + 2
    np.testing.assert_allclose(a[nl:nl + 3], b[nl:nl + 3], rtol=1e-6)
    assert np.all(a[nl + 3:] == b[nl + 3:])


_KEEP_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt
out = []
for n, B in ((420, 40), (300, 9)):
    rs = np.random.RandomState(3000 + n)
    A = np.zeros((B, n, n))
    for b in range(B):
        G = rs.randn(n, 3 * n) * np.exp(-0.01 * np.arange(3 * n))[None, :]
        A[b] = G @ G.T / (3 * n)
    At = torch.from_numpy(A).cuda()
    for rep in range(2):                      # the first call of a process launches the probe; the second runs with its verdict
        lam, U, repaired = mt.eigh_batched(At, return_repaired=True)
    out += [lam.cpu().numpy().ravel(), U.cpu().numpy().ravel(), np.array([float(repaired)])]
np.save(sys.argv[2], np.concatenate(out))
'''


def test_exchange_stores_kept_in_the_xcd_l2_give_the_same_bits_and_no_give_ups(tmp_path):
    """Round 4, second session: the cluster kernels' tagged exchange stores stay in the XCD's L2 (`sc0`) once eigh_xch_probe_kernel has
    shown, once per device, that the workgroups of a matrix do sit on one XCD (128 workgroup pairs laid out like the real launches,
    64 ping-pong rounds each); otherwise they are written through (`sc1`), as before.  Same arithmetic either way: eigenvalues and
    eigenvectors are bit-identical to NELE_EIGH_XCH_KEEP=0 (test library), and no matrix runs into the spin limit (repaired == 0)."""
    import subprocess
    import sys
    res = []
    for flag in ('1', '0'):
        out = str(tmp_path / ('keep_%s.npy' % flag))
        subprocess.run([sys.executable, '-c', _KEEP_CHILD, os.path.dirname(HERE), out], check=True, env=ab_env(NELE_EIGH_XCH_KEEP=flag), timeout=240)
        res.append(np.load(out))
    a, b = res
    assert a.shape == b.shape and np.all(np.isfinite(a)) and np.array_equal(a, b)
    n1 = 420 * 40 + 420 * 420 * 40
    assert a[n1] == 0.0 and a[-1] == 0.0
