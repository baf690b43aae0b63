#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py [--ref /root/reference] [--only NAME ...]

The reference checkout never travels to the GPU box, so everything the tests need is dumped here
as small .npz files (inputs + expected outputs).  Only data is written: no reference source.

Third-party modules the reference imports at module level but that are not installed here
(librosa, numba, pysiib, pystoi, pypesq, soundfile) are satisfied by empty in-memory stand-ins so
that `import audio_util / intel / pyhaspi2` succeeds; none of the goldens below is produced by a
function those stand-ins replace (numba.jit is the identity decorator: the decorated *reference*
functions still run, un-jitted).  scipy >= 1.13 dropped the 'hanning' window alias that
intel.py:90 uses; it is mapped to 'hann' (same periodic window) for the duration of the run.
"""
import argparse
import os
import sys
import types
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def install_import_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _absent(*a, **k):
        raise RuntimeError("third-party function not available in the build container")

    mod('librosa', stft=_absent, istft=_absent, load=_absent, resample=_absent)
    mod('numba', jit=lambda *a, **k: (lambda f: f))
    mod('pysiib', SIIB=_absent)
    mod('pystoi', stoi=_absent)
    sys.modules['pystoi'].stoi = mod('pystoi.stoi', stoi=_absent)
    mod('pypesq', pesq=_absent)
    mod('soundfile', write=_absent, read=_absent)
    import scipy.signal
    import scipy.signal.windows as W
    _gw = scipy.signal.get_window

    def get_window(window, Nx, fftbins=True):
        if window == 'hanning':
            window = 'hann'
        return _gw(window, Nx, fftbins)
    scipy.signal.get_window = get_window


def read_wav(path):
    w = wave.open(path)
    assert w.getsampwidth() == 2 and w.getnchannels() == 1
    x = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
    return x, w.getframerate()


def speechlike_spectrum(seed, T):
    """complex64 [257, T] STFT of synthetic speech + noise (exercises the IMCRA branches)."""
    from nele_gan_amd import synth
    from oracle import features as F
    L = 256 * (T - 1)
    c = synth.clean_utterance(seed, L)
    v = synth.noise_utterance(seed, L, c)
    return F.stft((0.3 * c + v).astype(np.float32)), F.stft(v)


def gen_features(ref):
    import audio_util as A
    from oracle import features as F
    rng = np.random.RandomState(7)
    out = {}
    # compute_band_E
    X = np.abs(rng.randn(8, 257)).astype(np.float32) * np.float32(0.1)
    out['bandE_in'] = X
    out['bandE_out'] = A.compute_band_E(X)
    # the reference squares float32 *scalars* with `**2` (glibc powf, not correctly rounded: ~8e-4 of
    # values are 1 ulp off x*x); the oracle uses the IEEE product, so allow 2 float32 ulps here.
    assert np.allclose(out['bandE_out'], F.compute_band_E(X), rtol=2.5e-7, atol=0), "oracle compute_band_E != reference"
    # interp_band_gain + Resyn's gain
    a = np.exp(rng.randn(5, 64)).astype(np.float32)
    out['gain_in'] = a
    out['gain_out'] = np.stack([A.interp_band_gain(a[t]) for t in range(5)])
    for t in range(5):
        assert np.array_equal(out['gain_out'][t], F.interp_band_gain(a[t]))
    assert np.array_equal(out['gain_out'].T, F.interp_band_gain_batch(a))
    # IMCRA: two spectra, T=96 (l<15, l>=15, and 5 minima-buffer rotations)
    Y1, Y2 = speechlike_spectrum(3, 96)
    for k, Y in (('a', Y1), ('b', Y2)):
        psd = A.NoisePSD(Y)
        out['imcra_in_' + k] = Y
        out['imcra_out_' + k] = psd
        mine = F.imcra_noise_psd(Y)
        assert psd.dtype == np.float32
        if not np.array_equal(psd, mine):
            d = np.abs(psd - mine) / np.abs(psd)
            raise AssertionError("oracle IMCRA != reference: max rel %g at %s" % (d.max(), np.unravel_index(d.argmax(), d.shape)))
    # a longer one (T=170: the 9th/10th minima rotations at l=149,164 take the np.roll branch)
    Y3, _ = speechlike_spectrum(5, 170)
    psd = A.NoisePSD(Y3)
    assert np.array_equal(psd, F.imcra_noise_psd(Y3))
    out['imcra_in_c'] = Y3
    out['imcra_out_c'] = psd
    # noise band feature given F (everything after the STFT in Sp_and_phase_Noise, audio_util.py:446-451)
    bandE = A.compute_band_E(np.sqrt(psd.T)) ** (1 / 6)
    out['noise_band_c'] = bandE
    assert np.allclose(bandE, F.compute_band_E(np.sqrt(psd.T)) ** (1 / 6), rtol=2.5e-7, atol=0)
    # rms
    out['rms_in'] = X[0]
    out['rms_out'] = np.float64(A.rms(X[0]))
    np.savez_compressed(os.path.join(HERE, 'features.npz'), **out)
    print('features.npz written; oracle == reference: bit-exact gain + IMCRA (3 spectra), <=2 ulp band_E')


def gen_intel(ref):
    import intel as I
    out = {}
    x, fs = read_wav(os.path.join(HERE, 'toy', 'Train_Clean.wav'))
    assert fs == 16000
    fr = I.framing(x, 400, 200, 'hanning')
    vad = I.get_vad(x, 400, 200, 'hanning', 40)
    S = I.stft(x, 400, 200, 'hanning')
    out['n_frames'] = np.int64(fr.shape[0])
    out['frames_head'] = fr[:3]
    out['frames_tail'] = fr[-2:]
    out['vad'] = vad
    out['stft_head'] = S[:4]
    # the replication factor of SIIB_Wrapper (intel.py:93-97)
    R = 1 / 200 * 16000
    nact = int(vad.sum())
    out['n_active'] = np.int64(nact)
    out['M'] = np.int64(int(np.floor(25 / (nact / R)))) if nact / R < 20 else np.int64(1)
    pts = np.array([-5.0, 0.0, 0.25, 1.0, 2.8, 10.0, 32.0, 60.0, 200.0])
    out['map_pts'] = pts
    out['map_siib'] = I.mapping_SIIB_harvard(pts)
    out['map_haspi'] = I.mapping_HASPI_harvard(pts)
    out['map_estoi'] = I.mapping_ESTOI_harvard(pts)
    np.savez_compressed(os.path.join(HERE, 'intel.npz'), **out)
    print('intel.npz written: %d frames, %d active, M=%d' % (fr.shape[0], nact, out['M']))


def seeded_state(module, seed):
    """Load tests/weights_recipe.py weights (numpy RandomState recipe) into a reference module."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from weights_recipe import seeded_state_arrays
    sd = module.state_dict()
    arrs = seeded_state_arrays([(k, tuple(v.shape)) for k, v in sd.items()], seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in arrs.items()})
    return module


def put_grad(out, key, g):
    """Large gradients are stored as a digest (sum, abs-sum, 256 strided samples)."""
    from weights_recipe import digest
    g = np.asarray(g)
    if g.size <= 4096:
        out[key] = g
    else:
        s, a, smp = digest(g)
        out[key + '#sum'] = s
        out[key + '#abs'] = a
        out[key + '#smp'] = smp


def gen_model(ref):
    import torch
    import model as M
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    B, T = 2, 40
    rs = np.random.RandomState(11)
    x = (0.2 + 0.3 * rs.rand(B, T, 64)).astype(np.float32)
    y = (0.2 + 0.3 * rs.rand(B, T, 64)).astype(np.float32)
    G = seeded_state(M.Generator_Conv1D_cLN(), 101)
    D = seeded_state(M.Discriminator(), 202)
    DQ = seeded_state(M.Discriminator_Quality(), 303)
    out['g_keys'] = np.array(list(G.state_dict().keys()))
    out['d_keys'] = np.array(list(D.state_dict().keys()))
    out['dq_keys'] = np.array(list(DQ.state_dict().keys()))
    out['g_shapes'] = np.array([str(tuple(v.shape)) for v in G.state_dict().values()])
    out['d_shapes'] = np.array([str(tuple(v.shape)) for v in D.state_dict().values()])
    out['x'] = x
    out['y'] = y
    # ---- G forward + grads of a scalar loss
    xt = torch.from_numpy(x).requires_grad_(True)
    yt = torch.from_numpy(y).requires_grad_(True)
    mask = G(xt, yt)
    gw = torch.from_numpy(rs.randn(B, T, 64).astype(np.float32))
    out['g_gw'] = gw.numpy()
    (mask * gw).sum().backward()
    out['g_mask'] = mask.detach().numpy()
    out['g_dx'] = xt.grad.numpy()
    out['g_dy'] = yt.grad.numpy()
    for k, p in G.named_parameters():
        put_grad(out, 'g_grad.' + k, p.grad.numpy())
    # ---- D eval-mode forward (spectral norm uses stored u,v, no power iteration) + input grad
    D.eval()
    din = torch.from_numpy((0.2 + 0.5 * rs.rand(B, 3, 64, T)).astype(np.float32)).requires_grad_(True)
    out['d_in'] = din.detach().numpy()
    sc = D(din)
    tgt = torch.tensor([[1.0, 1.0, 1.0], [0.3, 0.6, 0.9]])
    loss = torch.nn.functional.mse_loss(sc, tgt)
    loss.backward()
    out['d_tgt'] = tgt.numpy()
    out['d_eval_score'] = sc.detach().numpy()
    out['d_eval_loss'] = np.float32(loss.item())
    out['d_eval_din_grad'] = din.grad.numpy()
    for k, p in D.named_parameters():
        put_grad(out, 'd_eval_grad.' + k, p.grad.numpy())
    # ---- D train-mode forward: one power iteration per layer, u/v buffers advance
    D2 = seeded_state(M.Discriminator(), 202)
    D2.train()
    din2 = torch.from_numpy(out['d_in']).requires_grad_(True)
    sc2 = D2(din2)
    loss2 = torch.nn.functional.mse_loss(sc2, tgt)
    loss2.backward()
    out['d_train_score'] = sc2.detach().numpy()
    out['d_train_din_grad'] = din2.grad.numpy()
    for k, p in D2.named_parameters():
        put_grad(out, 'd_train_grad.' + k, p.grad.numpy())
    for k, v in D2.state_dict().items():
        if k.endswith('_u') or k.endswith('_v'):
            out['d_train_buf.' + k] = v.numpy()
    # ---- D_Qua eval forward
    DQ.eval()
    qin = torch.from_numpy(out['d_in'][:, [0, 2]].copy())
    out['dq_eval_score'] = DQ(qin).detach().numpy()
    # ---- G-step glue (train_nele.py:130-152), batch 1, D/D_Qua in train mode as in the reference
    G.zero_grad()
    D3 = seeded_state(M.Discriminator(), 202)
    Q3 = seeded_state(M.Discriminator_Quality(), 303)
    cb = torch.from_numpy(x[:1])
    nb = torch.from_numpy(y[:1])
    p_power, inv_p = (1 / 6), 6
    mask = G(cb, nb)
    clean_power = torch.pow(cb.detach(), inv_p)
    beta_2 = torch.sum(clean_power) / torch.sum(mask * clean_power)
    beta_p = beta_2 ** p_power
    enh = cb * torch.pow(mask, p_power) * beta_p
    refb = cb.detach()
    e4 = enh.view(1, 1, T, 64).transpose(2, 3).contiguous()
    n4 = nb.view(1, 1, T, 64).transpose(2, 3).contiguous()
    r4 = refb.view(1, 1, T, 64).transpose(2, 3).contiguous()
    d_inputs = torch.cat((e4, n4, r4), dim=1)
    d_inputs_q = torch.cat((e4, r4), dim=1)
    score = D3(d_inputs)
    score_q = Q3(d_inputs_q)
    mse = torch.nn.MSELoss()
    lossg = mse(score, torch.ones(1, 3)) + 0.5 * mse(score_q, torch.ones(1, 2))
    lossg.backward()
    out['gstep_beta2'] = np.float32(beta_2.item())
    out['gstep_enh'] = enh.detach().numpy()
    out['gstep_d_inputs'] = d_inputs.detach().numpy()
    out['gstep_score'] = score.detach().numpy()
    out['gstep_score_q'] = score_q.detach().numpy()
    out['gstep_loss'] = np.float32(lossg.item())
    out['gstep_grad_fc2_w'] = G.fc2.weight.grad.numpy()
    out['gstep_grad_c0_w_sum'] = np.float64(G.convolutions[0][0].conv.weight.grad.double().sum().item())
    out['gstep_grad_c0_w_abs'] = np.float64(G.convolutions[0][0].conv.weight.grad.double().abs().sum().item())
    out['gstep_grad_c5_b'] = G.convolutions[5][0].conv.bias.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'model.npz'), **out)
    print('model.npz written: G mask range [%.4f, %.4f], D score %s' % (out['g_mask'].min(), out['g_mask'].max(), out['d_eval_score'][0]))


GENS = {'features': gen_features, 'intel': gen_intel, 'model': gen_model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', nargs='*')
    a = ap.parse_args()
    install_import_stubs()
    sys.path.insert(0, a.ref)
    sys.path.insert(0, os.path.join(a.ref, 'pyHASPI'))
    try:
        from make_golden_haspi import gen_haspi, gen_haspi_hl, gen_haspi_quality
        GENS['haspi'] = gen_haspi
        GENS['haspi_quality'] = gen_haspi_quality
        GENS['haspi_hl'] = gen_haspi_hl
    except ImportError:
        pass
    for name, fn in GENS.items():
        if a.only and name not in a.only:
            continue
        fn(a.ref)


if __name__ == '__main__':
    main()
