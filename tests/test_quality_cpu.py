"""Host side of the Discriminator_Quality targets (nele_gan_amd/quality.py): the reference's maps, file conventions and its ViSQOL
command line, with STAND-IN programs in place of pypesq / the ViSQOL binary (neither is in the image; nothing here claims to be one).
Reference: intel.py:142-160, audio_util.py:205-265, 323-364."""
import os
import stat
import sys

import numpy as np
import pytest

from nele_gan_amd import dataio, quality


@pytest.fixture(autouse=True)
def _clean_registry():
    quality.clear_backends()
    yield
    quality.clear_backends()


def _fake_pesq(ref, deg, fs):
    """a deterministic function of both signals and the rate: 1 + 3.5 * normalised correlation, as a stand-in score in PESQ's range"""
    assert fs == 16000 and len(ref) == len(deg)
    r, d = np.asarray(ref, dtype=np.float64), np.asarray(deg, dtype=np.float64)
    return 1.0 + 3.5 * abs(float(r @ d)) / (np.sqrt(float(r @ r) * float(d @ d)) + 1e-30)


_FAKE_VISQOL = r'''#!%s
import csv, struct, sys
a = sys.argv[1:]
assert '--use_speech_mode' in a
model = a[a.index('--similarity_to_quality_model') + 1]
inp = a[a.index('--batch_input_csv') + 1]
res = a[a.index('--results_csv') + 1]
assert open(model).read().startswith('svm')
def rd(p):
    b = open(p, 'rb').read()
    n = (len(b) - 44) // 2
    return struct.unpack('<%%dh' %% n, b[44:44 + 2 * n])
rows = list(csv.DictReader(open(inp)))
with open(res, 'w') as f:
    f.write('reference,degraded,moslqo\n')
    for r in rows:
        x, y = rd(r['reference']), rd(r['degraded'])
        n = min(len(x), len(y))
        s = sum(abs(p - q) for p, q in zip(x[:n], y[:n])) / (n * 32768.0)
        f.write('%%s,%%s,%%.12f\n' %% (r['reference'], r['degraded'], 1.0 + 4.0 / (1.0 + 50.0 * s)))
''' % sys.executable


def _visqol_of(x, y):
    xi = np.round(np.asarray(x, dtype=np.float64) * 32768.0)
    yi = np.round(np.asarray(y, dtype=np.float64) * 32768.0)
    n = min(len(xi), len(yi))
    s = float(np.abs(xi[:n] - yi[:n]).sum()) / (n * 32768.0)
    return 1.0 + 4.0 / (1.0 + 50.0 * s)


def _program(tmp_path):
    prog = tmp_path / 'visqol'
    prog.write_text(_FAKE_VISQOL)
    prog.chmod(prog.stat().st_mode | stat.S_IEXEC)
    model = tmp_path / 'model.txt'
    model.write_text('svm_type nu_svr\n')
    return quality.visqol_program(str(prog), str(model))


def _pcm(rng, n):
    return (np.round(rng.standard_normal(n) * 3000.0).clip(-32768, 32767) / 32768.0).astype(np.float32)


def _corpus(tmp_path):
    rng = np.random.default_rng(5)
    clean_root = str(tmp_path / 'Clean') + '/'
    out = str(tmp_path / 'out')
    drc = str(tmp_path / 'MultiEnh')
    for d in (clean_root, out, drc):
        os.makedirs(d)
    clean, gen, pre = {}, {}, {}
    for name, n in (('p226_001', 24000), ('p226_002', 19200), ('x', 16000)):
        clean[name] = _pcm(rng, n)                                         # values k / 32768: written and read back exactly
        dataio.write_wav_pcm16(clean_root + name + '.wav', clean[name], quantised=True)
        m = 256 * (n // 256)                                               # a generated example: 256 * (L // 256) samples, '<name>@<epoch>.wav'
        gen[name] = (clean[name][:m] + _pcm(rng, m) / 4).astype(np.float32)
        gen[name] = (np.round(gen[name] * 32768.0) / 32768.0).astype(np.float32)
        dataio.write_wav_pcm16('%s/%s@7.wav' % (out, name), gen[name], quantised=True)
        pre[name] = (np.round(clean[name][:n - 100] * 16384.0) / 32768.0).astype(np.float32)   # a pre-enhanced example: the clean file's own name
        dataio.write_wav_pcm16('%s/%s.wav' % (drc, name), pre[name], quantised=True)
        for arr, path in ((clean[name], clean_root + name + '.wav'), (gen[name], '%s/%s@7.wav' % (out, name)), (pre[name], '%s/%s.wav' % (drc, name))):
            np.testing.assert_array_equal(dataio.load(path, sr=16000)[0], arr)
    return clean_root, out, drc, clean, gen, pre


def test_maps_are_the_references_logistic_functions():
    assert quality.mapping_PESQ_harvard(2.5) == pytest.approx(0.5)
    assert quality.mapping_PESQ_harvard(4.5) == pytest.approx(1.0 / (1.0 + np.exp(-3.0)))
    assert quality.mapping_PESQ_harvard(-0.5) == pytest.approx(1.0 / (1.0 + np.exp(4.5)))
    assert quality.mapping_VISQOL(2.2) == pytest.approx(0.5)
    assert quality.mapping_VISQOL(5.0) == pytest.approx(1.0 / (1.0 + np.exp(-7.0)))
    x = np.linspace(1.0, 4.5, 8)
    np.testing.assert_allclose(quality.mapping_PESQ_harvard(x), 1 / (1 + np.exp(-1.5 * (x - 2.5))), rtol=0, atol=0)


def test_nothing_is_scored_without_the_external_programs(tmp_path):
    clean_root, out, drc, *_ = _corpus(tmp_path)
    files = [out + '/p226_001@7.wav']
    with pytest.raises(quality.QualityBackendMissing, match='PESQ'):
        quality.read_batch_PESQ(clean_root, files)
    with pytest.raises(quality.QualityBackendMissing, match='ViSQOL'):
        quality.read_batch_VISQOL(clean_root, files)
    with pytest.raises(quality.QualityBackendMissing):
        quality.Scorer().raw([np.zeros(8, np.float32)], [np.zeros(8, np.float32)])
    with pytest.raises(quality.QualityBackendMissing):
        quality.PESQ_Wrapper_harvard(np.zeros(8), np.zeros(8), 16000)


def test_pesq_file_fan_out_follows_the_references_names_and_lengths(tmp_path):
    clean_root, out, drc, clean, gen, pre = _corpus(tmp_path)
    seen = []

    def pesq(ref, deg, fs):
        seen.append((np.array(ref), np.array(deg)))
        return _fake_pesq(ref, deg, fs)
    quality.set_backends(pesq=pesq)
    names = ['p226_002', 'x', 'p226_001']
    files = ['%s/%s@7.wav' % (out, n) for n in names]
    raw = quality.read_batch_PESQ(clean_root, files, norm=False)
    want = [_fake_pesq(clean[n][:len(gen[n])], gen[n], 16000) for n in names]
    assert raw == pytest.approx(want, abs=1e-12)                           # list order, '<name>@<epoch>.wav' -> '<name>.wav', cut to the shorter
    assert all(len(r) == len(d) for r, d in seen)
    mapped = quality.read_batch_PESQ(clean_root, files)
    assert mapped == pytest.approx([float(quality.mapping_PESQ_harvard(v)) for v in want], abs=1e-12)
    assert quality.read_PESQ(clean_root, files[1], True) == pytest.approx(mapped[1], abs=1e-12)
    assert quality.read_PESQ(clean_root, files[1], False) == pytest.approx(raw[1], abs=1e-12)
    # pre-enhanced examples: the file carries the clean file's name; always mapped
    dfiles = ['%s/%s.wav' % (drc, n) for n in names]
    got = quality.read_batch_PESQ_DRC(clean_root, dfiles)
    want_d = [float(quality.mapping_PESQ_harvard(_fake_pesq(clean[n][:len(pre[n])], pre[n], 16000))) for n in names]
    assert got == pytest.approx(want_d, abs=1e-12)
    assert quality.read_PESQ_DRC(clean_root, dfiles[0]) == pytest.approx(want_d[0], abs=1e-12)
    # a caller with its own pool takes every pair of a call at once
    calls = []
    quality.set_backends(pesq_batch=lambda refs, degs, fs: calls.append(len(refs)) or [_fake_pesq(r, d, fs) for r, d in zip(refs, degs)])
    assert quality.read_batch_PESQ(clean_root, files, norm=False) == pytest.approx(want, abs=1e-12) and calls == [3]


def test_visqol_runs_the_references_command_line_on_the_paths(tmp_path):
    clean_root, out, drc, clean, gen, pre = _corpus(tmp_path)
    quality.set_backends(visqol=_program(tmp_path))
    names = ['x', 'p226_001']
    files = ['%s/%s@7.wav' % (out, n) for n in names]
    raw = quality.read_batch_VISQOL(clean_root, files, norm=False)
    want = [_visqol_of(clean[n], gen[n]) for n in names]
    assert raw == pytest.approx(want, abs=1e-9)
    assert quality.read_batch_VISQOL(clean_root, files) == pytest.approx([float(quality.mapping_VISQOL(v)) for v in want], abs=1e-9)
    dfiles = ['%s/%s.wav' % (drc, n) for n in names]
    want_d = [float(quality.mapping_VISQOL(_visqol_of(clean[n], pre[n]))) for n in names]
    assert quality.read_batch_VISQOL_DRC(clean_root, dfiles) == pytest.approx(want_d, abs=1e-9)
    # the program's failure is the caller's failure (audio_util.py:247 `assert ret==0`)
    bad = quality.visqol_program(str(tmp_path / 'visqol'), str(tmp_path / 'no-such-model'))
    with pytest.raises(RuntimeError, match='ViSQOL exited'):
        bad([(clean_root + 'x.wav', files[0])])


def test_scorer_scores_a_batch_in_memory_like_the_files_would_be(tmp_path):
    clean_root, out, drc, clean, gen, pre = _corpus(tmp_path)
    quality.set_backends(pesq=_fake_pesq, visqol=_program(tmp_path))
    names = ['p226_001', 'p226_002', 'x']
    refs = [clean[n][:len(gen[n])] for n in names]
    degs = [gen[n] for n in names]
    sc = quality.Scorer(tmp_root=str(tmp_path))
    raw = sc.raw(refs, degs)
    files = ['%s/%s@7.wav' % (out, n) for n in names]
    np.testing.assert_allclose(raw[:, 0], quality.read_batch_PESQ(clean_root, files, norm=False), rtol=0, atol=1e-12)
    # ViSQOL sees wav files of the same samples (the clean rows cut like PESQ's: the stand-in program compares the common part anyway)
    np.testing.assert_allclose(raw[:, 1], quality.read_batch_VISQOL(clean_root, files, norm=False), rtol=0, atol=1e-9)
    m = sc.mapped(refs, degs)
    np.testing.assert_allclose(m[:, 0], quality.mapping_PESQ_harvard(raw[:, 0]))
    np.testing.assert_allclose(m[:, 1], quality.mapping_VISQOL(raw[:, 1]))
    assert sc.calls == 2 and sc.pairs == 6
    assert not [p for p in os.listdir(str(tmp_path)) if p.startswith('nele-quality-')]         # the call's wav files are gone
    # files already on disk can be named instead
    raw2 = sc.raw(refs, degs, files=[(clean_root + n + '.wav', f) for n, f in zip(names, files)])
    np.testing.assert_allclose(raw2, raw, rtol=0, atol=1e-9)
    assert quality.Scorer(use=('pesq',)).raw(refs, degs)[:, 1].tolist() == [0.0, 0.0, 0.0]
    assert sc.raw([], []).shape == (0, 2)


def test_the_module_entry_points_take_the_programs_from_the_command_line(tmp_path):
    import argparse
    _program(tmp_path)
    ap = argparse.ArgumentParser()
    quality.add_cli_arguments(ap)
    a = ap.parse_args(['--pesq', 'math:hypot', '--visqol', str(tmp_path / 'visqol'), '--visqol-model', str(tmp_path / 'model.txt')])
    assert quality.backends_from_cli(a)
    import math
    assert quality._BACKENDS['pesq'] is math.hypot and callable(quality._BACKENDS['visqol'])
    quality.clear_backends()
    assert not quality.backends_from_cli(ap.parse_args([]))
    with pytest.raises(SystemExit):
        quality.backends_from_cli(ap.parse_args(['--visqol', 'x']))
