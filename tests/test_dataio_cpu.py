"""CPU: the on-disk hand-off formats (wav PCM_16, name@epoch convention, score lists) of nele_gan_amd.dataio."""
import os
import struct
import wave

import numpy as np
import pytest

from nele_gan_amd import dataio

HERE = os.path.dirname(__file__)
TOY = os.path.join(HERE, 'golden', 'toy')


@pytest.mark.parametrize('name', ['Train_Clean.wav', 'Train_Noise.wav', 'Test_Clean.wav'])
def test_read_wav_matches_the_stdlib_parser_on_the_reference_toy_files(name):
    x, sr = dataio.read_wav(os.path.join(TOY, name))
    w = wave.open(os.path.join(TOY, name))
    ref = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
    assert sr == w.getframerate() == 16000 and x.dtype == np.float32
    assert np.array_equal(x, ref)


def test_pcm16_write_read_is_the_oracle_round_trip(tmp_path):
    from oracle.step import pcm16_roundtrip
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(5000) * 0.2).astype(np.float32)
    x[:6] = [0.5 / 32767, 1.5 / 32767, 2.5 / 32767, -0.5 / 32767, 1.2, -1.2]       # ties round to even; overload saturates
    p = str(tmp_path / 'a.wav')
    dataio.write_wav_pcm16(p, x)
    y, sr = dataio.read_wav(p)
    assert sr == 16000 and np.array_equal(y, pcm16_roundtrip(x))
    assert list(np.rint(y[:4] * 32768)) == [0, 2, 2, 0]
    # a signal that already went through the device-side emulation is stored without a second rounding
    p2 = str(tmp_path / 'b.wav')
    dataio.write_wav_pcm16(p2, y, quantised=True)
    assert np.array_equal(dataio.read_wav(p2)[0], y)
    with open(p, 'rb') as f:
        head = f.read(44)
    assert head[:4] == b'RIFF' and struct.unpack('<HHI', head[20:28]) == (1, 1, 16000) and struct.unpack('<H', head[34:36])[0] == 16


def test_other_sample_formats_and_stereo(tmp_path):
    v = np.array([0.25, -0.5, 0.75, -1.0], dtype=np.float64)

    def riff(tag, bits, nch, body):
        fmt = struct.pack('<HHIIHH', tag, nch, 16000, 16000 * nch * bits // 8, nch * bits // 8, bits)
        return b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<I', 16) + fmt + b'data' + struct.pack('<I', len(body)) + body
    cases = {
        'f32.wav': riff(3, 32, 1, v.astype('<f4').tobytes()),
        'i32.wav': riff(1, 32, 1, (v * 2 ** 31).clip(-2 ** 31, 2 ** 31 - 1).astype('<i4').tobytes()),
        'i24.wav': riff(1, 24, 1, b''.join(struct.pack('<i', int(s * 2 ** 23))[:3] for s in v.clip(-1, 1 - 2 ** -23))),
        'st16.wav': riff(1, 16, 2, np.stack([v, v], 1).reshape(-1).__mul__(32768).clip(-32768, 32767).astype('<i2').tobytes()),
    }
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        x, sr = dataio.read_wav(str(p))
        assert sr == 16000 and x.shape == (4,)
        np.testing.assert_allclose(x[:3], v[:3], atol=2e-7)
    with pytest.raises(ValueError):
        (tmp_path / 'bad.wav').write_bytes(b'RIFX' + b'\0' * 40)
        dataio.read_wav(str(tmp_path / 'bad.wav'))
    with pytest.raises(ValueError):
        dataio.load(os.path.join(TOY, 'Train_Clean.wav'), sr=8000)


def test_name_convention_and_score_lists():
    assert dataio.wave_name_of('/out/epoch3/Train_epoch3/Train_Clean@3.wav') == 'Train_Clean'       # audio_util.py:121-126
    assert dataio.wave_name_of('/data/Train_Clean.wav') == 'Train_Clean'
    assert dataio.enhanced_name('/out/temp', 'Train_Clean.wav', 12) == '/out/temp/Train_Clean@12.wav'  # train_nele.py:312
    s = [0.25, 0.5], [0.125, 1.0], [0.75, 0.0], [0.0, 0.0], [0.0, 0.0]
    paths = ['/o/a@1.wav', '/o/b@1.wav']
    lines = dataio.List_concat(dataio.List_concat_5scores(*s), paths)
    assert lines[0] == '0.25,0.125,0.75,0.0,0.0,/o/a@1.wav'
    intel, qua, path = dataio.parse_score_line(lines[1])
    assert list(intel) == [0.5, 1.0, 0.0] and list(qua) == [0.0, 0.0] and path == '/o/b@1.wav'
    assert dataio.List_concat_3scores([1], [2], [3]) == ['1,2,3'] and dataio.List_concat_score([1], [2]) == ['1,2']
    with pytest.raises(ValueError):
        dataio.parse_score_line('0.5,/o/a.wav')


def test_listread_and_get_filepaths(tmp_path):
    (tmp_path / 'sub').mkdir()
    for n in ('x.wav', 'sub/y.wav', 'z.txt'):
        (tmp_path / n).write_bytes(b'')
    assert sorted(os.path.basename(p) for p in dataio.get_filepaths(str(tmp_path))) == ['x.wav', 'y.wav']
    lst = tmp_path / 'list.txt'
    lst.write_text('a.wav\nb.wav\n')
    assert dataio.ListRead(str(lst)) == ['a.wav', 'b.wav']
    dataio.creatdir(str(tmp_path / 'p' / 'q'))
    assert (tmp_path / 'p' / 'q').is_dir()


def test_batch_reader_and_writer_move_the_same_samples_as_the_per_file_functions(tmp_path):
    """nele_wav_read_pcm16_batch / nele_wav_write_pcm16_batch (host-only entry points of the library, several library threads): rows equal
    the per-file reader's samples x 32768, zeros behind them, cut at the row length; other wav flavours are reported (-1), unreadable
    files too (-2); written files are byte-identical to write_wav_pcm16 of the same samples."""
    names = ['Train_Clean.wav', 'Train_Noise.wav', 'Test_Clean.wav']
    paths = [os.path.join(TOY, n) for n in names]
    f32 = tmp_path / 'f32.wav'
    body = np.array([0.25, -0.5], dtype='<f4').tobytes()
    f32.write_bytes(b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 3, 1, 16000, 64000, 4, 32) + b'data'
                    + struct.pack('<I', len(body)) + body)
    allp = paths + [str(f32), str(tmp_path / 'missing.wav')]
    ref = [dataio.read_wav(p)[0] for p in paths]
    for threads in (1, 3):
        cap = max(len(r) for r in ref) + 100
        rows = np.full((len(allp) + 1, cap), 7, dtype=np.int16)
        got, sr = dataio.read_wav_batch_pcm16(allp, rows, threads=threads)
        assert list(got) == [len(r) for r in ref] + [-1, -2] and list(sr[:3]) == [16000] * 3
        for r, x in enumerate(ref):
            assert np.array_equal(rows[r, :len(x)].astype(np.float32) / 32768.0, x) and not rows[r, len(x):].any()
        assert not rows[3].any() and not rows[4].any() and (rows[5] == 7).all()          # failed rows are zeroed, rows beyond n untouched
    short = np.zeros((3, 1000), dtype=np.int16)
    got, _ = dataio.read_wav_batch_pcm16(paths, short, threads=2)                        # rows shorter than the files: cut, not overrun
    assert list(got) == [1000] * 3 and np.array_equal(short[1].astype(np.float32) / 32768.0, ref[1][:1000])
    # writer
    rng = np.random.default_rng(5)
    q = rng.integers(-32768, 32768, size=(4, 900)).astype(np.int16)
    ns = [900, 0, 257, 512]
    outs = [str(tmp_path / ('w%d.wav' % k)) for k in range(4)]
    dataio.write_wav_batch_pcm16(outs, q, ns, threads=3)
    for k, p in enumerate(outs):
        one = str(tmp_path / 'one.wav')
        dataio.write_wav_pcm16(one, q[k, :ns[k]].astype(np.float32) / 32768.0, quantised=True)
        assert open(p, 'rb').read() == open(one, 'rb').read()
    with pytest.raises(IOError):
        dataio.write_wav_batch_pcm16([str(tmp_path / 'no_such_dir' / 'a.wav')], q, [10])


def test_wav_headers_probed_in_one_library_call(tmp_path):
    """nele_wav_probe_pcm16_batch (dataio.probe_wav_batch_pcm16): the sample counts a loader needs for its padded batch shapes, from the RIFF
    headers of a whole batch in one foreign call (no GPU): -1 = another wav flavour, -2 = unreadable."""
    import numpy as np
    from nele_gan_amd import dataio
    rs = np.random.RandomState(0)
    paths, want = [], []
    for k, n in enumerate((1, 255, 16000, 40001)):
        p = str(tmp_path / ('a%d.wav' % k))
        dataio.write_wav_pcm16(p, (0.1 * rs.randn(n)).astype(np.float32))
        paths.append(p)
        want.append(n)
    junk = str(tmp_path / 'junk.wav')
    open(junk, 'wb').write(b'RIFF' + b'\\0' * 60)
    got = dataio.probe_wav_batch_pcm16(paths + [junk, str(tmp_path / 'missing.wav')], threads=3)
    assert got.tolist() == want + [-1, -2]
    assert dataio.probe_wav_batch_pcm16([], threads=1).tolist() == []
