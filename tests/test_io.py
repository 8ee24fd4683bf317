"""the library's host-side wire formats (csrc/use_io.cpp) -- WAV in / out and the loader's FFT resampling --
against scipy (scipy.signal.resample is what librosa.resample(res_type="fft") of the reference's loader calls,
src/data/components/loadwav_dataset.py:95-98; scipy.io.wavfile as an independent WAV codec).  CPU only: no device is touched."""
import struct

import numpy as np
import pytest

from universal_speech_enhancement_amd import wavio
from universal_speech_enhancement_amd._lib import UseHipError


@pytest.mark.parametrize("n,num", [(48000, 24000), (44100, 24000), (16000, 24000), (6400, 9600), (9601, 4801), (4801, 9601),
                                   (1000, 999), (999, 1000), (997, 1499), (1024, 512), (512, 2048), (7, 7), (2, 5), (5, 2)])
def test_resample_fft_equals_scipy(n, num):
    from scipy.signal import resample
    x = np.random.RandomState(n + num).randn(n)
    want = resample(x, num)
    got = wavio.resample_fft(x, num)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * max(1.0, float(np.abs(want).max())))


def _write_raw(path, fmt_tag, channels, rate, bits, payload, extensible=False):
    align = channels * bits // 8
    if extensible:
        guid = struct.pack("<H", fmt_tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, rate, rate * align, align, bits, 22, bits, 0) + guid
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * align, align, bits)
    junk = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"            # odd-sized chunk before the data (padded)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + junk + b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_read_sample_formats(tmp_path):
    from scipy.io import wavfile
    rng = np.random.RandomState(0)
    i16 = (rng.randn(1000, 2) * 8000).astype(np.int16)
    wavfile.write(str(tmp_path / "i16.wav"), 48000, i16)
    a, sr = wavio.read_wav(str(tmp_path / "i16.wav"))
    assert sr == 48000 and a.shape == (1000, 2)
    np.testing.assert_array_equal(a, i16.astype(np.float64) / 32768.0)                   # soundfile scaling
    f32 = (rng.randn(777) * 0.3).astype(np.float32)
    wavfile.write(str(tmp_path / "f32.wav"), 24000, f32)
    a, sr = wavio.read_wav(str(tmp_path / "f32.wav"))
    assert sr == 24000 and a.shape == (777,)
    np.testing.assert_array_equal(a, f32.astype(np.float64))
    i32 = (rng.randn(300) * 1e8).astype(np.int32)
    wavfile.write(str(tmp_path / "i32.wav"), 16000, i32)
    np.testing.assert_array_equal(wavio.read_wav(str(tmp_path / "i32.wav"))[0], i32.astype(np.float64) / 2147483648.0)
    u8 = rng.randint(0, 256, 500).astype(np.uint8)
    wavfile.write(str(tmp_path / "u8.wav"), 8000, u8)
    np.testing.assert_array_equal(wavio.read_wav(str(tmp_path / "u8.wav"))[0], (u8.astype(np.float64) - 128.0) / 128.0)
    f64 = rng.randn(100)
    wavfile.write(str(tmp_path / "f64.wav"), 24000, f64)
    np.testing.assert_array_equal(wavio.read_wav(str(tmp_path / "f64.wav"))[0], f64)
    # 24-bit PCM, WAVE_FORMAT_EXTENSIBLE, an odd-sized chunk in front of the data
    i24 = rng.randint(-2 ** 23, 2 ** 23, 200)
    payload = b"".join(struct.pack("<i", int(v))[:3] for v in i24)
    _write_raw(str(tmp_path / "i24.wav"), 1, 1, 44100, 24, payload, extensible=True)
    a, sr = wavio.read_wav(str(tmp_path / "i24.wav"))
    assert sr == 44100
    np.testing.assert_array_equal(a, i24.astype(np.float64) / 8388608.0)


def test_wav_write_subtypes(tmp_path):
    from scipy.io import wavfile
    x = np.concatenate([np.random.RandomState(1).randn(4000) * 0.3, [1.0, -1.0, 1.5, -1.5, 0.5 / 32767, 1.5 / 32767]]).astype(np.float32)
    wavio.write_wav(str(tmp_path / "p.wav"), x, 24000)                                # soundfile's default subtype: PCM_16
    sr, a = wavfile.read(str(tmp_path / "p.wav"))
    assert sr == 24000 and a.dtype == np.int16
    np.testing.assert_array_equal(a, np.clip(np.rint(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16))
    wavio.write_wav(str(tmp_path / "f.wav"), x, 16000, wavio.FLOAT32)
    sr, a = wavfile.read(str(tmp_path / "f.wav"))
    assert sr == 16000 and a.dtype == np.float32
    np.testing.assert_array_equal(a, x)
    st = np.stack([x, -x], 1)
    wavio.write_wav(str(tmp_path / "s.wav"), st, 24000, wavio.FLOAT32)
    np.testing.assert_array_equal(wavfile.read(str(tmp_path / "s.wav"))[1], st)
    np.testing.assert_array_equal(wavio.read_wav(str(tmp_path / "s.wav"))[0], st.astype(np.float64))


@pytest.mark.parametrize("rate,n", [(48000, 9000), (16000, 6401), (44100, 4410), (44100, 147 * 37), (22050, 147 * 20), (24000, 3000)])
def test_load_utterance_is_the_reference_loader(tmp_path, rate, n):
    """sf.read -> first channel -> librosa fft resampling -> peak normalisation, float64, then float32 (loadwav_dataset.py:90-120)."""
    from scipy.io import wavfile
    from scipy.signal import resample
    pcm = (np.random.RandomState(rate).randn(n, 2) * 5000).astype(np.int16)
    wavfile.write(str(tmp_path / "x.wav"), rate, pcm)
    x = pcm[:, 0].astype(np.float64) / 32768.0
    if rate != 24000:
        x = resample(x, int(np.ceil(len(x) * (24000 / rate))))     # librosa's order: the ratio is rounded first (147 * k samples -> one more)
    want = (x / np.max(np.abs(x)) * 0.8).astype(np.float32)
    got, sr = wavio.load_utterance(str(tmp_path / "x.wav"), 24000, True)
    assert sr == 24000 and got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)
    raw, sr = wavio.load_utterance(str(tmp_path / "x.wav"), 0, False)                # sampling_rate=None, normalize=False
    assert sr == rate
    np.testing.assert_array_equal(raw, (pcm[:, 0].astype(np.float64) / 32768.0).astype(np.float32))


def test_wav_errors_are_reported(tmp_path):
    with pytest.raises(UseHipError, match="cannot open"):
        wavio.read_wav(str(tmp_path / "missing.wav"))
    (tmp_path / "bad.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(UseHipError, match="RIFF"):
        wavio.load_utterance(str(tmp_path / "bad.wav"))
    _write_raw(str(tmp_path / "adpcm.wav"), 2, 1, 8000, 4, b"\0" * 64)
    with pytest.raises(UseHipError):
        wavio.read_wav(str(tmp_path / "adpcm.wav"))
