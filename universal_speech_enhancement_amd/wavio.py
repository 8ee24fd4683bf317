"""WAV files and the loader's FFT resampling through the library's host functions (``use_wav_read`` / ``use_wav_write`` /
``use_resample_fft`` / ``use_load_utterance``, include/use_hip.h): the reference's ``sf.read`` -> first channel ->
``librosa.resample(res_type="fft")`` -> peak normalisation (``src/data/components/loadwav_dataset.py:90-120``) and
``sf.write`` (``src/models/SGMSE_module.py:80``) without soundfile / librosa / scipy."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._lib import check, lib

PCM16, FLOAT32 = 0, 1


def read_wav(path: str):
    """-> (float64 array [frames] or [frames, channels], sample rate), scaled like ``soundfile.read``."""
    p, n, ch, sr = C.POINTER(C.c_double)(), C.c_int64(), C.c_int(), C.c_int()
    check(lib().use_wav_read(os.fsencode(path), C.byref(p), C.byref(n), C.byref(ch), C.byref(sr)), "use_wav_read")
    try:
        a = np.ctypeslib.as_array(p, shape=(n.value * ch.value,)).copy() if n.value else np.zeros(0)
    finally:
        lib().use_free(p)
    return (a if ch.value == 1 else a.reshape(n.value, ch.value)), sr.value


def write_wav(path: str, wav: np.ndarray, sample_rate: int, subtype: int = PCM16) -> None:
    """``soundfile.write(path, wav, sample_rate)``: 16-bit PCM by default (soundfile's default WAV subtype), or 32-bit float."""
    a = np.ascontiguousarray(wav, dtype=np.float32)
    frames, ch = (a.shape[0], 1) if a.ndim == 1 else a.shape
    check(lib().use_wav_write(os.fsencode(path), a.ctypes.data_as(C.c_void_p), frames, ch, int(sample_rate), subtype), "use_wav_write")


def resample_fft(x: np.ndarray, num: int) -> np.ndarray:
    """``scipy.signal.resample(x, num)`` for a real 1-D signal (float64)."""
    a = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(int(num), np.float64)
    check(lib().use_resample_fft(a.ctypes.data_as(C.c_void_p), a.shape[0], int(num), y.ctypes.data_as(C.c_void_p)), "use_resample_fft")
    return y


def load_utterance(path: str, sampling_rate: int = 24000, normalize: bool = True):
    """One item of the reference's inference dataset: -> (float32 [L], sample rate)."""
    p, n, sr = C.POINTER(C.c_float)(), C.c_int64(), C.c_int()
    check(lib().use_load_utterance(os.fsencode(path), int(sampling_rate or 0), int(bool(normalize)), C.byref(p), C.byref(n), C.byref(sr)),
          "use_load_utterance")
    try:
        a = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    finally:
        lib().use_free(p)
    return a, sr.value
