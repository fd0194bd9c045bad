"""Slaney-style mel filterbank, generated instead of shipped.

The reference loads ``assets/mel_filters.npz`` (whisperlivekit/whisper/audio.py:91-107), whose
docstring states how it was made: ``librosa.filters.mel(sr=16000, n_fft=400, n_mels=80|128)``.
librosa is not a dependency here, so this module restates that published construction
(Slaney mel scale: linear below 1 kHz, logarithmic above; triangular filters normalised to
unit area, i.e. ``norm="slaney"``) and ``tests/test_oracle_golden.py`` pins the result against
a digest of the reference's file.
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np

from .dims import N_FFT, SAMPLE_RATE

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    return np.where(log_t, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-30) / _MIN_LOG_HZ) / _LOGSTEP, mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = _F_SP * m
    log_t = m >= _MIN_LOG_MEL
    return np.where(log_t, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), f)


@lru_cache(maxsize=None)
def mel_filterbank(n_mels: int, sr: int = SAMPLE_RATE, n_fft: int = N_FFT) -> np.ndarray:
    """float32 [n_mels, n_fft//2+1] filterbank."""
    n_freq = n_fft // 2 + 1
    fft_freqs = np.linspace(0.0, sr / 2.0, n_freq)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    weights = np.zeros((n_mels, n_freq), dtype=np.float32)  # librosa rounds here, then scales
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    weights *= enorm[:, None]
    out = weights
    out.setflags(write=False)
    return out
