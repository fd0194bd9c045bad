"""CPU ORACLE for the diarization front end (a12) - TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic restated here lives in NeMo (nemo-toolkit[asr] >=3,<4, pyproject.toml:80-83 of
the reference; exact pin unknown, uv.lock is not in the tree), which is not installed where this code is
built and measured, and the reference's own tests hold no numeric vectors for it
(tests/test_sortformer_real_fixture.py is statistical and skipped without NeMo).  What follows restates
the published algorithm of ``nemo.collections.asr.parts.preprocessing.features.FilterbankFeatures`` as
configured at whisperlivekit/diarization/sortformer_backend.py:181-187:
  window 25 ms (400 samples, symmetric hann), stride 10 ms, n_fft 512, 128 slaney mel bins 0-8000 Hz,
  pre-emphasis 0.97, centred STFT with zero padding, power 2, log(x + 2^-24), normalize "NA", pad_to 0.
"""
import numpy as np
import torch


def nemo_log_mel(pcm: np.ndarray, filters: np.ndarray, n_fft: int = 512, win_length: int = 400, hop: int = 160,
                 preemph: float = 0.97, log_guard: float = 2.0 ** -24) -> np.ndarray:
    """-> [n_frames, n_mels], n_frames = len(pcm) // hop + 1 (FilterbankFeatures.get_seq_len)."""
    x = torch.from_numpy(np.asarray(pcm, np.float32)).unsqueeze(0)
    x = torch.cat((x[:, :1], x[:, 1:] - preemph * x[:, :-1]), dim=1)
    window = torch.hann_window(win_length, periodic=False)
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=win_length, center=True, window=window,
                      return_complex=True, pad_mode="constant")
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1))
    power = mag.pow(2.0)
    mel = torch.matmul(torch.from_numpy(np.asarray(filters, np.float32)), power)
    out = torch.log(mel + log_guard)
    n_frames = x.shape[1] // hop + 1
    return out[0, :, :n_frames].transpose(0, 1).contiguous().numpy()
