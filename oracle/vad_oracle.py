"""CPU ORACLE for the Silero VAD gate - TEST INFRASTRUCTURE ONLY.

PINNED: tests/test_vad_host.py checks this restatement against per-window probabilities, final LSTM state and
iterator events that the REFERENCE produced with its own vendored checkpoint
(whisperlivekit/silero_vad_models/silero_vad.jit through load_jit_vad / FixedVADIterator,
silero_vad_iterator.py:163-319; fixtures by scripts/gen_golden_vad.py).

The 16 kHz network inside the TorchScript archive (structure read from its code objects):
  x = [64 context samples | 512 new samples] -> reflect-pad 64 on the right -> conv1d with a fixed [258, 1, 256]
  DFT basis, stride 128 -> 4 frames x (129 re | 129 im) -> magnitude [129, 4]
  -> 4 x (Conv1d k=3 p=1 + ReLU): 129->128 s1, 128->64 s2, 64->64 s2, 64->128 s1 -> [128, 1]
  -> LSTMCell(128, 128) carrying (h, c) across windows -> ReLU -> Conv1d(128, 1, 1) -> sigmoid -> mean over time.
"""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

CONTEXT = 64
WINDOW = 512


def window_forward(sd: Dict[str, torch.Tensor], x1: torch.Tensor, h: torch.Tensor, c: torch.Tensor):
    """One 576-sample window (context + 512) -> (speech probability, h', c')."""
    x = F.pad(x1.view(1, 1, -1), (0, 64), mode="reflect")
    spec = F.conv1d(x, sd["stft.forward_basis_buffer"], stride=128)
    mag = torch.sqrt(spec[:, :129] ** 2 + spec[:, 129:] ** 2)
    y = mag
    for i, stride in enumerate((1, 2, 2, 1)):
        y = F.relu(F.conv1d(y, sd[f"encoder.{i}.reparam_conv.weight"], sd[f"encoder.{i}.reparam_conv.bias"],
                            stride=stride, padding=1))
    feat = y[:, :, 0]
    gates = F.linear(feat, sd["decoder.rnn.weight_ih"], sd["decoder.rnn.bias_ih"]) + \
        F.linear(h, sd["decoder.rnn.weight_hh"], sd["decoder.rnn.bias_hh"])
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    logit = F.conv1d(F.relu(h2).unsqueeze(-1), sd["decoder.decoder.2.weight"], sd["decoder.decoder.2.bias"])
    return torch.sigmoid(logit).mean(), h2, c2


class OracleSileroVAD:
    """The TorchScript wrapper's state handling (its forward(): context of 64 samples, LSTM state, reset_states)."""

    def __init__(self, sd):
        self.sd = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
        self.reset_states()

    def reset_states(self):
        self.h = torch.zeros(1, 128)
        self.c = torch.zeros(1, 128)
        self.context = torch.zeros(CONTEXT)

    @torch.no_grad()
    def __call__(self, x, sr: int = 16000):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32).reshape(-1)
        if sr != 16000 or x.shape[0] != WINDOW:
            raise ValueError("the oracle covers the 16 kHz model: 512 samples per call")
        x1 = torch.cat([self.context, x])
        p, self.h, self.c = window_forward(self.sd, x1, self.h, self.c)
        self.context = x1[-CONTEXT:]
        return p


class OracleVADIterator:
    """VADIterator + FixedVADIterator (silero_vad_iterator.py:186-319): buffers ragged input into 512-sample
    windows; start when prob >= threshold, end after min_silence of prob < threshold - 0.15."""

    def __init__(self, model, threshold=0.5, sampling_rate=16000, min_silence_duration_ms=100, speech_pad_ms=30):
        self.model, self.threshold = model, threshold
        self.min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
        self.speech_pad_samples = sampling_rate * speech_pad_ms / 1000
        self.reset_states()

    def reset_states(self):
        self.model.reset_states()
        self.triggered, self.temp_end, self.current_sample = False, 0, 0
        self.buffer = np.array([], dtype=np.float32)

    def _window(self, x) -> Optional[dict]:
        self.current_sample += WINDOW
        p = float(self.model(x, 16000))
        if p >= self.threshold and self.temp_end:
            self.temp_end = 0
        if p >= self.threshold and not self.triggered:
            self.triggered = True
            return {"start": int(max(0, self.current_sample - self.speech_pad_samples - WINDOW))}
        if p < self.threshold - 0.15 and self.triggered:
            if not self.temp_end:
                self.temp_end = self.current_sample
            if self.current_sample - self.temp_end < self.min_silence_samples:
                return None
            end = self.temp_end + self.speech_pad_samples - WINDOW
            self.temp_end, self.triggered = 0, False
            return {"end": int(end)}
        return None

    def __call__(self, x) -> List[dict]:
        self.buffer = np.append(self.buffer, np.asarray(x, np.float32))
        events = []
        while len(self.buffer) >= WINDOW:
            r = self._window(self.buffer[:WINDOW])
            self.buffer = self.buffer[WINDOW:]
            if r is not None:
                events.append(r)
        return events
