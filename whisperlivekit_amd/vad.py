"""Silero VAD gate on the HIP backend (the step immediately before the simul_whisper path).

The reference loads ``whisperlivekit/silero_vad_models/silero_vad.jit`` (``load_jit_vad``,
silero_vad_iterator.py:163-184) and evaluates it on the CPU once per 512-sample window from
``VADIterator.__call__`` (:225-285), wrapped by ``FixedVADIterator`` (:288-319), which ``AudioProcessor`` calls on
every incoming PCM array (audio_processor.py:1189-1190).  Here:

* ``HipSileroVAD`` has the model's duck type (``model(x, sr)`` -> a scalar with ``.item()``, ``reset_states()``), so
  the reference's own ``FixedVADIterator(HipSileroVAD(...))`` works unmodified - one device call per window;
* ``HipFixedVADIterator`` is the same iterator with the windows of one call evaluated together (two launches per
  call instead of ~16 model evaluations per 0.5 s chunk); its events are identical.

Weights come from the reference's own archive (``load_silero_state_dict``: torch.jit.load on the host, tensors
only).  No CPU fallback: without libwlk_hip.so / a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from . import _lib

WINDOW = 512
CONTEXT = 64
SAMPLE_RATE = 16000


def find_silero_jit() -> str:
    """``WLK_SILERO_VAD_JIT`` or the file shipped inside an installed WhisperLiveKit."""
    env = os.environ.get("WLK_SILERO_VAD_JIT")
    if env:
        return env
    try:
        import importlib.util
        spec = importlib.util.find_spec("whisperlivekit")
        if spec and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "silero_vad_models", "silero_vad.jit")
            if os.path.exists(cand):
                return cand
    except Exception:
        pass
    raise FileNotFoundError("silero_vad.jit not found: set WLK_SILERO_VAD_JIT or install WhisperLiveKit")


def load_silero_state_dict(path: Optional[str] = None) -> Dict[str, np.ndarray]:
    """The 16 kHz sub-model's tensors of the TorchScript archive, as numpy (names without the ``_model.`` prefix)."""
    import torch
    model = torch.jit.load(path or find_silero_jit(), map_location="cpu")
    return {k[len("_model."):]: v.detach().to(torch.float32).numpy() for k, v in model.state_dict().items()
            if k.startswith("_model.")}


def pack_vad_weights(sd: Dict[str, np.ndarray]) -> np.ndarray:
    """Archive tensors -> the flat buffer of ``wlk_vad_tensor_lookup``: every matrix input-major (so that consecutive
    GPU threads read consecutive floats), Conv1d weights [out, in, tap] -> [(in, tap), out]."""
    lib = _lib.load()
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
    conv = lambda w: f32(f32(w).transpose(1, 2, 0).reshape(-1, w.shape[0]))
    named = {
        "stft.basisT": f32(f32(sd["stft.forward_basis_buffer"])[:, 0, :].T),
        "rnn.wihT": f32(f32(sd["decoder.rnn.weight_ih"]).T), "rnn.whhT": f32(f32(sd["decoder.rnn.weight_hh"]).T),
        "rnn.bih": f32(sd["decoder.rnn.bias_ih"]), "rnn.bhh": f32(sd["decoder.rnn.bias_hh"]),
        "dec.w": f32(sd["decoder.decoder.2.weight"]).reshape(-1), "dec.b": f32(sd["decoder.decoder.2.bias"]).reshape(-1),
    }
    for i in range(4):
        named[f"enc{i}.wT"] = conv(sd[f"encoder.{i}.reparam_conv.weight"])
        named[f"enc{i}.b"] = f32(sd[f"encoder.{i}.reparam_conv.bias"])
    total = C.c_uint64()
    _lib.check(lib.wlk_vad_weights_floats(C.byref(total)))
    flat = np.zeros(total.value, np.float32)
    i = 0
    while True:
        name = C.c_char_p()
        if lib.wlk_vad_tensor_name(i, C.byref(name)) != 0:
            break
        off, numel = C.c_uint64(), C.c_uint64()
        _lib.check(lib.wlk_vad_tensor_lookup(name.value, C.byref(off), C.byref(numel)))
        a = named[name.value.decode()].reshape(-1)
        if a.size != numel.value:
            raise ValueError(f"{name.value.decode()}: expected {numel.value} values, the archive gives {a.size} "
                             "(only the 16 kHz Silero VAD v5/v6 geometry is supported)")
        flat[off.value: off.value + a.size] = a
        i += 1
    return flat


class HipSileroVADWeights:
    """The network's weights on one GPU, shared by every stream's ``HipSileroVAD``."""

    def __init__(self, state_dict: Optional[Dict[str, np.ndarray]] = None, device: int = 0):
        self.lib = _lib.load()
        if _lib.device_count() <= 0:
            raise _lib.WlkError("no HIP device visible: the VAD HIP backend has no CPU fallback")
        flat = pack_vad_weights(state_dict if state_dict is not None else load_silero_state_dict())
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_vad_create(device, flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(self._h)))

    def close(self):
        if self._h:
            self.lib.wlk_vad_destroy(self._h)
            self._h = C.c_void_p()


class HipSileroVAD:
    """Per-stream model object with the TorchScript wrapper's duck type (forward(x, sr), reset_states())."""

    def __init__(self, weights: HipSileroVADWeights, max_windows: int = 64):
        self.lib, self.weights, self.max_windows = weights.lib, weights, max_windows
        self._h = C.c_void_p()
        _lib.check(self.lib.wlk_vad_stream_create(weights._h, max_windows, C.byref(self._h)))

    def reset_states(self, batch_size: int = 1):
        _lib.check(self.lib.wlk_vad_stream_reset(self._h))

    def probs(self, pcm: np.ndarray) -> np.ndarray:
        """Speech probability of each complete 512-sample window of ``pcm`` (len must be a multiple of 512)."""
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        if a.size == 0 or a.size % WINDOW:
            raise ValueError("HipSileroVAD.probs needs a positive multiple of 512 samples")
        out = np.empty(a.size // WINDOW, np.float32)
        for lo in range(0, out.size, self.max_windows):
            n = min(self.max_windows, out.size - lo)
            _lib.check(self.lib.wlk_vad_stream_run(self._h, a[lo * WINDOW:].ctypes.data_as(C.c_void_p), n,
                                                   out[lo:].ctypes.data_as(C.c_void_p)))
        return out

    def __call__(self, x, sr: int = SAMPLE_RATE):
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        a = np.asarray(x, np.float32).reshape(-1)
        if sr != SAMPLE_RATE:
            raise ValueError("HipSileroVAD supports 16000 Hz only")
        if a.size != WINDOW:
            raise ValueError(f"Provided number of samples is {a.size} (Supported values: 512 for 16000)")
        return self.probs(a)[0]          # np.float32: has .item() like the reference's tensor

    def state(self):
        h, c = np.empty(128, np.float32), np.empty(128, np.float32)
        _lib.check(self.lib.wlk_vad_stream_state(self._h, h.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
        return h, c

    def close(self):
        if self._h:
            self.lib.wlk_vad_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class HipFixedVADIterator:
    """VADIterator + FixedVADIterator (silero_vad_iterator.py:186-319) with all complete windows of a call evaluated
    in one device call.  ``model`` needs ``probs(pcm)`` and ``reset_states()``."""

    def __init__(self, model, threshold: float = 0.5, sampling_rate: int = SAMPLE_RATE,
                 min_silence_duration_ms: int = 100, speech_pad_ms: int = 30):
        if sampling_rate != SAMPLE_RATE:
            raise ValueError("HipFixedVADIterator supports 16000 Hz only")
        self.model, self.threshold, self.sampling_rate = model, threshold, sampling_rate
        self.min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
        self.speech_pad_samples = sampling_rate * speech_pad_ms / 1000
        self.reset_states()

    def reset_states(self):
        self.model.reset_states()
        self.triggered = False
        self.temp_end = 0
        self.current_sample = 0
        self.buffer = np.array([], dtype=np.float32)

    def _step(self, speech_prob: float, return_seconds: bool, time_resolution: int = 1) -> Optional[dict]:
        self.current_sample += WINDOW
        if speech_prob >= self.threshold and self.temp_end:
            self.temp_end = 0
        if speech_prob >= self.threshold and not self.triggered:
            self.triggered = True
            start = max(0, self.current_sample - self.speech_pad_samples - WINDOW)
            return {"start": int(start) if not return_seconds else round(start / self.sampling_rate, time_resolution)}
        if speech_prob < self.threshold - 0.15 and self.triggered:
            if not self.temp_end:
                self.temp_end = self.current_sample
            if self.current_sample - self.temp_end < self.min_silence_samples:
                return None
            end = self.temp_end + self.speech_pad_samples - WINDOW
            self.temp_end = 0
            self.triggered = False
            return {"end": int(end) if not return_seconds else round(end / self.sampling_rate, time_resolution)}
        return None

    def __call__(self, x, return_seconds: bool = False) -> List[dict]:
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        self.buffer = np.append(self.buffer, np.asarray(x, np.float32).reshape(-1))
        n = len(self.buffer) // WINDOW
        if n == 0:
            return []
        probs = self.model.probs(self.buffer[: n * WINDOW])
        self.buffer = self.buffer[n * WINDOW:]
        events = []
        for p in probs:
            r = self._step(float(p), return_seconds)
            if r is not None:
                events.append(r)
        return events
