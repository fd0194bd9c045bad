"""Oracle-backed stand-ins for HipWhisperModel / HipSession (TEST infrastructure only): the same methods the batch-Whisper
host code calls (whisperlivekit_amd/transcribe.py, timing.py), answered by the CPU oracle (oracle/whisper_oracle.py,
oracle/timing_oracle.py).  They let the CPU suite pin the host logic of `transcribe()` against the reference's goldens
without a GPU; the gpu-marked twin of every such test runs the real library."""
import numpy as np
import torch

from oracle import timing_oracle
from oracle import whisper_oracle as wo
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
from whisperlivekit_amd.melbank import mel_filterbank

import helpers as H


class OracleSession:
    def __init__(self, model, beam):
        self.model, self.beam = model, beam
        self.mel = self.xa = self.cache = None
        self.logits_last = self.logits_sot = None

    def log_mel(self, audio, padding=0):
        with torch.no_grad():
            return wo.log_mel_spectrogram(torch.from_numpy(np.asarray(audio, np.float32)), self.model.filters,
                                          padding=padding).numpy()

    def encode_mel(self, mel):
        assert mel.shape == (self.model.dims.n_mels, 3000)
        self.mel = torch.from_numpy(np.ascontiguousarray(mel, np.float32))
        with torch.no_grad():
            self.xa = wo.encoder_forward(self.model.sd, self.model.dims, self.mel.unsqueeze(0))

    def decode(self, tokens, first, sot_index=0):
        t = torch.from_numpy(np.ascontiguousarray(tokens, np.int64))
        assert t.shape[0] == self.beam
        if first:
            self.cache = wo.DecoderCache(self.model.dims.n_text_layer)
        with torch.no_grad():
            logits, _ = wo.decoder_forward(self.model.sd, self.model.dims, t, self.xa, self.cache)
        self.logits_last = logits[:, -1].clone()
        if first:
            self.logits_sot = logits[:, sot_index].clone()

    def no_speech_prob(self, token):
        return self.logits_sot.softmax(dim=-1)[:, token].numpy()

    def export(self, what, max_floats=None):
        return {"logits_last": self.logits_last, "logits_sot": self.logits_sot}[what].numpy().reshape(-1).copy()

    def kv_reorder(self, source_rows):
        self.cache.reorder(list(source_rows))

    def sync(self):
        pass

    def find_alignment(self, tokens, n_sot, eot, num_frames, qk_scale=1.0, want_cost=False):
        tokens = [int(t) for t in tokens]
        cost, probs = timing_oracle.alignment_cost(self.model.sd, self.model.dims, self.model.align_heads, self.mel,
                                                   tokens[:n_sot], tokens[n_sot], tokens[n_sot + 1:-1], eot, num_frames,
                                                   qk_scale=qk_scale)
        return timing_oracle.dtw_trace(cost), np.asarray(probs, np.float32), (cost if want_cost else None)

    def close(self):
        pass


class OracleModel:
    def __init__(self, name, seed=0):
        self.dims = MODEL_DIMS[name]
        self.sd = H.oracle_sd(name, seed)
        self.align_heads = ALIGNMENT_HEADS[name]
        self.filters = torch.from_numpy(np.array(mel_filterbank(self.dims.n_mels)))
        self.is_multilingual = self.dims.is_multilingual
        self.num_languages = self.dims.num_languages

    def new_session(self, beam=1, max_audio_seconds=64.0, batched=None):
        return OracleSession(self, beam)
