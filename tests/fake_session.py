"""CPU stand-in for engine.HipWhisperModel / HipSession, for TESTS ONLY.

It implements the same Python surface on top of the oracle's numerics so that the host-side code
of the product (policy.py, align_att.py, backend.py - everything above the C ABI) can be pinned
against the reference's golden streams on a machine without a GPU.  It is not importable from the
package and nothing in the product can reach it."""
import types

import numpy as np
import torch

from oracle import whisper_oracle as wo
from whisperlivekit_amd.melbank import mel_filterbank


class FakeHipModel:
    _wlk_hip_model = True

    def __init__(self, dims, sd_torch, alignment_heads, device=0):
        self.dims, self.sd, self.alignment_heads, self.device = dims, sd_torch, list(alignment_heads), device
        self.decoder = types.SimpleNamespace(blocks=[None] * dims.n_text_layer)
        self.filters = torch.from_numpy(np.array(mel_filterbank(dims.n_mels)))
        self.finalized = True

    is_multilingual = property(lambda self: self.dims.is_multilingual)
    num_languages = property(lambda self: self.dims.num_languages)

    def new_session(self, beam=1, max_audio_seconds=64.0):
        return FakeSession(self, beam)


class FakeSession:
    def __init__(self, model, beam):
        self.model, self.beam = model, beam
        self.audio = np.zeros(0, np.float32)
        self.calls = []            # numeric trace, same shape as the oracle's
        self.steps = []

    # audio
    def append(self, pcm):
        self.audio = np.concatenate([self.audio, np.asarray(pcm, np.float32)])

    def append_pcm16(self, pcm):
        self.append(np.asarray(pcm, np.int16).astype(np.float32) / 32768.0)     # audio_processor.py:416-418

    def append_zeros(self, n):
        self.audio = np.concatenate([self.audio, np.zeros(n, np.float32)])

    def drop_front(self, n):
        self.audio = self.audio[n:]

    def clear_audio(self):
        self.audio = np.zeros(0, np.float32)

    @property
    def audio_len(self):
        return len(self.audio)

    # hot path
    @torch.no_grad()
    def encode(self):
        m = self.model
        mel, cml = wo.encoder_input_from_audio(torch.from_numpy(self.audio.copy()), m.filters)
        self.enc = wo.encoder_forward(m.sd, m.dims, mel)
        self.cml = cml
        self.cache = None
        return cml

    @torch.no_grad()
    def decode(self, tokens, first, sot_index=0):
        m = self.model
        if first:
            self.cache = wo.DecoderCache(m.dims.n_text_layer)
            self.kept = []
        logits, cross = wo.decoder_forward(m.sd, m.dims, torch.from_numpy(np.asarray(tokens)), self.enc, self.cache)
        self.kept = (self.kept + [cross])[-16:]
        self.logits_last = logits[:, -1, :].clone()
        self.logits_sot = logits[:, sot_index, :].clone() if first else None

    def no_speech_prob(self, token):
        return self.logits_sot.float().softmax(-1)[:, token].numpy()

    @torch.no_grad()
    def select(self, adj_rows, adj_ids, adj_deltas, k, content_mel_len):
        lg = self.logits_last
        for r, t, dl in zip(adj_rows, adj_ids, adj_deltas):
            if r < 0:
                lg[:, t] += dl
            else:
                lg[r, t] += dl
        lp = torch.log_softmax(lg.float(), -1)
        vals, ids = lp.topk(k, dim=-1)
        attn = wo.alignatt_attention(self.kept, self.model.alignment_heads, self.model.dims.n_text_layer,
                                     content_mel_len, self.beam)
        if content_mel_len > 0:
            frames = torch.argmax(attn[:, -1, :], dim=-1).numpy().astype(np.int32)
        else:
            frames = np.zeros(self.beam, np.int32)
        return vals.numpy(), ids.numpy().astype(np.int32), frames

    def kv_reorder(self, src):
        self.cache.reorder(list(src))

    def decode_until_stop(self, tokens, params, suppress_ids, blank_ids):
        """wlk_decode_until_stop on CPU: the LIBRARY's own host logic of the loop (wlk_job_*: budget, DRY penalty,
        beam-1 update, stop rules) driven with the oracle's numerics in place of the kernels."""
        import ctypes as C
        from whisperlivekit_amd import _lib
        from whisperlivekit_amd.engine import LoopOutcome
        lib = _lib.load()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        t = np.ascontiguousarray(tokens, dtype=np.int64).reshape(-1)
        sup = np.ascontiguousarray(suppress_ids, dtype=np.int32)
        blank = np.ascontiguousarray(blank_ids, dtype=np.int32)
        job = C.c_void_p()
        _lib.check(lib.wlk_job_create(C.byref(params), vp(t), t.size, vp(sup), sup.size, vp(blank), blank.size, C.byref(job)))
        seq = [int(x) for x in t]
        first = True                                # each call is one infer: its first decoder forward is the prefill
        try:
            while True:
                n_feed = C.c_int32()
                _lib.check(lib.wlk_job_begin_step(job, C.byref(n_feed)))
                if n_feed.value == 0:
                    break
                feed = np.asarray([seq if first else seq[-1:]], dtype=np.int64)
                self.decode(feed, first=first, sot_index=int(params.sot_index))
                if first and params.no_speech_token >= 0:
                    stops = C.c_int32()
                    _lib.check(lib.wlk_job_no_speech(job, float(self.no_speech_prob(int(params.no_speech_token))[0]), C.byref(stops)))
                    if stops.value:
                        break
                first = False
                ids_p, dl_p, n = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int32()
                _lib.check(lib.wlk_job_adjustments(job, C.byref(ids_p), C.byref(dl_p), C.byref(n)))
                ids = [ids_p[i] for i in range(n.value)]
                dls = [dl_p[i] for i in range(n.value)]
                assert len(set(ids)) == len(ids)
                lp, top, frames = self.select([-1] * len(ids), ids, dls, 2, int(params.content_mel_len))
                lp = np.ascontiguousarray(lp[0], np.float32)
                top = np.ascontiguousarray(top[0], np.int32)
                go = C.c_int32()
                _lib.check(lib.wlk_job_consume(job, vp(lp), vp(top), int(frames[0]), C.byref(go)))
                nxt = int(top[1]) if int(top[0]) == int(params.eot) else int(top[0])
                seq.append(nxt)
                if not go.value:
                    break
            cap = int(params.max_text_len) + 8
            res = _lib.LoopResult()
            new = np.empty(cap, np.int64)
            st, sf, ss = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
            _lib.check(lib.wlk_job_result(job, C.byref(res), vp(new), vp(st), vp(sf), vp(ss), cap))
            return LoopOutcome(res, new, st, sf, ss)
        finally:
            lib.wlk_job_destroy(job)

    def export(self, what, max_floats=None):
        if what == "enc":
            return self.enc.numpy().reshape(-1)
        if what == "logits_last":
            return self.logits_last.numpy().reshape(-1)
        raise KeyError(what)

    def close(self):
        pass
