"""Silero VAD gate, CPU side: the oracle is PINNED by vectors the reference produced with its own
vendored checkpoint (scripts/gen_golden_vad.py: silero_vad.jit via load_jit_vad + FixedVADIterator); the product's
iterator and weight packing are checked against it."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers as H
from oracle import vad_oracle as vo
from whisperlivekit_amd import _lib, synth
from whisperlivekit_amd import vad as V

META = H.golden_json("vad_cases.json")
GOLD = H.golden_npz("vad_cases.npz")
WEIGHTS = dict(H.golden_npz("vad_weights_16k.npz"))


def case_audio(name):
    sp = lambda sec, seed: synth.to_pcm16_roundtrip(synth.speech_like(sec, seed))
    gap = lambda sec: np.zeros(int(16000 * sec), np.float32)
    table = {
        "speech12": lambda: sp(12.0, 0),
        "speech8_ragged": lambda: sp(8.0, 1),
        "noise6": lambda: synth.to_pcm16_roundtrip(synth.white_noise(6.0, 3)),
        "silence3": lambda: gap(3.0),
        "gaps": lambda: np.concatenate([sp(2.0, 2), gap(1.0), sp(2.5, 3) * 1.6, gap(0.5), sp(1.0, 4)]).astype(np.float32),
        "loud_short_chunks": lambda: np.clip(sp(6.0, 5) * 1.9, -1, 1).astype(np.float32),
    }
    a = table[name]()
    assert len(a) == META[name]["n_samples"]
    return a


def feed(iterator, audio, chunking):
    events, per_call, at, k = [], [], 0, 0
    while at < len(audio):
        n = chunking[k % len(chunking)]
        k += 1
        ev = iterator(audio[at:at + n])
        per_call.append(ev)
        events += ev
        at += n
    return events, per_call


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_reproduces_the_reference_model_and_iterator(name):
    torch.set_num_threads(1)
    audio = case_audio(name)
    m = vo.OracleSileroVAD(WEIGHTS)
    probs = np.array([float(m(audio[i:i + 512])) for i in range(0, len(audio) - 511, 512)], np.float32)
    assert np.abs(probs - GOLD[name + "_probs"]).max() <= 1e-6            # same torch operators: expect 0
    assert np.abs(torch.stack([m.h, m.c]).numpy() - GOLD[name + "_state"]).max() <= 1e-6
    events, per_call = feed(vo.OracleVADIterator(vo.OracleSileroVAD(WEIGHTS)), audio, META[name]["chunking"])
    assert events == META[name]["events"] and per_call == META[name]["events_per_call"]


def test_golden_cases_exercise_both_sides_of_the_threshold():
    assert sum(len(m["events"]) for m in META.values()) >= 8
    assert any(m["frac_speech"] == 0 for m in META.values()) and any(m["frac_speech"] > 0.05 for m in META.values())


class ProbeModel:
    """Stands in for HipSileroVAD on the CPU: probabilities from the oracle network."""

    def __init__(self):
        self.m = vo.OracleSileroVAD(WEIGHTS)
        self.calls = []

    def reset_states(self, batch_size=1):
        self.m.reset_states()

    def probs(self, pcm):
        self.calls.append(len(pcm) // 512)
        return np.array([float(self.m(pcm[i:i + 512])) for i in range(0, len(pcm), 512)], np.float32)

    def __call__(self, x, sr=16000):
        return self.probs(np.asarray(x, np.float32).reshape(-1))[0]


@pytest.mark.parametrize("name", sorted(META))
def test_product_iterator_batches_windows_and_emits_the_reference_events(name):
    torch.set_num_threads(1)
    model = ProbeModel()
    events, per_call = feed(V.HipFixedVADIterator(model), case_audio(name), META[name]["chunking"])
    assert events == META[name]["events"] and per_call == META[name]["events_per_call"]
    if META[name]["chunking"] == [8000]:
        assert max(model.calls) >= 15        # a 0.5 s chunk is one device call, not 15-16


def test_model_duck_type_drives_the_references_own_iterator():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("reference tree not available")
    ref_stubs.install()
    from whisperlivekit.silero_vad_iterator import FixedVADIterator
    torch.set_num_threads(1)
    name = "gaps"
    events, _ = feed(FixedVADIterator(ProbeModel()), case_audio(name), META[name]["chunking"])
    assert events == META[name]["events"]


def packed_forward(flat, lookup, x1, h, c):
    """The index arithmetic of vad.hip on the PACKED buffer, in numpy float64."""
    g = lambda n: flat[lookup[n][0]: lookup[n][0] + lookup[n][1]].astype(np.float64)
    xs = np.concatenate([x1, x1[574 - np.arange(64)]]).astype(np.float64)
    basis = g("stft.basisT").reshape(256, 258)
    spec = np.stack([xs[t * 128: t * 128 + 256] @ basis for t in range(4)], 1)          # [258, 4]
    act = np.sqrt(spec[:129] ** 2 + spec[129:] ** 2)
    for i, (cin, cout, stride) in enumerate(((129, 128, 1), (128, 64, 2), (64, 64, 2), (64, 128, 1))):
        w = g(f"enc{i}.wT").reshape(cin, 3, cout)
        pad = np.zeros((cin, act.shape[1] + 2))
        pad[:, 1:-1] = act
        n_out = (act.shape[1] + 2 - 3) // stride + 1
        act = np.stack([np.einsum("ck,cko->o", pad[:, stride * t: stride * t + 3], w) for t in range(n_out)], 1)
        act = np.maximum(act + g(f"enc{i}.b")[:, None], 0)
    feat = act[:, 0]
    gates = (feat @ g("rnn.wihT").reshape(128, 512) + g("rnn.bih")) + (h @ g("rnn.whhT").reshape(128, 512) + g("rnn.bhh"))
    sig = lambda v: 1 / (1 + np.exp(-v))
    i_, f_, g_, o_ = gates[:128], gates[128:256], gates[256:384], gates[384:]
    c2 = sig(f_) * c + sig(i_) * np.tanh(g_)
    h2 = sig(o_) * np.tanh(c2)
    return sig(np.maximum(h2, 0) @ g("dec.w") + g("dec.b")[0]), h2, c2


def test_weight_packing_preserves_the_network_function():
    lib = _lib.load()
    flat = V.pack_vad_weights(WEIGHTS)
    lookup, i = {}, 0
    while True:
        name = C.c_char_p()
        if lib.wlk_vad_tensor_name(i, C.byref(name)) != 0:
            break
        off, numel = C.c_uint64(), C.c_uint64()
        assert lib.wlk_vad_tensor_lookup(name.value, C.byref(off), C.byref(numel)) == 0
        lookup[name.value.decode()] = (off.value, numel.value)
        i += 1
    assert len(lookup) == 15 and lib.wlk_vad_tensor_lookup(b"nope", None, None) != 0
    audio = case_audio("gaps")
    m = vo.OracleSileroVAD(WEIGHTS)
    h, c, ctx = np.zeros(128), np.zeros(128), np.zeros(64, np.float32)
    for wdw in range(70, 78):                       # windows around the second speech burst, both from a zero state
        x = audio[wdw * 512: (wdw + 1) * 512]
        p_ref = float(m(x))
        p, h, c = packed_forward(flat, lookup, np.concatenate([ctx, x]), h, c)
        ctx = x[-64:]
        assert abs(p - p_ref) <= 2e-5
    bad = dict(WEIGHTS)
    bad["decoder.rnn.weight_hh"] = np.zeros((512, 64), np.float32)
    with pytest.raises(ValueError):
        V.pack_vad_weights(bad)
