#!/usr/bin/env python
"""Golden vectors for the Silero VAD gate, produced by the REFERENCE itself:
whisperlivekit/silero_vad_models/silero_vad.jit (real weights, vendored in the reference tree) driven exactly as
audio_processor.py:1189-1190 drives it (FixedVADIterator, silero_vad_iterator.py:186-319).

Writes tests/golden/vad_weights_16k.npz (the 16 kHz sub-model's tensors - the checkpoint the HIP path needs on the
GPU box, where /root/reference does not exist) and tests/golden/vad_cases.npz / vad_cases.json: per-window speech
probabilities, the final LSTM state, and the iterator's events for several audios and chunkings."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_stubs  # noqa: E402

ref_stubs.install()
from whisperlivekit.silero_vad_iterator import FixedVADIterator, load_jit_vad  # noqa: E402

from whisperlivekit_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
torch.set_num_threads(1)


def cases():
    sp = lambda sec, seed: synth.to_pcm16_roundtrip(synth.speech_like(sec, seed))
    gap = lambda sec: np.zeros(int(16000 * sec), np.float32)
    yield "speech12", sp(12.0, 0), [8000]
    yield "speech8_ragged", sp(8.0, 1), [700, 8000, 333, 512, 1, 4095, 16000]
    yield "noise6", synth.to_pcm16_roundtrip(synth.white_noise(6.0, 3)), [8000]
    yield "silence3", gap(3.0), [8000]
    yield "gaps", np.concatenate([sp(2.0, 2), gap(1.0), sp(2.5, 3) * 1.6, gap(0.5), sp(1.0, 4)]).astype(np.float32), [8000]
    yield "loud_short_chunks", np.clip(sp(6.0, 5) * 1.9, -1, 1).astype(np.float32), [640]


def main():
    model = load_jit_vad()
    sd = {k[len("_model."):]: v.detach().numpy() for k, v in model.state_dict().items() if k.startswith("_model.")}
    np.savez_compressed(os.path.join(OUT, "vad_weights_16k.npz"), **sd)
    arrays, meta = {}, {}
    with torch.no_grad():
        for name, audio, chunking in cases():
            model.reset_states()
            probs = [float(model(torch.from_numpy(audio[i:i + 512]).unsqueeze(0), 16000))
                     for i in range(0, len(audio) - 511, 512)]
            state = model._state.detach().numpy().copy()
            vad = FixedVADIterator(model)          # resets the model's states
            events, at, k = [], 0, 0
            per_call = []
            while at < len(audio):
                n = chunking[k % len(chunking)]
                k += 1
                ev = vad(audio[at:at + n])
                per_call.append(ev)
                events += ev
                at += n
            arrays[name + "_probs"] = np.asarray(probs, np.float32)
            arrays[name + "_state"] = state.astype(np.float32)
            meta[name] = dict(n_samples=int(len(audio)), chunking=chunking, events=events, events_per_call=per_call,
                              frac_speech=float(np.mean(np.asarray(probs) >= 0.5)))
            print(name, len(probs), "windows,", len(events), "events, speech fraction", round(meta[name]["frac_speech"], 3))
    np.savez_compressed(os.path.join(OUT, "vad_cases.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "vad_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
