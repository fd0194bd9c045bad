#!/usr/bin/env python
"""Golden runs of the REFERENCE's batch Whisper (whisperlivekit/whisper/transcribe.py + decoding.py + timing.py) for the
LocalAgreement row.  Build container only: needs /root/reference or WLK_REFERENCE_ROOT.

The unmodified reference `transcribe()` runs on the seeded micro Whisper of the other goldens (synth weights in the
reference's own `Whisper`, real vocabulary) over seeded synthetic audio.  Two observers are wrapped around reference
methods (they call the original and only take notes):

* `DecodingTask.run`      - one record per decode call: temperature, rows, the initial tokens, the DecodingResult fields;
* `GreedyDecoder.update`  - one record per step of the greedy / sampling calls: the token every row got and its log-prob
  under the filtered logits.  A test replays these choices (checking, step by step, that its own logits make the same
  choice or sit within the tie margin / that the drawn token has the same probability), so the comparison of everything
  downstream does not hinge on torch's generator or on 1-ulp arg-max ties.

Writes tests/golden/transcribe_kat.json.gz.
"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402

CASES = [
    # what LocalAgreement's WhisperASR.transcribe passes (local_agreement/backends.py:63-79): every temperature is tried
    dict(name="local_agreement_call", model="micro.en", audio=dict(kind="speech_like", seconds=41.0, seed=11), torch_seed=1234,
         kwargs=dict(language="en", initial_prompt="", condition_on_previous_text=True, word_timestamps=True)),
    # greedy only, a carried prompt, no word timestamps, two windows
    dict(name="greedy_prompt", model="micro.en", audio=dict(kind="speech_like", seconds=33.0, seed=12), torch_seed=1,
         kwargs=dict(language="en", temperature=0.0, initial_prompt="Hello world, this is the glossary.",
                     carry_initial_prompt=True, logprob_threshold=None, compression_ratio_threshold=None)),
    # beam search first, then best-of-2 sampling; thresholds off for the last temperature only by construction
    dict(name="beam_then_best_of", model="micro.en", audio=dict(kind="white_noise", seconds=7.3, seed=13), torch_seed=7,
         kwargs=dict(language="en", temperature=(0.0, 0.6), beam_size=3, patience=1.5, best_of=2, length_penalty=0.6,
                     word_timestamps=True)),
    # multilingual twin: language detection, clips, the hallucination rules, no timestamps suppressed blanks off
    dict(name="multilingual_clips", model="micro", audio=dict(kind="speech_like", seconds=52.0, seed=14), torch_seed=99,
         kwargs=dict(temperature=(0.0, 1.0), clip_timestamps="1.5,12.25,20", word_timestamps=True,
                     hallucination_silence_threshold=0.5, condition_on_previous_text=False, suppress_blank=False)),
    # the same rules with a threshold that lets segments through; one clip, text fed back between windows
    dict(name="hallucination_rules_kept", model="micro", audio=dict(kind="speech_like", seconds=38.0, seed=16), torch_seed=17,
         kwargs=dict(language="de", temperature=(0.0, 0.8), clip_timestamps=[0.5, 36.0], word_timestamps=True,
                     hallucination_silence_threshold=4.0, no_speech_threshold=None)),
    dict(name="without_timestamps_prefix", model="micro", audio=dict(kind="speech_like", seconds=4.0, seed=15), torch_seed=3,
         kwargs=dict(language="fr", task="translate", temperature=0.0, without_timestamps=True, prefix="Bonjour",
                     sample_len=40, suppress_tokens="11,13", logprob_threshold=None, compression_ratio_threshold=None,
                     no_speech_threshold=None)),
    # shorter than one hop of content / silence
    dict(name="silence", model="micro.en", audio=dict(kind="zeros", seconds=2.0, seed=0), torch_seed=5,
         kwargs=dict(language="en", temperature=(0.0, 0.4), word_timestamps=True)),
]


def make_audio(spec):
    from whisperlivekit_amd import synth
    if spec["kind"] == "speech_like":
        return synth.speech_like(spec["seconds"], seed=spec["seed"])
    if spec["kind"] == "white_noise":
        return synth.white_noise(spec["seconds"], seed=spec["seed"])
    return np.zeros(int(spec["seconds"] * 16000), np.float32)


def jsonable(x):
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def main():
    ref_stubs.install(synthetic_vocab=False)
    import torch
    import torch.nn.functional as F
    from whisperlivekit.whisper import decoding as D
    from whisperlivekit.whisper.transcribe import transcribe

    import gen_golden

    calls = []
    real_run, real_update = D.DecodingTask.run, D.GreedyDecoder.update

    def run_spy(self, mel):
        rec = dict(temperature=float(self.options.temperature), rows=int(self.n_group),
                   initial_tokens=[int(t) for t in self.initial_tokens], beam=self.options.beam_size, steps=[], logprobs=[])
        calls.append(rec)
        out = real_run(self, mel)
        r = out[0]
        rec["result"] = dict(tokens=[int(t) for t in r.tokens], text=r.text, avg_logprob=float(r.avg_logprob),
                             no_speech_prob=float(r.no_speech_prob), compression_ratio=float(r.compression_ratio),
                             language=r.language)
        return out

    def update_spy(self, tokens, logits, sum_logprobs):
        lp = F.log_softmax(logits.float(), dim=-1)
        new_tokens, completed = real_update(self, tokens, logits, sum_logprobs)
        chosen = new_tokens[:, -1]
        was_live = tokens[:, -1] != self.eot
        # rows that already ended get <|endoftext|> forced: their entry is marked with probability None
        calls[-1]["steps"].append([int(t) for t in chosen])
        calls[-1]["logprobs"].append([float(lp[r, chosen[r]]) if bool(was_live[r]) else None for r in range(len(chosen))])
        return new_tokens, completed

    D.DecodingTask.run = run_spy
    D.GreedyDecoder.update = update_spy

    out = []
    models = {}
    for case in CASES:
        name = case["model"]
        if name not in models:
            models[name] = gen_golden.build_reference_model(name, 0)
        model = models[name]
        audio = make_audio(case["audio"])
        calls.clear()
        torch.manual_seed(case["torch_seed"])
        kwargs = dict(case["kwargs"])
        result = transcribe(model, audio, **kwargs)
        for seg in result["segments"]:
            for w in seg.get("words", []):
                w["probability"] = float(w["probability"])
        out.append(dict(name=case["name"], model=name, audio=case["audio"], torch_seed=case["torch_seed"],
                        kwargs=jsonable(case["kwargs"]), result=jsonable(result), calls=jsonable(list(calls))))
        print(case["name"], "->", len(result["segments"]), "segments,", len(calls), "decode calls,",
              sum(len(c["steps"]) for c in calls), "greedy steps; temps", sorted({c["temperature"] for c in calls}),
              "language", result["language"])
    D.DecodingTask.run, D.GreedyDecoder.update = real_run, real_update

    path = os.path.join(ROOT, "tests", "golden", "transcribe_kat.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(out, fh)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
