#!/usr/bin/env python
"""Known answers for the host half of the reference's word timestamps (whisperlivekit/whisper/timing.py:220-388),
produced by the REFERENCE's own functions (build container only: needs /root/reference or WLK_REFERENCE_ROOT).

* `find_alignment` on a seeded micro Whisper with the reference's real English vocabulary: the cost matrix it hands to
  dtw, the path, the token probabilities and the resulting WordTiming list (pins `word_timings`, and the DTW on a
  matrix that really came out of z-scored, median-filtered attention);
* `merge_punctuations` on synthetic alignments full of opening / closing punctuation;
* `add_word_timestamps` with `find_alignment` replaced by prepared alignments (pins `attach_words`: duration clamps,
  dealing words to segments, reconciling segment and word boundaries).

Writes tests/golden/word_timing_kat.json.gz.
"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402


def timing_dict(t):
    return dict(word=t.word, tokens=[int(x) for x in t.tokens], start=float(t.start), end=float(t.end),
                probability=float(t.probability))


def alignment_cases(rng, WordTiming):
    """Synthetic alignments: words with leading spaces, opening / closing punctuation, sentence ends, long words."""
    vocab = [" the", " cat", " sat", " on", " a", " mat", " Mr", " extraordinarily", " (", " \"", " -", " [", ".", ",",
             "!", "?", ")", "\"", ":", "。", " well", " so", " ¿", "'s", " 42", "%"]
    cases = []
    for n in (1, 2, 3, 5, 8, 13, 21, 34):
        for rep in range(3):
            t, out, tok = 0.0, [], 100
            for _ in range(n):
                w = vocab[int(rng.integers(len(vocab)))]
                dur = float(rng.choice([0.0, 0.02, 0.1, 0.24, 0.3, 0.5, 1.4, 3.0]))
                k = int(rng.integers(1, 4))
                out.append(WordTiming(w, list(range(tok, tok + k)), round(t, 2), round(t + dur, 2), float(rng.random())))
                tok += k
                t += dur + float(rng.choice([0.0, 0.0, 0.2, 2.5]))
            cases.append(out)
    return cases


def main():
    ref_stubs.install(synthetic_vocab=False)
    import torch
    from whisperlivekit.whisper import timing as T
    from whisperlivekit.whisper.tokenizer import get_tokenizer

    out = {"find_alignment": [], "merge": [], "attach": []}
    rng = np.random.default_rng(0)

    # ---- find_alignment on the seeded micro model of the other goldens (synth weights in the reference's Whisper) ----
    import gen_golden
    model = gen_golden.build_reference_model("micro.en", 0)
    dims = model.dims
    tokenizer = get_tokenizer(False, language="en", task="transcribe")
    texts = [" Hello, world! This is a test.", " (well) \"quoted\" - text: done", " naïve café — 你好。", " one"]
    captured = {}
    real_dtw = T.dtw

    def spy(x):
        path = real_dtw(x)
        captured["matrix"] = x.detach().cpu().numpy().astype(np.float32)
        captured["path"] = np.asarray(path)
        return path
    T.dtw = spy
    for i, text in enumerate(texts):
        text_tokens = tokenizer.encode(text)
        case_rng = np.random.default_rng(100 + i)      # tests rebuild the mel from this seed
        mel = torch.from_numpy(case_rng.standard_normal((dims.n_mels, 3000)).astype(np.float32))
        num_frames = int(case_rng.integers(400, 3000))
        # the tail of find_alignment consumes text_token_probs: capture them through the returned probabilities
        got = T.find_alignment(model, tokenizer, text_tokens, mel, num_frames)
        words, word_tokens = tokenizer.split_to_word_tokens(text_tokens + [tokenizer.eot])
        # per-token probabilities are not returned: recompute them exactly as find_alignment does
        tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot])
        with torch.no_grad():
            logits = model(mel.unsqueeze(0), tokens.unsqueeze(0))[0]
        probs = logits[len(tokenizer.sot_sequence):, : tokenizer.eot].softmax(dim=-1)
        token_probs = probs[np.arange(len(text_tokens)), text_tokens].tolist()
        out["find_alignment"].append(dict(
            text=text, text_tokens=[int(t) for t in text_tokens], num_frames=num_frames, mel_seed=100 + i,
            sot_sequence=[int(t) for t in tokenizer.sot_sequence], no_timestamps=int(tokenizer.no_timestamps),
            eot=int(tokenizer.eot),
            matrix_shape=list(captured["matrix"].shape),
            matrix=[float(v) for v in captured["matrix"].ravel()],
            path=captured["path"].tolist(), words=list(words), word_tokens=[[int(t) for t in w] for w in word_tokens],
            text_token_probs=[float(p) for p in token_probs], timings=[timing_dict(t) for t in got]))
        print("find_alignment", repr(text), captured["matrix"].shape, len(got), "words")
    T.dtw = real_dtw

    # ---- merge_punctuations ----------------------------------------------------------------------------------------
    for case in alignment_cases(rng, T.WordTiming):
        before = [timing_dict(t) for t in case]
        T.merge_punctuations(case, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、")
        out["merge"].append(dict(before=before, after=[timing_dict(t) for t in case]))

    # ---- add_word_timestamps with a prepared alignment --------------------------------------------------------------
    real_find = T.find_alignment
    for case in alignment_cases(rng, T.WordTiming):
        n_tok = sum(len(t.tokens) for t in case)
        all_tokens = [tok for t in case for tok in t.tokens]
        # split the tokens into 1..3 segments at word boundaries; segment times loosely around the words
        n_seg = int(rng.integers(1, 4))
        cuts = sorted(set(int(c) for c in rng.integers(0, len(case) + 1, size=n_seg - 1)))
        bounds = [0] + cuts + [len(case)]
        seek = int(rng.choice([0, 300, 2999]))
        segments = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            toks = [tok for t in case[lo:hi] for tok in t.tokens]
            if hi > lo:
                s0 = case[lo].start + float(rng.choice([-0.8, 0.0, 0.3, 0.9]))
                e0 = case[hi - 1].end + float(rng.choice([-0.9, -0.2, 0.0, 0.7]))
            else:
                s0 = e0 = 0.0
            # a timestamp token (>= eot) inside the segment: add_word_timestamps filters those out
            segments.append(dict(seek=seek, start=round(seek * 0.01 + max(s0, 0.0), 2), end=round(seek * 0.01 + max(e0, 0.0), 2),
                                 tokens=toks + [50363 + 5]))
        last_speech = float(rng.choice([0.0, 1.0, 7.5]))
        before = dict(segments=copy.deepcopy(segments), alignment=[timing_dict(t) for t in case],
                      last_speech_timestamp=last_speech)
        T.find_alignment = lambda *a, _c=case, **k: _c
        T.add_word_timestamps(segments=segments, model=None, tokenizer=tokenizer, mel=None, num_frames=0,
                              last_speech_timestamp=last_speech)
        out["attach"].append(dict(before=before, after=segments, eot=int(tokenizer.eot)))
        assert n_tok == len(all_tokens)
    T.find_alignment = real_find

    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "word_timing_kat.json.gz"), "wt") as fh:
        json.dump(out, fh)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
