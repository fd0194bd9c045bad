#!/usr/bin/env python
"""tests/golden/tokenizer_kat.json: known answers of the REFERENCE's Tokenizer (whisper/tokenizer.py:161-332) and
TokenBuffer (simul_whisper/token_buffer.py:5-95) over its own vendored rank tables (gpt2.tiktoken,
multilingual.tiktoken): decode, split_to_word_tokens (space / unicode splitting, incomplete UTF-8 tails), encode of
prompt text, context trimming and the pending-token carry of append_token_ids.  Also re-packs multilingual.tiktoken
next to vocab_gpt2.npz.  Build container only (needs /root/reference):

    python scripts/gen_golden_tokenizer.py

tiktoken itself is absent from the image: the reference's classes run over this repo's BpeEncoding as the byte-pair
engine (scripts/ref_stubs.py), whose merges are pinned separately by public GPT-2 known answers
(tests/test_tokenizer_real_vocab.py)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install(synthetic_vocab=False)
from whisperlivekit.simul_whisper.token_buffer import TokenBuffer  # noqa: E402
from whisperlivekit.whisper.tokenizer import get_tokenizer  # noqa: E402

from whisperlivekit_amd.tokenizer import load_tiktoken_ranks  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = ref_stubs.REFERENCE_ROOT

TEXTS = [
    " Hello, world. The quick brown fox jumps over the lazy dog!",
    "It's 9:45 - don't panic... (really?) [ok] \"quoted\" 'single'",
    " naïve café déjà vu — über straße",
    " 你好，世界。今天天气很好",
    " こんにちは世界 ありがとう",
    " 🙂 emoji 👍🏽 mixed ✈️ text",
    " Привет, мир! Как дела?",
    "  double  spaces\n\nnew lines\ttabs ",
    " e-mail: a.b@c.org, $3.50, 50% off; #tag @user",
    " สวัสดีครับ ขอบคุณ",
]


def cases_for(tok, lang, multilingual, rng):
    out = []
    def add(ids, note):
        words, groups = tok.split_to_word_tokens(list(ids))
        out.append(dict(multilingual=multilingual, language=lang, note=note, ids=[int(i) for i in ids],
                        decode=tok.decode(list(ids)), decode_ts=tok.decode_with_timestamps(list(ids)),
                        words=words, groups=groups))
    for text in TEXTS:
        ids = tok.encode(text)
        add(ids, "text")
        for cut in sorted(set(int(x) for x in rng.integers(1, max(2, len(ids)), 3))):
            add(ids[:cut], "prefix")          # may end inside a UTF-8 sequence
            add(ids[cut:], "suffix")          # may start inside one
    n_base = 50256 if not multilingual else 50257
    for _ in range(12):
        n = int(rng.integers(1, 24))
        ids = rng.integers(0, n_base, n).tolist()
        if rng.random() < 0.4:
            ids.insert(int(rng.integers(0, len(ids) + 1)), int(tok.timestamp_begin + rng.integers(0, 1500)))
        if rng.random() < 0.3:
            ids.append(int(tok.eot))
        add(ids, "random")
    return out


def buffer_cases(tok, rng):
    out = []
    for text in TEXTS[:7]:
        ids = tok.encode(text)
        buf = TokenBuffer.from_text(" ctx", tokenizer=tok, prefix_token_ids=[tok.sot_prev])
        steps = []
        pos = 0
        while pos < len(ids):
            n = int(rng.integers(1, 4))
            piece = ids[pos:pos + n]
            pos += n
            buf.append_token_ids(list(piece))
            steps.append(dict(append=[int(i) for i in piece], text=buf.text, pending=[int(i) for i in buf.pending_token_ids]))
        trims = []
        for _ in range(3):
            dropped = buf.trim_words(after=4)
            trims.append(dict(dropped=int(dropped), text=buf.text, as_token_ids=[int(i) for i in buf.as_token_ids()]))
        out.append(dict(steps=steps, trims=trims))
    return out


def pack_vocab(name):
    ranks = load_tiktoken_ranks(os.path.join(REF, "whisperlivekit", "whisper", "assets", f"{name}.tiktoken"))
    toks = [b for b, _ in sorted(ranks.items(), key=lambda kv: kv[1])]
    np.savez_compressed(os.path.join(OUT, f"vocab_{name}.npz"), lengths=np.array([len(b) for b in toks], np.uint16),
                        blob=np.frombuffer(b"".join(toks), np.uint8))


if __name__ == "__main__":
    pack_vocab("multilingual")
    rng = np.random.default_rng(17)
    kat = dict(split=[], buffer=[], specials={})
    for multilingual, lang in ((False, None), (True, "en"), (True, "zh"), (True, "ja"), (True, "th")):
        tok = get_tokenizer(multilingual, num_languages=99, language=lang, task="transcribe" if multilingual else None)
        kat["split"] += cases_for(tok, lang, multilingual, rng)
        key = f"{'multi' if multilingual else 'en'}:{lang}"
        kat["specials"][key] = dict(eot=tok.eot, sot=tok.sot, sot_prev=tok.sot_prev, sot_lm=tok.sot_lm,
                                    no_speech=tok.no_speech, no_timestamps=tok.no_timestamps,
                                    timestamp_begin=tok.timestamp_begin, transcribe=tok.transcribe, translate=tok.translate,
                                    sot_sequence=list(tok.sot_sequence),
                                    sot_sequence_including_notimestamps=list(tok.sot_sequence_including_notimestamps),
                                    n_language_tokens=len(tok.all_language_tokens),
                                    language_tokens_sorted=sorted(int(t) for t in tok.all_language_tokens),  # set-iteration order in the reference
                                    blank=tok.encode(" "))
    tok = get_tokenizer(False, num_languages=99)
    kat["buffer"] = buffer_cases(tok, rng)
    json.dump(kat, open(os.path.join(OUT, "tokenizer_kat.json"), "w"), ensure_ascii=True)
    n_bad = sum(1 for c in kat["split"] if "�" in c["decode_ts"])
    print(len(kat["split"]), "split cases,", n_bad, "with incomplete UTF-8;", len(kat["buffer"]), "buffer cases")
