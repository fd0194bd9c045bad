"""`encode` of the byte-pair encoding (whisperlivekit_amd/tokenizer.BpeEncoding) against an implementation that shares no
code with it: the reference's rank files (tests/golden/vocab_*.npz = whisper/assets/*.tiktoken) converted by `transformers`'
TikTokenConverter into a `tokenizers` (Rust) BPE model with the reference's split pattern (whisper/tokenizer.py:342).

Why: the golden generators run the reference with `tiktoken` replaced by this repository's own encodings (tiktoken is not
installed here), so the tokenizer known answers pin `decode`, the special-token layout and the word splitting, but `encode`
only through the public GPT-2 examples.  This test closes that gap for both vocabularies on a few thousand seeded strings
(ASCII words, digits, punctuation runs, contractions, leading / trailing / repeated whitespace, Latin-1, CJK, emoji)."""
import base64
import random
import sys
import types

import pytest

import helpers as H
from whisperlivekit_amd import tokenizer as wtok

GPT2_SPLIT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


def _load_ranks(path):
    ranks = {}
    with open(path, "rb") as fh:
        for line in fh:
            if line.strip():
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    return ranks


@pytest.fixture(scope="module")
def vocab_dir(tmp_path_factory):
    return H.real_vocab_dir(tmp_path_factory.mktemp("vocab"))


def _independent(path, monkeypatch):
    pytest.importorskip("tokenizers")
    conv = pytest.importorskip("transformers.convert_slow_tokenizer")
    fake = types.ModuleType("tiktoken")
    fake.load = types.ModuleType("tiktoken.load")
    fake.load.load_tiktoken_bpe = _load_ranks          # the converter only needs the rank table
    monkeypatch.setitem(sys.modules, "tiktoken", fake)
    monkeypatch.setitem(sys.modules, "tiktoken.load", fake.load)
    return conv.TikTokenConverter(vocab_file=path, pattern=GPT2_SPLIT).converted()


def _texts(seed, n):
    rng = random.Random(seed)
    words = ["the", "Hello", "world", "I'm", "don't", "we'll", "they've", "it's", "naïve", "café", "Zürich", "über", "señor",
             "你好", "世界", "こんにちは", "한국어", "Привет", "мир", "שלום", "مرحبا", "🙂", "🎵🎵", "3", "42", "2024", "3.14", "1,000",
             "e-mail", "co-op", "U.S.A.", "...", "--", "?!", "(", ")", "[music]", "♪", "%", "$5", "#tag", "@you", "a_b", "x=y",
             "antidisestablishmentarianism", "supercalifragilisticexpialidocious", "ＡＢＣ", "\t", "\n", "\r\n"]
    seps = [" ", " ", " ", "  ", "   ", "", "\n", " \n ", "\t"]
    out = ["", " ", "  ", "\n", " a", "a ", " a ", "a  b", "'s", " 's", "''", "' '"]
    for _ in range(n):
        k = rng.randint(1, 12)
        s = "".join(rng.choice(words) + rng.choice(seps) for _ in range(k))
        if rng.random() < 0.3:
            s = " " + s
        if rng.random() < 0.2:
            s = "".join(chr(rng.choice([rng.randint(32, 126), rng.randint(160, 0x24F), rng.randint(0x4E00, 0x4E80),
                                        rng.randint(0x1F600, 0x1F640)])) for _ in range(rng.randint(1, 24)))
        out.append(s)
    return out


@pytest.mark.parametrize("name", ["gpt2", "multilingual"])
def test_encode_matches_an_independent_bpe(name, vocab_dir, monkeypatch):
    path = f"{vocab_dir}/{name}.tiktoken"
    theirs = _independent(path, monkeypatch)
    ranks = _load_ranks(path)
    mine = wtok.BpeEncoding(ranks, wtok.special_token_names(99), name=name)
    n_tok = 0
    for text in _texts(7 if name == "gpt2" else 8, 3000):
        want = theirs.encode(text, add_special_tokens=False).ids
        got = mine.encode(text)
        assert got == want, repr(text)
        assert mine.decode(got) == text
        n_tok += len(got)
    assert n_tok > 20000
