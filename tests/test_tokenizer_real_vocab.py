"""a17 on the REAL vocabulary: the product's tokenizer (whisperlivekit_amd/tokenizer.py: BpeEncoding over a
``*.tiktoken`` rank file, WhisperTokenizer word splitting) and prompt-context buffer (policy.TextContext) against

* public GPT-2 byte-pair known answers (the merges themselves);
* known answers of the reference's own Tokenizer / TokenBuffer over its vendored rank tables
  (tests/golden/tokenizer_kat.json, scripts/gen_golden_tokenizer.py): decode, split_to_word_tokens on spaces /
  unicode / incomplete UTF-8 tails, special-token ids, context trimming, the pending-token carry;
* whole streams the unmodified reference produced with the real vocabulary behind its tokenizer
  (stream_micro_realvocab*.json.gz), replayed through the product's host logic on CPU and, with -m gpu, end to end.

The rank tables travel as tests/golden/vocab_*.npz and are materialised in the on-disk format the product loads."""
import os

import pytest

import helpers as H
from whisperlivekit_amd import policy as P
from whisperlivekit_amd import tokenizer as T


@pytest.fixture()
def real_vocab(tmp_path, monkeypatch):
    d = H.real_vocab_dir(tmp_path)
    monkeypatch.delenv("WLK_SYNTHETIC_VOCAB", raising=False)
    monkeypatch.setenv("WLK_VOCAB_DIR", d)
    T.get_encoding.cache_clear()
    T._get_tokenizer.cache_clear()
    yield d
    T.get_encoding.cache_clear()
    T._get_tokenizer.cache_clear()


GPT2_KATS = [   # public GPT-2 encodings (openai/gpt-2 encoder.json + vocab.bpe)
    ("Hello, world!", [15496, 11, 995, 0]),
    ("hello world", [31373, 995]),
    ("The quick brown fox jumps over the lazy dog.", [464, 2068, 7586, 21831, 18045, 625, 262, 16931, 3290, 13]),
    (" ", [220]),
    ("\n", [198]),
]


def test_bpe_merges_match_public_gpt2_known_answers(real_vocab):
    tok = T.get_tokenizer(False, num_languages=99)
    assert isinstance(tok.encoding, T.BpeEncoding) and tok.encoding.n_base == 50256
    for text, ids in GPT2_KATS:
        assert tok.encode(text) == ids, text
        assert tok.decode(ids) == text
    for text in (" naïve café — 你好 🙂", "It's 9:45... don't"):
        assert tok.decode(tok.encode(text)) == text
    assert tok.eot == 50256 and tok.sot == 50257 and tok.no_timestamps == 50362 and tok.timestamp_begin == 50363


def test_split_and_decode_match_reference_tokenizer(real_vocab):
    kat = H.golden_json("tokenizer_kat.json")
    n_partial = 0
    for c in kat["split"]:
        tok = T.get_tokenizer(c["multilingual"], num_languages=99, language=c["language"],
                              task="transcribe" if c["multilingual"] else None)
        assert tok.decode(c["ids"]) == c["decode"], c["note"]
        assert tok.decode_with_timestamps(c["ids"]) == c["decode_ts"]
        words, groups = tok.split_to_word_tokens(c["ids"])
        assert words == c["words"] and [list(g) for g in groups] == c["groups"], (c["language"], c["note"], c["ids"])
        n_partial += int("�" in c["decode_ts"])
    assert n_partial >= 20 and len(kat["split"]) >= 300
    for key, sp in kat["specials"].items():
        multi, lang = key.split(":")
        tok = T.get_tokenizer(multi == "multi", num_languages=99, language=None if lang == "None" else lang,
                              task="transcribe" if multi == "multi" else None)
        for name in ("eot", "sot", "sot_prev", "sot_lm", "no_speech", "no_timestamps", "timestamp_begin", "transcribe",
                     "translate"):
            assert getattr(tok, name) == sp[name], (key, name)
        assert list(tok.sot_sequence) == sp["sot_sequence"]
        assert list(tok.sot_sequence_including_notimestamps) == sp["sot_sequence_including_notimestamps"]
        assert len(tok.all_language_tokens) == sp["n_language_tokens"]
        # the reference builds this tuple by iterating a set of strings: only its content is defined
        assert sorted(tok.all_language_tokens) == sp["language_tokens_sorted"]
        assert tok.encode(" ") == sp["blank"]


def test_prompt_context_matches_reference_token_buffer(real_vocab):
    kat = H.golden_json("tokenizer_kat.json")
    tok = T.get_tokenizer(False, num_languages=99)
    for case in kat["buffer"]:
        ctx = P.TextContext(" ctx", tok, [tok.sot_prev])
        for st in case["steps"]:
            ctx.append_token_ids(list(st["append"]))
            assert ctx.text == st["text"] and ctx.pending_token_ids == st["pending"]
        for tr in case["trims"]:
            assert ctx.trim_words(after=4) == tr["dropped"]
            assert ctx.text == tr["text"] and ctx.as_token_ids() == tr["as_token_ids"]


def test_missing_rank_file_fails_loudly(monkeypatch, tmp_path):
    """A deployment without WLK_VOCAB_DIR / WhisperLiveKit must not silently decode ids to made-up word pieces."""
    monkeypatch.delenv("WLK_SYNTHETIC_VOCAB", raising=False)
    monkeypatch.setenv("WLK_VOCAB_DIR", str(tmp_path))          # empty directory
    monkeypatch.setattr(T, "find_vocab_file", lambda name, vocab_path=None: None)
    T.get_encoding.cache_clear()
    T._get_tokenizer.cache_clear()
    with pytest.raises(FileNotFoundError):
        T.get_tokenizer(False, num_languages=99)
    assert isinstance(T.get_tokenizer(False, num_languages=99, synthetic=True).encoding, T.SyntheticEncoding)
    monkeypatch.setenv("WLK_SYNTHETIC_VOCAB", "1")
    assert isinstance(T.get_tokenizer(False, num_languages=99).encoding, T.SyntheticEncoding)
    T.get_encoding.cache_clear()
    T._get_tokenizer.cache_clear()


def test_encode_refuses_special_token_text(real_vocab):
    """tiktoken's encode default raises on text that spells a special token (disallowed_special="all")."""
    tok = T.get_tokenizer(False, num_languages=99)
    with pytest.raises(ValueError):
        tok.encode(" before <|endoftext|> after")
    assert tok.encode(" <|notaspecial|> <| x |>")       # looks similar, is plain text
    with pytest.raises(ValueError):
        T.get_tokenizer(False, num_languages=99, synthetic=True).encode("<|startoftranscript|>")


REAL_STREAMS = ["micro_realvocab", "micro_realvocab_beam2"]


@pytest.mark.parametrize("case", REAL_STREAMS)
def test_product_host_logic_matches_reference_on_real_vocabulary(case, real_vocab):
    from test_oracle_golden import check_stream_against_golden, replay_stream
    from test_policy_golden import make_fake_processor
    g, proc, got = replay_stream(case, make_fake_processor)
    assert isinstance(proc.model.tokenizer.encoding, T.BpeEncoding)
    check_stream_against_golden(g, proc.trace, got)
    last = [ev for ev, _, _ in got if ev["kind"] == "chunk"][-1]
    assert proc.model.state.context.text == last["context"]
    assert sum(len(ev["tokens"]) for ev in g["events"]) >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("case", REAL_STREAMS)
def test_gpu_stream_matches_reference_on_real_vocabulary(case, real_vocab):
    from test_gpu_parity import make_hip_processor
    from test_oracle_golden import check_stream_against_golden, replay_stream
    g, proc, got = replay_stream(case, make_hip_processor)
    try:
        assert isinstance(proc.model.tokenizer.encoding, T.BpeEncoding)
        check_stream_against_golden(g, proc.trace, got, tol=1e-3, allow_ties=True)
    finally:
        proc.close()
