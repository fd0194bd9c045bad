"""Host-side tokenizer for the AlignAtt path, without the ``tiktoken`` dependency.

Mirrors the surface the hot path consumes from the reference's ``Tokenizer``
(whisperlivekit/whisper/tokenizer.py:130-332): special-token ids, ``encode``/``decode``,
``split_to_word_tokens``.  Two encodings back it:

* :class:`BpeEncoding` - byte-pair encoding over a ``*.tiktoken`` rank file (one
  ``base64(token) rank`` pair per line, the format the reference ships under
  ``whisper/assets``).  The files are not copied here; pass ``vocab_path`` or set
  ``WLK_VOCAB_DIR``, or install WhisperLiveKit next to this package.
* :class:`SyntheticEncoding` - a deterministic stand-in vocabulary of the same size
  (256 byte tokens + generated word pieces).  Random-weight parity runs and the benchmark
  use it on machines that have no vocabulary file; token ids - the thing parity is judged
  on - do not depend on which byte strings the ids map to.  It is strictly OPT-IN
  (``synthetic=True`` or ``WLK_SYNTHETIC_VOCAB=1``): a deployment with a real checkpoint and
  no rank file must fail loudly instead of decoding ids to made-up word pieces.

Special-token numbering follows whisperlivekit/whisper/tokenizer.py:335-368.
"""
from __future__ import annotations

import base64
import os
import string
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

LANGUAGE_CODES: Tuple[str, ...] = tuple(
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no "
    "th ur hr bg lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw "
    "gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl "
    "mg as tt haw ln ha ba jw su yue".split()
)

_GPT2_SPLIT = (r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")


def special_token_names(num_languages: int) -> List[str]:
    names = ["<|endoftext|>", "<|startoftranscript|>"]
    names += [f"<|{code}|>" for code in LANGUAGE_CODES[:num_languages]]
    names += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>",
              "<|nospeech|>", "<|notimestamps|>"]
    names += [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
    return names


class Encoding:
    """Common part: id <-> bytes tables plus appended special tokens."""

    name = "encoding"

    def __init__(self, token_bytes: Sequence[bytes], specials: Sequence[str]):
        self._token_bytes: List[bytes] = list(token_bytes)
        self.n_base = len(self._token_bytes)
        self.special_tokens: Dict[str, int] = {s: self.n_base + i for i, s in enumerate(specials)}
        self._special_by_id = {v: k for k, v in self.special_tokens.items()}
        self.n_vocab = self.n_base + len(specials)

    # -- tiktoken-compatible surface -------------------------------------------------
    @property
    def special_tokens_set(self):
        return set(self.special_tokens)

    @property
    def eot_token(self) -> int:
        return self.special_tokens["<|endoftext|>"]

    def encode_single_token(self, text: str) -> int:
        if text in self.special_tokens:
            return self.special_tokens[text]
        ids = self.encode(text)
        if len(ids) != 1:
            raise KeyError(text)
        return ids[0]

    def decode_bytes(self, ids: Iterable[int]) -> bytes:
        out = bytearray()
        for t in ids:
            if t < self.n_base:
                out += self._token_bytes[t]
            else:
                out += self._special_by_id[t].encode("utf-8")
        return bytes(out)

    def decode(self, ids: Iterable[int], errors: str = "replace") -> str:
        return self.decode_bytes(ids).decode("utf-8", errors=errors)

    def encode(self, text: str, **_kw) -> List[int]:  # pragma: no cover - abstract
        raise NotImplementedError

    def _refuse_special_text(self, text: str) -> None:
        """tiktoken's ``encode`` default (disallowed_special="all") raises when the text spells a special token;
        the reference relies on that default everywhere on this path (whisper/tokenizer.py:161-162)."""
        if "<|" in text:
            import re
            for m in re.finditer(r"<\|[^|<>]*\|>", text):
                if m.group(0) in self.special_tokens:
                    raise ValueError(f"Encountered text corresponding to disallowed special token {m.group(0)!r}.")


class BpeEncoding(Encoding):
    """Rank-ordered byte-pair merges over GPT-2 style pre-split pieces."""

    def __init__(self, ranks: Dict[bytes, int], specials: Sequence[str], name: str = "bpe"):
        by_rank = sorted(ranks.items(), key=lambda kv: kv[1])
        assert [r for _, r in by_rank] == list(range(len(by_rank))), "ranks must be dense"
        super().__init__([b for b, _ in by_rank], specials)
        self._ranks = ranks
        self.name = name
        import regex  # third-party `regex` (unicode classes); present in the image
        self._pat = regex.compile(_GPT2_SPLIT)

    def _bpe(self, piece: bytes) -> List[int]:
        if piece in self._ranks:
            return [self._ranks[piece]]
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self._ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best_rank is None:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        return [self._ranks[p] for p in parts]

    def encode(self, text: str, **_kw) -> List[int]:
        self._refuse_special_text(text)
        out: List[int] = []
        for piece in self._pat.findall(text):
            out.extend(self._bpe(piece.encode("utf-8")))
        return out


def _letters(n: int) -> str:
    """bijective base-26 spelling, at least two letters."""
    n += 26
    s = ""
    while True:
        s = chr(ord("a") + n % 26) + s
        n = n // 26 - 1
        if n < 0:
            break
    return s


class SyntheticEncoding(Encoding):
    """Deterministic stand-in vocabulary: ids 0..255 are the single bytes; id 256+2j is
    ``" " + letters(j)`` (a word start) and id 257+2j is ``letters(j)`` (a continuation)."""

    def __init__(self, n_base: int, specials: Sequence[str], name: str = "synthetic"):
        toks = [bytes([i]) for i in range(256)]
        j = 0
        while len(toks) < n_base:
            w = _letters(j).encode()
            toks.append(b" " + w)
            if len(toks) < n_base:
                toks.append(w)
            j += 1
        super().__init__(toks, specials)
        self.name = name
        self._lookup = {b: i for i, b in enumerate(toks)}
        self._max_len = max(len(b) for b in toks)

    def encode(self, text: str, **_kw) -> List[int]:
        self._refuse_special_text(text)
        data = text.encode("utf-8")
        out: List[int] = []
        pos = 0
        while pos < len(data):
            for ln in range(min(self._max_len, len(data) - pos), 0, -1):
                tid = self._lookup.get(data[pos:pos + ln])
                if tid is not None:
                    out.append(tid)
                    pos += ln
                    break
        return out


def load_tiktoken_ranks(path: str) -> Dict[bytes, int]:
    ranks: Dict[bytes, int] = {}
    with open(path, "rb") as fh:
        for line in fh:
            line = line.strip()
            if line:
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    return ranks


def find_vocab_file(name: str, vocab_path: Optional[str] = None) -> Optional[str]:
    """Locate ``<name>.tiktoken``: explicit path, $WLK_VOCAB_DIR, or an installed WhisperLiveKit."""
    cands = []
    if vocab_path:
        cands.append(vocab_path if vocab_path.endswith(".tiktoken")
                     else os.path.join(vocab_path, f"{name}.tiktoken"))
    if os.environ.get("WLK_VOCAB_DIR"):
        cands.append(os.path.join(os.environ["WLK_VOCAB_DIR"], f"{name}.tiktoken"))
    try:
        import importlib.util
        spec = importlib.util.find_spec("whisperlivekit")
        if spec and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "whisper", "assets",
                                      f"{name}.tiktoken"))
    except (ImportError, ValueError):
        pass
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def synthetic_vocab_requested() -> bool:
    return os.environ.get("WLK_SYNTHETIC_VOCAB", "") not in ("", "0")


@lru_cache(maxsize=None)
def get_encoding(name: str = "gpt2", num_languages: int = 99, vocab_path: Optional[str] = None,
                 synthetic: Optional[bool] = None) -> Encoding:
    """``name`` is "gpt2" (.en models, 50256 base tokens) or "multilingual" (50257).

    ``synthetic=None`` (the default) means: the stand-in vocabulary only if ``WLK_SYNTHETIC_VOCAB=1`` is set
    (tests, bench.py, the GPU box), otherwise the real rank file - and FileNotFoundError if there is none."""
    specials = special_token_names(num_languages)
    if synthetic is None:
        synthetic = synthetic_vocab_requested()
    if synthetic:
        n_base = 50256 if name == "gpt2" else 50257
        return SyntheticEncoding(n_base, specials, name=f"synthetic-{name}")
    path = find_vocab_file(name, vocab_path)
    if path is None:
        raise FileNotFoundError(
            f"no {name}.tiktoken rank file found: pass vocab_path, set WLK_VOCAB_DIR, or install WhisperLiveKit "
            "(whisper/assets); WLK_SYNTHETIC_VOCAB=1 selects the stand-in vocabulary of random-weight test runs")
    return BpeEncoding(load_tiktoken_ranks(path), specials, name=os.path.basename(path))


class WhisperTokenizer:
    """Special-token bookkeeping + word splitting on top of an :class:`Encoding`."""

    def __init__(self, encoding: Encoding, num_languages: int, language: Optional[str] = None,
                 task: Optional[str] = None):
        self.encoding = encoding
        self.num_languages = num_languages
        self.language = language
        self.task = task
        sp = encoding.special_tokens
        self.special_tokens = dict(sp)
        self.eot = encoding.eot_token
        self.sot = sp["<|startoftranscript|>"]
        self.translate = sp["<|translate|>"]
        self.transcribe = sp["<|transcribe|>"]
        self.sot_lm = sp["<|startoflm|>"]
        self.sot_prev = sp["<|startofprev|>"]
        self.no_speech = sp["<|nospeech|>"]
        self.no_timestamps = sp["<|notimestamps|>"]
        self.timestamp_begin = sp["<|0.00|>"]
        seq = [self.sot]
        if language is not None:
            seq.append(self.sot + 1 + LANGUAGE_CODES[:num_languages].index(language))
        if task is not None:
            seq.append(self.transcribe if task == "transcribe" else self.translate)
        self.sot_sequence = tuple(seq)
        self.sot_sequence_including_notimestamps = tuple(seq + [self.no_timestamps])
        self.all_language_tokens = tuple(
            sp[f"<|{c}|>"] for c in LANGUAGE_CODES[:num_languages])
        self.all_language_codes = tuple(LANGUAGE_CODES[:num_languages])

    # -- text <-> ids ------------------------------------------------------------------
    def encode(self, text: str, **kw) -> List[int]:
        return self.encoding.encode(text, **kw)

    def decode(self, token_ids: Sequence[int], **kw) -> str:
        return self.encoding.decode([t for t in token_ids if t < self.timestamp_begin], **kw)

    def decode_with_timestamps(self, token_ids: Sequence[int], **kw) -> str:
        return self.encoding.decode(token_ids, **kw)

    # -- annotation symbols the batch decoder suppresses (whisper/tokenizer.py:241-275) -----
    _ANNOTATION_CHARS = '"#()*+/:;<=>@[\\]^_`{|}~「」『』'
    _ANNOTATION_RUNS = ("<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", '("', "((", "))", "(((", ")))", "[[", "]]",
                        "{{", "}}", "♪♪", "♪♪♪")
    _MUSIC_CHARS = "♩♪♫♬♭♮♯"       # U+2640..U+267F: one token or several sharing the first (the first is suppressed)

    @property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """Ids of speaker tags / non-speech annotations ("♪♪♪", "[DAVID]", "( SPEAKING ... )"): every listed symbol that
        is a single token with or without a leading space, the first token of the music symbols either way, and the
        word-initial " -" / " '" (hyphens and apostrophes stay allowed inside words)."""
        cached = self.__dict__.get("_non_speech_tokens")
        if cached is None:
            ids = {self.encoding.encode(" -")[0], self.encoding.encode(" '")[0]}
            for sym in [*self._ANNOTATION_CHARS, *self._ANNOTATION_RUNS, *self._MUSIC_CHARS]:
                for spelled in (sym, " " + sym):
                    toks = self.encoding.encode(spelled)
                    if len(toks) == 1 or sym in self._MUSIC_CHARS:
                        ids.add(toks[0])
            cached = self.__dict__["_non_speech_tokens"] = tuple(sorted(ids))
        return cached

    # -- word grouping (whisper/tokenizer.py:277-332) ----------------------------------
    def split_to_word_tokens(self, tokens: Sequence[int]):
        if self.language in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)

    def split_tokens_on_unicode(self, tokens: Sequence[int]):
        """Cut wherever the bytes decoded so far form complete code points."""
        full = self.decode_with_timestamps(tokens)
        bad = "�"
        words: List[str] = []
        groups: List[List[int]] = []
        pending: List[int] = []
        consumed = 0
        for tok in tokens:
            pending.append(tok)
            text = self.decode_with_timestamps(pending)
            at = text.find(bad)
            # a replacement char is acceptable only if the full decoding has one there too
            if at < 0 or (consumed + at < len(full) and full[consumed + at] == bad):
                words.append(text)
                groups.append(pending)
                pending = []
                consumed += len(text)
        return words, groups

    def split_tokens_on_spaces(self, tokens: Sequence[int]):
        pieces, piece_tokens = self.split_tokens_on_unicode(tokens)
        words: List[str] = []
        groups: List[List[int]] = []
        for piece, toks in zip(pieces, piece_tokens):
            starts_word = (toks[0] >= self.eot or piece.startswith(" ")
                           or piece.strip() in string.punctuation or not words)
            if starts_word:
                words.append(piece)
                groups.append(toks)
            else:
                words[-1] += piece
                groups[-1].extend(toks)
        return words, groups


def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language: Optional[str] = None,
                  task: Optional[str] = None, vocab_path: Optional[str] = None,
                  synthetic: Optional[bool] = None) -> WhisperTokenizer:
    """Same defaulting as the reference's ``get_tokenizer`` (whisper/tokenizer.py:371-400)."""
    if synthetic is None:
        synthetic = synthetic_vocab_requested()
    if vocab_path is None and not synthetic:
        vocab_path = os.environ.get("WLK_VOCAB_DIR") or None     # part of the cache key: tests switch directories
    return _get_tokenizer(multilingual, num_languages, language, task, vocab_path, bool(synthetic))


@lru_cache(maxsize=None)
def _get_tokenizer(multilingual: bool, num_languages: int, language: Optional[str], task: Optional[str],
                   vocab_path: Optional[str], synthetic: bool) -> WhisperTokenizer:
    if language is not None:
        language = language.lower()
        if language not in LANGUAGE_CODES:
            raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        name, language, task = "multilingual", language or "en", task or "transcribe"
    else:
        name, language, task = "gpt2", None, None
    enc = get_encoding(name, num_languages, vocab_path, synthetic)
    return WhisperTokenizer(enc, num_languages, language, task)
