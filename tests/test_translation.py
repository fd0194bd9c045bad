"""Config 5 session glue (round 5): `HipOnlineTranslation` - the duck type AudioProcessor.translation_processor drives
(whisperlivekit/audio_processor.py:887-920) - over the NLLB network.  `nllw.OnlineTranslation`, which the reference
instantiates there (core.py:483-493), is third-party and absent from the reference tree, so these tests pin the CONTRACT:
the four calls, their return types, append-only validated text, punctuation-closed segments, the local-agreement rule and
the ValueError a per-session target relies on (translation.py:40-47) - on CPU over the oracle session, and on the GPU
over the library with identical outputs.  Every hypothesis is a `nllb.generate` result, which tests/test_nllb.py pins
against `transformers`."""
import types

import numpy as np
import pytest

from oracle.nllb_oracle import NllbOracle, OracleNllbSession
from whisperlivekit_amd import nllb
from whisperlivekit_amd import translation as T
from whisperlivekit_amd.policy import ASRToken

CFG = nllb.NLLB_MICRO
LANGS = {"eng_Latn": 1990, "fra_Latn": 1991, "deu_Latn": 1992}


class WordTokenizer:
    """Stand-in with the transformers tokenizer surface translation.py uses (no SentencePiece model exists offline):
    one id per lower-cased word (stable hash into the text range), [language code] words </s>."""
    unk_token_id = 3

    def __init__(self):
        self.src_lang = "eng_Latn"
        self.seen = {}

    def _id(self, w):
        h = 0
        for ch in w:
            h = (h * 131 + ord(ch)) % 1800
        return 10 + h

    def __call__(self, text):
        ids = [LANGS[self.src_lang]] + [self._id(w) for w in text.lower().split()] + [CFG.eos_token_id]
        return types.SimpleNamespace(input_ids=ids)

    def convert_tokens_to_ids(self, tok):
        return LANGS.get(tok, self.unk_token_id)

    def decode(self, ids, skip_special_tokens=True):
        special = {0, 1, 2, 3} | set(LANGS.values())
        return " ".join(f"w{int(i)}" for i in ids if not (skip_special_tokens and int(i) in special))


class OracleModel:
    """What HipNllbTranslationModel needs from a HipNllbModel, answered by the CPU oracle."""
    def __init__(self):
        self.cfg = CFG
        self.oracle = NllbOracle(CFG, nllb.synth_state_dict(CFG, 0))

    def new_session(self, rows=1):
        s = OracleNllbSession(self.oracle, rows)
        s.close = lambda: None
        return s


def words(spec, t0=0.0):
    """'the cat sat. on' -> ASRTokens of 0.4 s each with a leading space (as simul_whisper emits them)."""
    out = []
    for i, w in enumerate(spec.split()):
        out.append(ASRToken(start=round(t0 + 0.4 * i, 2), end=round(t0 + 0.4 * i + 0.4, 2), text=" " + w))
    return out


def drive(tm, script):
    """The calls of translation_processor (audio_processor.py:895-916) for a scripted session; -> (state, session)."""
    tr = tm.new_session("eng_Latn", "fra_Latn")
    state = types.SimpleNamespace(new_translation=[], new_translation_buffer=T.TimedText(), log=[])
    for kind, arg in script:
        new, buf = None, None
        if kind == "silence_start":
            new, buf = tr.validate_buffer_and_reset()
        elif kind == "silence_end":
            tr.insert_silence(arg)
            continue
        elif kind == "speaker":
            new, buf = tr.validate_buffer_and_reset()
        else:
            tr.insert_tokens(arg)
            new, buf = tr.process()
        if new is not None:
            assert isinstance(new, T.Translation)
            if new.text:
                state.new_translation.append(new)
            state.new_translation_buffer = buf
        assert isinstance(buf, T.TimedText)
        state.log.append((kind, None if new is None else (new.start, new.end, new.text), (buf.start, buf.end, buf.text)))
    return state, tr


SCRIPT = [("tokens", words("the quick brown")), ("tokens", words("fox jumps", 1.2)), ("tokens", words("over the lazy dog.", 2.0)),
          ("tokens", words("and then", 3.6)), ("tokens", words("it sleeps", 4.4)), ("silence_start", None), ("silence_end", 2.5),
          ("tokens", words("hello again friend", 9.0)), ("tokens", words("hello.", 10.2) + words("new sentence here", 10.6)),
          ("speaker", None), ("tokens", words("last words", 13.0)), ("tokens", [])]


def check_contract(state, tr, tm):
    pieces = state.new_translation
    assert pieces, "nothing was validated"
    for a, b in zip(pieces, pieces[1:]):                      # validated text is append-only and ordered in time
        assert b.start >= a.start - 1e-9 and b.end >= b.start - 1e-9 and abs(b.start - a.end) < 1e-9
    # a closed sentence's validated text has the length of its final translation (nothing behind the validated part is lost)
    full = tm.decode(nllb.generate(tr.session, tm.encode("the quick brown fox jumps over the lazy dog.", "eng_Latn"),
                                   LANGS["fra_Latn"], max_new_tokens=tm.max_new_tokens)).split()
    first_sentence = []
    for p in pieces:
        first_sentence += p.text.split()
        if p.end >= 3.6 - 1e-9:
            break
    assert len(first_sentence) == len(full) and first_sentence[-1] == full[-1]
    # silence start / speaker change hand out the buffer and leave none
    for kind, new, buf in state.log:
        if kind in ("silence_start", "speaker"):
            assert buf == (0, 0, "")
    assert tr._silence == 2.5 and tr.translations >= 8


def test_contract_over_the_oracle_session():
    tm = T.HipNllbTranslationModel(OracleModel(), WordTokenizer(), max_new_tokens=24)
    state, tr = drive(tm, SCRIPT)
    check_contract(state, tr, tm)


def test_local_agreement_validates_only_what_two_hypotheses_share():
    """Scripted hypotheses instead of a network: the rule itself."""
    tm = T.HipNllbTranslationModel(OracleModel(), WordTokenizer())
    tr = tm.new_session("eng_Latn", "fra_Latn")
    hyps = iter(["le chat", "le chat noir dort", "le chat noir mange ici", "un autre", "le chat noir mange ici maintenant."])
    tr._translate = lambda seg: next(hyps).split()
    out = []
    for spec in ("the cat", "black sleeps", "eats here", "x", "now."):
        tr.insert_tokens(words(spec, 0.4 * len(out)))
        new, buf = tr.process()
        out.append((None if new is None else new.text, buf.text))
    assert out == [(None, "le chat"),                      # first hypothesis: nothing to agree with
                   ("le chat", "noir dort"),               # agreed prefix validated, the rest is the buffer
                   ("noir", "mange ici"),
                   (None, ""),                              # a hypothesis that rewrites validated text validates nothing
                   ("mange ici maintenant.", "")]          # the sentence ended: its final translation behind the validated part
    assert tr._segment.tokens == [] and tr._validated_words == []


def test_final_translation_that_rewrites_the_validated_prefix_drops_and_repeats_nothing():
    """A sentence closes with a final hypothesis that changed (or shortened) the already validated prefix: what is emitted
    continues from where the final hypothesis and the validated words still agree - by content, not by count (round-5
    advisor finding: a slice by count dropped words when the final hypothesis was shorter and repeated none of the rewrite)."""
    tm = T.HipNllbTranslationModel(OracleModel(), WordTokenizer())
    for final, emitted in (("le chien noir dort.", "chien noir dort."),          # rewrote validated word 2: continue behind "le"
                           ("le chat.", "chat."),                                # shortened: only what differs behind the agreed "le"
                           ("le chat noir court vite.", "court vite.")):           # plain extension: only the new words
        tr = tm.new_session("eng_Latn", "fra_Latn")
        hyps = iter(["le chat noir", "le chat noir dort", final])
        tr._translate = lambda seg: next(hyps).split()
        got = []
        for spec in ("the cat", "black", "sleeps."):
            tr.insert_tokens(words(spec, 0.4 * len(got)))
            new, _buf = tr.process()
            got.append(None if new is None else new.text)
        assert got[1] == "le chat noir"
        assert got[2] == emitted, (final, got)       # never the count-based slice (that gave "dort." / nothing / "court vite.")
        assert tr._validated_words == [] and tr._segment.tokens == []


def test_unknown_language_raises_value_error_and_the_factory_falls_back():
    tm = T.HipNllbTranslationModel(OracleModel(), WordTokenizer())
    with pytest.raises(ValueError):
        T.HipOnlineTranslation(tm, ["eng_Latn"], ["xx_Nope"])
    with pytest.raises(ValueError):
        T.HipOnlineTranslation(tm, [], ["fra_Latn"])
    s = T.online_translation_factory(tm, "eng_Latn", "xx_Nope", fallback_target="deu_Latn")
    assert s.target_language == "deu_Latn" and s.target_id == LANGS["deu_Latn"]
    assert not T.HipOnlineTranslation.wants_hypothesis_tail


def test_hypothesis_tail_and_empty_items_are_ignored():
    tm = T.HipNllbTranslationModel(OracleModel(), WordTokenizer())
    tr = tm.new_session("eng_Latn", "fra_Latn")
    HypothesisTail = type("HypothesisTail", (), {"text": " draft", "start": 0.0, "end": 1.0})
    tr.insert_tokens([HypothesisTail(), ASRToken(0.0, 0.1, "  "), object()])
    assert tr._segment.tokens == [] and tr.process() == (None, T.TimedText())
    assert tr.translations == 0


@pytest.mark.gpu
def test_hip_session_equals_the_oracle_session():
    """The same scripted session over the HIP library and over the CPU oracle: identical validated pieces and buffers."""
    tok = WordTokenizer()
    ref_state, _ = drive(T.HipNllbTranslationModel(OracleModel(), tok, max_new_tokens=24), SCRIPT)
    model = nllb.HipNllbModel.synthetic(CFG, 0, device=0, max_src=92, max_tgt=64)
    try:
        tm = T.HipNllbTranslationModel(model, tok, max_new_tokens=24)
        state, tr = drive(tm, SCRIPT)
        check_contract(state, tr, tm)
        assert state.log == ref_state.log
        tr.close()
        # beams: the session owns a num_beams-row device session and runs nllb.beam_search
        tm3 = T.HipNllbTranslationModel(model, tok, num_beams=3, max_new_tokens=16)
        s3 = tm3.new_session("eng_Latn", "deu_Latn")
        s3.insert_tokens(words("a short sentence."))
        new, buf = s3.process()
        assert new is not None and new.text and buf.text == ""
        s3.close()
    finally:
        model.close()
