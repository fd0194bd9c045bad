"""Host half of the reference's word timestamps (whisper/timing.py:220-388) in whisperlivekit_amd/timing.py, against
known answers produced by the reference's own find_alignment / merge_punctuations / add_word_timestamps
(scripts/gen_golden_word_timing.py).  Also: the DTW oracle - and on the GPU the HIP kernel - on cost matrices that
really came out of the reference's z-scored, median-filtered attention."""
import copy

import numpy as np
import pytest

import helpers as H
import helpers
from oracle import timing_oracle
from whisperlivekit_amd import timing
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS

KAT = H.golden_json("word_timing_kat.json")


def as_timings(rows):
    return [timing.WordTiming(r["word"], list(r["tokens"]), r["start"], r["end"], r["probability"]) for r in rows]


def as_rows(alignment):
    return [dict(word=t.word, tokens=[int(x) for x in t.tokens], start=float(t.start), end=float(t.end),
                 probability=float(t.probability)) for t in alignment]


@pytest.mark.parametrize("case", KAT["find_alignment"], ids=lambda c: c["text"].strip()[:12])
def test_word_timings_from_the_reference_path(case):
    got = timing.word_timings(case["path"][0], case["path"][1], case["words"], case["word_tokens"],
                              case["text_token_probs"])
    want = case["timings"]
    assert [t.word for t in got] == [w["word"] for w in want]
    assert [t.tokens for t in got] == [w["tokens"] for w in want]
    np.testing.assert_array_equal([t.start for t in got], [w["start"] for w in want])
    np.testing.assert_array_equal([t.end for t in got], [w["end"] for w in want])
    # the probabilities were recomputed by a second forward pass of the generator: equal up to that pass's rounding
    np.testing.assert_allclose([t.probability for t in got], [w["probability"] for w in want], rtol=1e-5, atol=1e-9)


def test_word_timings_of_eot_only():
    assert timing.word_timings([0], [0], ["<eot>"], [[50256]], []) == []


@pytest.mark.parametrize("case", KAT["find_alignment"], ids=lambda c: c["text"].strip()[:12])
def test_dtw_oracle_on_reference_attention(case):
    x = np.array(case["matrix"], dtype=np.float32).reshape(case["matrix_shape"])
    np.testing.assert_array_equal(timing_oracle.dtw(x), np.array(case["path"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", KAT["find_alignment"], ids=lambda c: c["text"].strip()[:12])
def test_hip_dtw_on_reference_attention(case):
    x = np.array(case["matrix"], dtype=np.float32).reshape(case["matrix_shape"])
    np.testing.assert_array_equal(timing.dtw(x), np.array(case["path"]))


@pytest.mark.parametrize("i", range(len(KAT["merge"])))
def test_merge_punctuations(i):
    case = KAT["merge"][i]
    alignment = as_timings(case["before"])
    timing.merge_punctuations(alignment)
    assert as_rows(alignment) == case["after"]


@pytest.mark.parametrize("i", range(len(KAT["attach"])))
def test_attach_words(i):
    case = KAT["attach"][i]
    before = case["before"]
    segments = copy.deepcopy(before["segments"])
    per_segment = [[t for t in s["tokens"] if t < case["eot"]] for s in segments]
    timing.attach_words(segments, as_timings(before["alignment"]), per_segment,
                        last_speech_timestamp=before["last_speech_timestamp"])
    assert segments == case["after"]


def test_attach_words_without_segments():
    timing.attach_words([], [], [], last_speech_timestamp=0.0)


@pytest.mark.parametrize("case", KAT["find_alignment"], ids=lambda c: c["text"].strip()[:12])
def test_oracle_alignment_cost_matches_reference(case):
    """The arithmetic in front of the DTW (decoder pass, softmax over the content frames, z-score over tokens, median
    filter, head mean) restated on the oracle's encoder / decoder: the matrix the reference handed to dtw, and from it
    the same path and the same words."""
    import torch
    from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
    dims = MODEL_DIMS["micro.en"]
    sd = H.oracle_sd("micro.en", 0)
    mel = torch.from_numpy(np.random.default_rng(case["mel_seed"]).standard_normal((dims.n_mels, 3000)).astype(np.float32))
    cost, probs = timing_oracle.alignment_cost(sd, dims, ALIGNMENT_HEADS["micro.en"], mel, case["sot_sequence"],
                                               case["no_timestamps"], case["text_tokens"], case["eot"], case["num_frames"])
    want = np.array(case["matrix"], dtype=np.float32).reshape(case["matrix_shape"])
    assert cost.shape == want.shape
    np.testing.assert_allclose(cost, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(probs, case["text_token_probs"], rtol=1e-5, atol=1e-9)
    path = timing_oracle.dtw(cost)
    np.testing.assert_array_equal(path, np.array(case["path"]))
    got = timing.word_timings(path[0], path[1], case["words"], case["word_tokens"], probs)
    assert [(t.word, t.start, t.end) for t in got] == [(w["word"], w["start"], w["end"]) for w in case["timings"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", KAT["find_alignment"], ids=lambda c: c["text"].strip()[:12])
def test_hip_find_alignment_matches_the_reference(case):
    """The device half of find_alignment (wlk_encode_mel + wlk_find_alignment) on the seeded micro Whisper the reference
    ran: the cost matrix it handed to its dtw (2e-5), the SAME warping path, the token probabilities, and through the
    host tail the same words with the same times."""
    from whisperlivekit_amd.engine import HipWhisperModel
    dims = MODEL_DIMS["micro.en"]
    model = HipWhisperModel.from_state_dict(dims, helpers.synth_sd("micro.en", 0), ALIGNMENT_HEADS["micro.en"], device=0)
    sess = model.new_session(beam=1, batched=False)
    try:
        mel = np.random.default_rng(case["mel_seed"]).standard_normal((dims.n_mels, 3000)).astype(np.float32)
        sess.encode_mel(mel)
        tokens = [*case["sot_sequence"], case["no_timestamps"], *case["text_tokens"], case["eot"]]
        trace, probs, cost = sess.find_alignment(tokens, len(case["sot_sequence"]), case["eot"], case["num_frames"], want_cost=True)
        want = np.array(case["matrix"], dtype=np.float32).reshape(case["matrix_shape"])
        assert cost.shape == want.shape
        np.testing.assert_allclose(cost, want, rtol=0, atol=2e-5)
        np.testing.assert_array_equal(timing.backtrace(trace), np.array(case["path"]))
        np.testing.assert_allclose(probs, case["text_token_probs"], rtol=2e-4, atol=1e-9)

        class Tok:       # what find_alignment needs of a tokenizer, with the word split the reference computed
            sot_sequence, no_timestamps, eot = case["sot_sequence"], case["no_timestamps"], case["eot"]

            @staticmethod
            def split_to_word_tokens(_tokens):
                return case["words"], case["word_tokens"]

        got = timing.find_alignment(sess, Tok, case["text_tokens"], mel, case["num_frames"])
        assert [t.word for t in got] == [w["word"] for w in case["timings"]]
        np.testing.assert_array_equal([t.start for t in got], [w["start"] for w in case["timings"]])
        np.testing.assert_array_equal([t.end for t in got], [w["end"] for w in case["timings"]])
        np.testing.assert_allclose([t.probability for t in got], [w["probability"] for w in case["timings"]], rtol=2e-4)
        # the session keeps working as a streaming session afterwards: a fresh encode + prefill
        sess.append(np.zeros(16000, np.float32))
        sess.encode()
        sess.decode(np.array([[50257, 50362, 1000]]), first=True, sot_index=0)
        sess.select([], [], [], 2, 50)
    finally:
        sess.close()
        model.close()
