"""Silero VAD gate on the GPU (wlk_vad_*) against what the REFERENCE computed with its own checkpoint
(tests/golden/vad_cases.npz, scripts/gen_golden_vad.py): per-window speech probability within 2e-5 (fp32, different
accumulation order than torch's conv1d), identical iterator events, identical results for any batching of windows."""
import numpy as np
import pytest

import helpers as H
from test_vad_host import META, GOLD, WEIGHTS, case_audio, feed
from whisperlivekit_amd import _lib
from whisperlivekit_amd import vad as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    w = V.HipSileroVADWeights(WEIGHTS)
    yield w
    w.close()


@pytest.mark.parametrize("name", sorted(META))
def test_probabilities_state_and_events_match_the_reference(weights, name):
    audio = case_audio(name)
    n = len(audio) // 512
    model = V.HipSileroVAD(weights, max_windows=64)
    probs = model.probs(audio[: n * 512])
    err = float(np.abs(probs - GOLD[name + "_probs"]).max())
    h, c = model.state()
    serr = float(np.abs(np.stack([h, c]) - GOLD[name + "_state"][:, 0, :]).max())
    assert err <= 2e-5 and serr <= 1e-4, (err, serr)
    it = V.HipFixedVADIterator(model)                 # resets the stream
    events, per_call = feed(it, audio, META[name]["chunking"])
    assert events == META[name]["events"] and per_call == META[name]["events_per_call"]
    model.close()


def test_any_batching_of_windows_gives_the_same_bits(weights):
    audio = case_audio("gaps")[: 96 * 512]
    a, b, c = (V.HipSileroVAD(weights, max_windows=m) for m in (96, 7, 1))
    pa = a.probs(audio)
    pb = b.probs(audio)
    pc = np.array([c(audio[i:i + 512]).item() for i in range(0, len(audio), 512)], np.float32)
    assert np.array_equal(pa, pb) and np.array_equal(pa, pc)
    a.reset_states()
    assert np.array_equal(a.probs(audio[: 10 * 512]), pa[:10])
    for m in (a, b, c):
        m.close()


def test_argument_errors(weights):
    m = V.HipSileroVAD(weights, max_windows=4)
    with pytest.raises(ValueError):
        m.probs(np.zeros(100, np.float32))
    with pytest.raises(ValueError):
        m(np.zeros(512, np.float32), 8000)
    with pytest.raises(ValueError):
        m(np.zeros(256, np.float32), 16000)
    m.close()
    with pytest.raises(_lib.WlkError):
        V.HipSileroVAD(weights, max_windows=0)
