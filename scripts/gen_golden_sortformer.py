#!/usr/bin/env python
"""Pin the streaming Sortformer path against NeMo itself.

Runs on a box that has NeMo (``nemo_toolkit[asr]``), soundfile, the checkpoint
``nvidia/diar_streaming_sortformer_4spk-v2`` (revision 5240a64075176943f677d30fa2171c780229f341; the ``.nemo`` file has
SHA-256 b371afce2c4958186469df33d939936b9746c89f38b10a69cfd2c61254e83329) and a WhisperLiveKit tree - NOT in the build
container and not on the benchmark box (neither has NeMo).  One command:

    WLK_REFERENCE_ROOT=/path/to/WhisperLiveKit WLK_SORTFORMER_MODEL_PATH=/path/to/model.nemo \
        python scripts/gen_golden_sortformer.py [out.npz]

What it does: builds the 13 s two-speaker signal of the reference's own real-model test
(tests/test_sortformer_real_fixture.py: silence, utterance A, B, A again, an overlap of both at half gain; the two
LibriSpeech FLACs under tests/fixtures/sortformer_2spk/), drives the REFERENCE's SortformerDiarizationOnline over it in
0.5 s chunks with preprocessor dither forced to 0, and records per forward_streaming_step call, through forward hooks on
the NeMo modules: the log-mel features handed in, the pre-encode output (chunk embeddings), the Conformer encoder
output, the Transformer output, the sigmoid speaker activities, the speaker-cache / FIFO lengths after the update, and
at the end the emitted speaker segments.  The file (default tests/golden/sortformer_nemo.npz, ~2 MB) is consumed by
tests/test_gpu_sortformer.py::test_against_nemo_golden together with the same ``.nemo`` file: until it exists that
test skips and a12 stays "parity unpinned"."""
import asyncio
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("WLK_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)


def two_speaker_signal():
    """The derived signal of tests/test_sortformer_real_fixture.py:_build_two_speaker_signal (re-stated; asserted below
    against the reference's own helper when its test module is importable)."""
    import soundfile as sf
    fx = os.path.join(REF, "tests", "fixtures", "sortformer_2spk")
    a, sr_a = sf.read(os.path.join(fx, "6930-75918-0000.flac"), dtype="float32")
    b, sr_b = sf.read(os.path.join(fx, "7902-96591-0000.flac"), dtype="float32")
    assert sr_a == sr_b == 16000
    try:
        sys.path.insert(0, os.path.join(REF, "tests"))
        import test_sortformer_real_fixture as T
        signal, sr, windows = T._build_two_speaker_signal()
        return np.asarray(signal, np.float32), {k: [float(x) for x in v] for k, v in windows.items()}
    except Exception as e:      # the helper moved: fall back to the documented recipe
        print("reference helper not importable (%s); using the documented recipe" % e, file=sys.stderr)
        sil = np.zeros(int(0.5 * 16000), np.float32)
        n = min(len(a), len(b))
        overlap = 0.5 * a[:n] + 0.5 * b[:n]
        signal = np.concatenate([sil, a, sil, b, sil, a, sil, overlap, sil])
        return signal.astype(np.float32), {}


def main(out_path):
    import torch
    from whisperlivekit.diarization.sortformer_backend import SortformerDiarization, SortformerDiarizationOnline
    model_path = os.environ["WLK_SORTFORMER_MODEL_PATH"]
    shared = SortformerDiarization(model_path=model_path)
    online = SortformerDiarizationOnline(shared_model=shared)
    # no dither, no padding noise: the HIP front end is deterministic
    feat = online.audio2mel.featurizer
    feat.dither = 0.0
    if hasattr(feat, "pad_to"):
        feat.pad_to = 0
    m = shared.diar_model
    rec = {}
    calls = []

    def hook(name):
        def fn(_mod, _inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            calls[-1][name] = t.detach().float().cpu().numpy()
        return fn

    handles = [m.encoder.pre_encode.register_forward_hook(hook("pre_encode")),
               m.encoder.register_forward_hook(hook("fc_out")),
               m.transformer_encoder.register_forward_hook(hook("tf_out"))]
    orig_step = m.forward_streaming_step

    def step(processed_signal, processed_signal_length, streaming_state, total_preds, left_offset=0, right_offset=0, **kw):
        calls.append(dict(feats=processed_signal.detach().float().cpu().numpy()[0], left_offset=int(left_offset),
                          right_offset=int(right_offset)))
        st, tp = orig_step(processed_signal=processed_signal, processed_signal_length=processed_signal_length,
                           streaming_state=streaming_state, total_preds=total_preds, left_offset=left_offset,
                           right_offset=right_offset, **kw)
        calls[-1].update(total_preds=tp.detach().float().cpu().numpy()[0],
                         spkcache_len=int(st.spkcache.shape[1]) if st.spkcache is not None else 0,
                         fifo_len=int(st.fifo.shape[1]) if st.fifo is not None else 0)
        return st, tp

    m.forward_streaming_step = step
    signal, windows = two_speaker_signal()
    segments = []

    async def run():
        for lo in range(0, len(signal), 8000):
            online.insert_audio_chunk(signal[lo:lo + 8000])
            segments.extend(await online.diarize())

    asyncio.run(run())
    for h in handles:
        h.remove()
    rec["signal"] = signal
    rec["n_calls"] = np.int64(len(calls))
    for i, c in enumerate(calls):
        for k, v in c.items():
            rec[f"c{i}_{k}"] = np.asarray(v)
    rec["segments"] = np.asarray([[float(s.start), float(s.end), float(s.speaker)] for s in segments], np.float64)
    rec["windows"] = np.asarray([[k] + [str(x) for x in v] for k, v in windows.items()], dtype=object) if windows else np.zeros(0)
    np.savez_compressed(out_path, **{k: v for k, v in rec.items() if k != "windows"})
    print(f"{out_path}: {len(calls)} streaming steps, {len(segments)} segments, torch {torch.__version__}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden", "sortformer_nemo.npz"))
