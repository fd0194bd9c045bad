#!/usr/bin/env python
"""Stream a 16 kHz mono s16le WAV file through the HIP backend in 0.5 s chunks, the way AudioProcessor would feed it
(optionally through the Silero VAD gate and the Sortformer diarizer), and print the committed words with timestamps.

    python scripts/transcribe_wav.py audio.wav --model-path base.en.pt            # openai-whisper checkpoint
    python scripts/transcribe_wav.py audio.wav --synthetic base.en                # seeded random weights (smoke runs)

Real text needs the reference's tiktoken rank files (WLK_VOCAB_DIR or an installed WhisperLiveKit); without them a real
checkpoint fails loudly.  --synthetic selects the stand-in vocabulary too (only ids / timestamps are meaningful)."""
import argparse
import sys
import time
import wave

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])


def read_wav(path):
    with wave.open(path, "rb") as w:
        if w.getframerate() != 16000 or w.getsampwidth() != 2:
            raise SystemExit("need a 16 kHz, 16-bit WAV (ffmpeg -ar 16000 -ac 1 -sample_fmt s16)")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        if w.getnchannels() > 1:
            pcm = pcm.reshape(-1, w.getnchannels())[:, 0].copy()
    return pcm


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("wav")
    ap.add_argument("--model-path")
    ap.add_argument("--synthetic", metavar="SIZE", help="seeded random weights of this size instead of a checkpoint")
    ap.add_argument("--model-size", default="base.en")
    ap.add_argument("--lan", default="en")
    ap.add_argument("--beams", type=int, default=1)
    ap.add_argument("--frame-threshold", type=int, default=25)
    ap.add_argument("--vad", action="store_true", help="gate with the Silero VAD (needs WLK_SILERO_VAD_JIT or WhisperLiveKit)")
    ap.add_argument("--chunk", type=float, default=0.5)
    args = ap.parse_args(argv)

    from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor
    if args.synthetic:
        import os
        os.environ.setdefault("WLK_SYNTHETIC_VOCAB", "1")
        asr = HipSimulStreamingASR(args.synthetic, synthetic_seed=0, lan=args.lan, beams=args.beams,
                                   frame_threshold=args.frame_threshold)
    elif args.model_path:
        asr = HipSimulStreamingASR(args.model_size, model_path=args.model_path, lan=args.lan, beams=args.beams,
                                   frame_threshold=args.frame_threshold)
    else:
        raise SystemExit("give --model-path or --synthetic")
    proc = HipSimulStreamingOnlineProcessor(asr)
    vad = None
    if args.vad:
        from whisperlivekit_amd.vad import HipFixedVADIterator, HipSileroVAD, HipSileroVADWeights
        vad = HipFixedVADIterator(HipSileroVAD(HipSileroVADWeights()))
    pcm = read_wav(args.wav)
    n = int(args.chunk * 16000)
    t0 = time.perf_counter()
    words = []
    for lo in range(0, len(pcm), n):
        chunk = pcm[lo:lo + n]
        if vad is not None:
            for ev in vad(chunk.astype(np.float32) / 32768.0):
                print(f"[vad] {ev}")
        proc.insert_pcm16_chunk(chunk, (lo + len(chunk)) / 16000)
        tokens, _ = proc.process_iter()
        for t in tokens:
            words.append(t)
            print(f"{t.start:7.2f} {t.end:7.2f}  {t.text!r}")
    tokens, _ = proc.process_iter(is_last=True)
    for t in tokens:
        words.append(t)
        print(f"{t.start:7.2f} {t.end:7.2f}  {t.text!r}")
    wall = time.perf_counter() - t0
    print(f"# {len(pcm) / 16000:.1f} s of audio, {len(words)} words, {wall:.2f} s wall, RTF {wall / (len(pcm) / 16000):.4f}",
          file=sys.stderr)
    proc.close()
    return words


if __name__ == "__main__":
    main()
