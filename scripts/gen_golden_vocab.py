#!/usr/bin/env python
"""tests/golden/vocab_gpt2.npz: the GPT-2 byte-pair rank table the reference vendors as
whisperlivekit/whisper/assets/gpt2.tiktoken, re-packed (token bytes in rank order: `lengths` uint16 + `blob` uint8)
so the real-vocabulary parity tests (a17: BpeEncoding, word splitting, pending UTF-8 on real byte sequences) can run
where the reference tree is absent (the GPU box).  Run in the build container only:

    python scripts/gen_golden_vocab.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from whisperlivekit_amd.tokenizer import load_tiktoken_ranks  # noqa: E402

REF = os.environ.get("WLK_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

for name in ("gpt2",):
    ranks = load_tiktoken_ranks(os.path.join(REF, "whisperlivekit", "whisper", "assets", f"{name}.tiktoken"))
    toks = [b for b, _ in sorted(ranks.items(), key=lambda kv: kv[1])]
    assert sorted(ranks.values()) == list(range(len(toks)))
    lengths = np.array([len(b) for b in toks], np.uint16)
    blob = np.frombuffer(b"".join(toks), np.uint8)
    np.savez_compressed(os.path.join(OUT, f"vocab_{name}.npz"), lengths=lengths, blob=blob)
    print(name, len(toks), "tokens,", blob.size, "bytes")
