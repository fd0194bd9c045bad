#!/usr/bin/env python
"""tests/golden/alignment_heads.json: the reference's base85+gzip alignment-head masks
(whisperlivekit/whisper/__init__.py:39-54) decoded the way Whisper.set_alignment_heads does
(whisper/model.py:363-370) into (decoder layer, head) pairs in mask.to_sparse().indices() order.
Build container only:  python scripts/gen_golden_alignment_heads.py"""
import base64
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
import whisperlivekit.whisper as W  # noqa: E402
from whisperlivekit_amd.dims import MODEL_DIMS  # noqa: E402

out = {}
for name, dump in W._ALIGNMENT_HEADS.items():
    d = MODEL_DIMS[name]
    mask = np.frombuffer(gzip.decompress(base64.b85decode(dump)), dtype=bool).reshape(d.n_text_layer, d.n_text_head)
    out[name] = [[int(l), int(h)] for l, h in zip(*np.nonzero(mask))]
json.dump(out, open(os.path.join(os.path.dirname(HERE), "tests", "golden", "alignment_heads.json"), "w"), indent=0)
print({k: len(v) for k, v in out.items()})
