#!/usr/bin/env python
"""GPU box: teacher-forced logits of the HIP NLLB session against the CPU oracle on the sentences of
tests/test_translation.py's script (where does a greedy hypothesis part, and by how much do the rows differ there?)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_translation as TT  # noqa: E402
from oracle.nllb_oracle import OracleNllbSession  # noqa: E402
from whisperlivekit_amd import nllb  # noqa: E402

om = TT.OracleModel()
model = nllb.HipNllbModel.synthetic(TT.CFG, 0, device=0, max_src=92, max_tgt=64)
tok = TT.WordTokenizer()
for text in ("the quick brown", "the quick brown fox jumps", "the quick brown fox jumps over the lazy dog.", "and then it sleeps"):
    src = tok(text).input_ids
    hs, os_ = model.new_session(1), OracleNllbSession(om.oracle, 1)
    ref = nllb.generate(os_, src, TT.LANGS["fra_Latn"], max_new_tokens=24)
    got = nllb.generate(hs, src, TT.LANGS["fra_Latn"], max_new_tokens=24)
    print(text, "| same" if ref == got else f"| DIFFER at {next(i for i, (a, b) in enumerate(zip(ref, got)) if a != b)}", len(ref), len(got))
    hs.encode(src); os_.encode(src)
    print("   encoder max diff", float(np.abs(hs.encoder_output() - os_.encoder_output()).max()))
    for i in range(len(ref) - 1):
        t = np.asarray([[ref[i]]], np.int64)
        hs.decode(t, first=(i == 0)); os_.decode(t, first=(i == 0))
        a, b = hs.logits()[0], os_.logits()[0]
        top = np.sort(b)[-2:]
        print(f"   step {i}: fed {ref[i]} max |hip - oracle| {float(np.abs(a - b).max()):.2e}  oracle margin {float(top[1] - top[0]):.2e}  argmax hip {int(a.argmax())} oracle {int(b.argmax())}")
    hs.close()
model.close()
