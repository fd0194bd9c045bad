"""a15: the model tables of the package against the reference's (whisper/__init__.py:20-54, whisper/model.py:363-370)."""
import os
import sys

import pytest

import helpers as H
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS

REFERENCE_NAMES = ["tiny.en", "tiny", "base.en", "base", "small.en", "small", "medium.en", "medium", "large-v1", "large-v2",
                   "large-v3", "large", "large-v3-turbo", "turbo"]


def test_alignment_heads_equal_the_decoded_reference_masks():
    """tests/golden/alignment_heads.json = the reference's base85+gzip masks decoded in to_sparse().indices() order
    (scripts/gen_golden_alignment_heads.py): the order is the alignment-head RANK AlignAtt uses."""
    ref = H.golden_json("alignment_heads.json")
    assert sorted(ref) == sorted(REFERENCE_NAMES)
    for name in REFERENCE_NAMES:
        assert [list(p) for p in ALIGNMENT_HEADS[name]] == ref[name], name
        d = MODEL_DIMS[name]
        assert all(0 <= l < d.n_text_layer and 0 <= h < d.n_text_head for l, h in ALIGNMENT_HEADS[name])
    assert ALIGNMENT_HEADS["large"] == ALIGNMENT_HEADS["large-v3"] and MODEL_DIMS["large"] == MODEL_DIMS["large-v3"]
    assert ALIGNMENT_HEADS["turbo"] == ALIGNMENT_HEADS["large-v3-turbo"] and MODEL_DIMS["turbo"].n_text_layer == 4


@pytest.mark.reference
def test_alignment_head_fixture_is_what_the_reference_decodes():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import ref_stubs
    if not ref_stubs.reference_available():
        pytest.skip("reference tree not present")
    ref_stubs.install()
    import base64
    import gzip
    import numpy as np
    import whisperlivekit.whisper as W
    assert sorted(W._ALIGNMENT_HEADS) == sorted(REFERENCE_NAMES) == sorted(W._MODELS)
    for name, dump in W._ALIGNMENT_HEADS.items():
        d = MODEL_DIMS[name]
        mask = np.frombuffer(gzip.decompress(base64.b85decode(dump)), dtype=bool).reshape(d.n_text_layer, d.n_text_head)
        assert [tuple(int(x) for x in p) for p in zip(*np.nonzero(mask))] == list(ALIGNMENT_HEADS[name]), name
