"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/wlk_hip.h declares,
its layout queries agree with the host packer, and it fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from whisperlivekit_amd import _lib, engine, synth
from whisperlivekit_amd.dims import MODEL_DIMS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "wlk_hip.h")).read()
    declared = set(re.findall(r"\b(wlk_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/wlk_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("name", ["micro.en", "tiny.en", "base.en", "large-v3"])
def test_packed_layout_matches_host_packer(name):
    dims = MODEL_DIMS[name]
    names = engine.packed_tensor_names(dims)
    assert len(names) == len(set(names))
    lib = _lib.load()
    import ctypes as C
    cd = engine._cdims(dims)
    sizes = {}
    end = 0
    for n in names:
        off, numel = C.c_uint64(), C.c_uint64()
        _lib.check(lib.wlk_tensor_lookup(C.byref(cd), n.encode(), C.byref(off), C.byref(numel)))
        assert off.value % 64 == 0 and off.value >= end
        end = off.value + numel.value
        sizes[n] = numel.value
    assert end <= engine.arena_floats(dims)
    if name in ("micro.en", "tiny.en"):
        packed = engine.pack_state_dict(dims, synth.synth_state_dict(dims, 0))
        assert set(packed) == set(names)
        for n in names:
            assert packed[n].size == sizes[n], n


def test_conv_and_qkv_repacking():
    dims = MODEL_DIMS["micro.en"]
    sd = synth.synth_state_dict(dims, 3)
    p = engine.pack_state_dict(dims, sd)
    d = dims.n_audio_state
    w = p["enc.conv1.w"].reshape(d, 3, dims.n_mels)
    assert np.array_equal(w[5, 2, 7], sd["encoder.conv1.weight"][5, 7, 2])
    qkv = p["dec.1.qkv.w"].reshape(3 * d, d)
    assert np.array_equal(qkv[d:2 * d], sd["decoder.blocks.1.attn.key.weight"])
    assert not p["dec.1.qkv.b"].reshape(3, d)[1].any()          # key has no bias (whisper/model.py:89)
    assert np.array_equal(p["dec.0.xkv.b"].reshape(2, d)[1], sd["decoder.blocks.0.cross_attn.value.bias"])


def test_unknown_tensor_and_bad_dims_are_errors():
    import ctypes as C
    lib = _lib.load()
    cd = engine._cdims(MODEL_DIMS["micro.en"])
    assert lib.wlk_tensor_lookup(C.byref(cd), b"nope", None, None) != 0
    assert b"nope" in lib.wlk_last_error()
    bad = _lib.Dims(80, 1500, 100, 2, 2, 51864, 448, 100, 2, 2)
    n = C.c_uint64()
    assert lib.wlk_arena_floats(C.byref(bad), C.byref(n)) != 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "whisperlivekit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.skipif(_lib.load().wlk_device_count() > 0, reason="GPU present")
def test_fails_loudly_without_gpu():
    with pytest.raises(_lib.WlkError):
        engine.HipWhisperModel(MODEL_DIMS["micro.en"])
