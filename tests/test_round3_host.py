"""Host-side contracts added in round 3 (CPU only): importing the package does not touch the process environment, the
hardware-queue knob is explicit, the reference-staging recipe produces an importable archive, bench.py names its golden
streams and fails on parity, new C-ABI entries validate their arguments without a GPU."""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import tarfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_importing_the_package_leaves_the_environment_alone():
    code = ("import os; before = dict(os.environ); import whisperlivekit_amd._lib, whisperlivekit_amd.backend, __graft_entry__; "
            "changed = {k for k in set(before) | set(os.environ) if before.get(k) != os.environ.get(k)}; print(sorted(changed))")
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "WLK_SYNTHETIC_VOCAB")}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == "[]", out


def test_hw_queue_knob_is_explicit(monkeypatch):
    from whisperlivekit_amd import _lib
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with pytest.raises(ValueError):
        _lib.configure_hw_queues(0)
    if _lib._lib is None:                       # the HIP library is not loaded yet in this process: the knob may set it
        assert _lib.configure_hw_queues(2) is True
        assert os.environ["GPU_MAX_HW_QUEUES"] == "2"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert _lib.configure_hw_queues(2) is False  # an exported value wins
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4"


@pytest.mark.skipif(not os.path.isdir("/root/reference/whisperlivekit"), reason="needs the reference tree (build container)")
def test_stage_reference_archive_is_complete_and_reproducible(tmp_path):
    from oracle import stage_reference as sr
    a = sr.stage("/root/reference", verbose=False)
    first = open(a, "rb").read()
    sr.stage("/root/reference", verbose=False)
    assert open(a, "rb").read() == first                     # same tree -> same bytes
    with tarfile.open(a) as tar:
        names = tar.getnames()
    for need in ("whisperlivekit/simul_whisper/backend.py", "whisperlivekit/simul_whisper/align_att_base.py",
                 "whisperlivekit/whisper/model.py", "whisperlivekit/whisper/audio.py", "whisperlivekit/whisper/assets/mel_filters.npz"):
        assert need in names, need
    assert not any("silero_vad_models" in n or n.endswith(".pyc") for n in names)
    root = sr.unpack(str(tmp_path))
    code = ("import sys, os; sys.path.insert(0, os.path.join(%r, 'scripts')); os.environ['WLK_REFERENCE_ROOT'] = %r; "
            "import ref_stubs; ref_stubs.install(); import whisperlivekit.simul_whisper.backend as b; print(b.__file__)" % (ROOT, root))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, check=True).stdout
    assert out.strip().startswith(str(tmp_path)), out      # imported from the unpacked archive, not from /root/reference


def test_the_staged_archive_is_not_tracked():
    ignored = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in ignored
    runignore = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(runignore) or "oracle/_ref" not in open(runignore).read()


def test_bench_golden_names_and_parity_gate():
    sys.path.insert(0, ROOT)
    import bench
    importlib.reload(bench)
    ns = lambda **k: type("A", (), dict(dict(audio="speech", seconds=30.0, model="base.en", no_parity=False), **k))()
    assert bench.golden_stem(ns()) == "stream_bench_base_30s"
    assert bench.golden_stem(ns(model="large-v3", seconds=10.0)) == "stream_bench_large-v3_10s"
    assert bench.golden_stem(ns(audio="noise")) is None
    assert bench.golden_stem(ns(seconds=7.5)) is None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    for seed in range(8):
        assert helpers.golden_exists(f"stream_bench_base_30s_s{seed}.json")
    assert helpers.golden_exists("stream_bench_large-v3_10s_s0.json")
    # config 3 as the default bench line times it (round 6): 30 s, the word-committing audio seed
    assert bench.golden_stem(ns(model="large-v3", seconds=float(bench.LV3_SECONDS))) == "stream_bench_large-v3_30s"
    assert helpers.golden_exists(f"stream_bench_large-v3_{bench.LV3_SECONDS}s_s{bench.LV3_SEED}.json")
    g = helpers.golden_json(f"stream_bench_large-v3_{bench.LV3_SECONDS}s_s{bench.LV3_SEED}.json")
    words = [len(ev["tokens"]) for ev in g["events"] if ev["kind"] == "chunk"]
    assert len(words) == 60 and sum(words) >= 40 and sum(words[:bench.LV3_CPU_CHUNKS]) >= 1      # the CPU leg's prefix commits words
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "sys.exit(3)" in src and '"parity_ok": parity_ok' in src
    # the log tally counts swallowed calls as errors
    import logging
    t = bench._LogTally()
    t.emit(logging.LogRecord("x", logging.WARNING, "", 0, "[SimulStreaming guard] reset", (), None))
    t.emit(logging.LogRecord("x", logging.ERROR, "", 0, "SimulStreaming processing error: boom", (), None))
    assert (t.guard, t.errors) == (1, 1) and "boom" in t.first_error


def test_word_alignment_entry_points_validate_arguments_without_a_gpu():
    from whisperlivekit_amd import _lib
    lib = _lib.load()
    mel = np.zeros(8, np.float32)
    assert lib.wlk_encode_mel(None, mel.ctypes.data_as(C.POINTER(C.c_float)), 3000) != 0
    assert b"NULL" in lib.wlk_last_error()
    toks = np.zeros(4, np.int64)
    assert lib.wlk_find_alignment(None, toks.ctypes.data_as(C.POINTER(C.c_int64)), 4, 1, 50256, 3000, 1.0, None, None, None) != 0
