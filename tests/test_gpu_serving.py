"""GPU tests of the serving shape of the path (run with -m gpu on an MI355X):

* concurrency: 8 host threads, 8 sessions, ONE model - the reference's threading contract
  (whisperlivekit/audio_processor.py:543-551 runs process_iter on asyncio.to_thread workers concurrently across
  sessions with no model lock; simul_whisper/simul_whisper.py:109-114 "the model can be shared");
* the multi-GPU weight path: sharding.replicated_model (torch CUDA arena adopted by wlk_model_create) at world size 1
  and, when two devices are visible, over a real 2-rank RCCL broadcast;
* bench.py end to end: it must check its own outputs against the reference's golden streams, and `--gpus N` must
  refuse to run on fewer than N devices.
"""
import json
import os
import subprocess
import tempfile
import sys
import threading

import numpy as np
import pytest

import helpers as H
from whisperlivekit_amd import _lib, synth
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 8000


def _run_stream(proc, audio, chunk=CHUNK):
    words = []
    t_end = 0.0
    for lo in range(0, len(audio), chunk):
        t_end += len(audio[lo:lo + chunk]) / 16000
        proc.insert_audio_chunk(audio[lo:lo + chunk].copy(), t_end)
        toks, _ = proc.process_iter()
        words.append([(float(t.start), float(t.end), t.text) for t in toks])
    return words


def _new_proc(asr):
    from whisperlivekit_amd.backend import HipSimulStreamingOnlineProcessor
    p = HipSimulStreamingOnlineProcessor(asr)
    p.model.decision_log = []
    return p


@pytest.mark.parametrize("model_name,cases", [
    ("base.en", [f"bench_base_30s_s{i}" for i in range(8)]),
    ("micro.en", ["micro_12s", "micro_34s_evict", "micro_neverfire", "micro_12s", "micro_34s_evict", "micro_neverfire",
                  "micro_12s", "micro_12s"]),
])
def test_eight_threads_share_one_model(model_name, cases, monkeypatch):
    """8 threads replay 8 golden streams simultaneously on ONE HipWhisperModel: every session's decisions equal the
    reference's, and equal (bit for bit: token ids, frames, emitted words) what the same session yields when run alone.
    The prefill lane is opened from two busy sessions on (default nine) so that stacked prefills are part of the run."""
    monkeypatch.setenv("WLK_PREFILL_MIN_SESSIONS", "2")
    from whisperlivekit_amd.backend import HipSimulStreamingASR
    from whisperlivekit_amd.engine import HipWhisperModel
    for c in cases:
        if not H.golden_exists(f"stream_{c}.json"):
            pytest.skip(f"golden stream {c} not generated")
    goldens = [H.golden_json(f"stream_{c}.json") for c in cases]
    audios = [H.stream_audio(c) for c in cases]
    chunks = [g["chunk"] for g in goldens]
    cfgs = [H.asr_kwargs(g["cfg"]) for g in goldens]
    model = HipWhisperModel.synthetic(model_name, 0)
    asrs = [HipSimulStreamingASR(model_name, hip_model=model, **kw) for kw in cfgs]

    serial = []
    for asr, a, ch in zip(asrs, audios, chunks):
        p = _new_proc(asr)
        w = _run_stream(p, a, ch)
        serial.append((p.model.decision_log, w))
        p.close()

    for round_ in range(2):                 # second round: sessions created while others are mid-stream
        procs = [_new_proc(asr) for asr in asrs]
        start = threading.Barrier(len(procs))
        out = [None] * len(procs)
        errs = []

        def work(i):
            try:
                start.wait()
                out[i] = _run_stream(procs[i], audios[i], chunks[i])
            except Exception as e:          # pragma: no cover - reported below
                errs.append((i, repr(e)))

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(procs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        for i, p in enumerate(procs):
            assert getattr(p, "last_error", None) is None, (i, p.last_error)
            assert p.model.decision_log == serial[i][0], f"session {i}: concurrent decisions differ from the serial run"
            assert out[i] == serial[i][1], f"session {i}: concurrent words differ from the serial run"
            r = H.compare_decisions(goldens[i], p.model.decision_log, out[i])
            assert r["mismatch"] is None, (cases[i], r)
            if r["tie_divergence"] is None:
                assert r["identical"] == r["decisions"] and r["words_identical"], (cases[i], r)
            p.close()
    stats = model.engine_stats()
    assert stats["batched_steps"] > 0 and stats["mean_rows_per_batched_step"] >= 2.0, stats   # the loops really shared steps
    if MODEL_DIMS[model_name].n_text_state >= 256:      # narrower models never qualify for the stacked chain (k-wave GEMMs)
        assert stats["stacked_prefills"] > 0, stats
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"engine_stats_{model_name}.json"), "w") as fh:
        json.dump(stats, fh)
    model.close()


def test_replicated_model_world1_equals_uploaded_model():
    """sharding.replicated_model: the arena is a torch CUDA tensor (what RCCL broadcasts) adopted by
    wlk_model_create(arena_dev); logits / encoder output must equal the model that uploaded tensor by tensor."""
    import torch
    from whisperlivekit_amd import sharding
    from whisperlivekit_amd.engine import HipWhisperModel, pack_state_dict
    name = "tiny.en"
    dims = MODEL_DIMS[name]
    sd = synth.synth_state_dict(dims, 0)
    timing = {}
    a = sharding.replicated_model(dims, pack_state_dict(dims, sd), ALIGNMENT_HEADS[name], device=0, timing=timing)
    b = HipWhisperModel.from_state_dict(dims, sd, ALIGNMENT_HEADS[name], device=0)
    assert timing["arena_bytes"] > 0 and a._arena_keepalive.is_cuda
    audio = synth.to_pcm16_roundtrip(synth.speech_like(3.0, 3))
    outs = []
    for m in (a, b):
        s = m.new_session()
        s.append(audio)
        cml = s.encode()
        s.decode(np.array([[50257, 50362, 1000, 2000]]), first=True, sot_index=0)
        lp, ids, fr = s.select([], [], [], 2, cml)
        outs.append((cml, s.export("enc").copy(), s.export("logits_last").copy(), lp.copy(), ids.copy(), fr.copy()))
        s.close()
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    a.close(); b.close()
    torch.cuda.synchronize()


_RCCL_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from whisperlivekit_amd import sharding, synth
from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
from whisperlivekit_amd.engine import pack_state_dict
rank, world, local = sharding.init_process_group("nccl")
name = "tiny.en"
dims = MODEL_DIMS[name]
packed = pack_state_dict(dims, synth.synth_state_dict(dims, 0)) if rank == 0 else None
timing = {}
model = sharding.replicated_model(dims, packed, ALIGNMENT_HEADS[name], device=local, timing=timing)
s = model.new_session()
s.append(synth.to_pcm16_roundtrip(synth.speech_like(2.0, 1)))
cml = s.encode()
s.decode(np.array([[50257, 50362, 1000]]), first=True, sot_index=0)
logits = torch.from_numpy(s.export("logits_last").copy()).to(f"cuda:{local}")
gathered = [torch.empty_like(logits) for _ in range(world)]
dist.all_gather(gathered, logits)
same = all(bool(torch.equal(gathered[0], g)) for g in gathered)
if rank == 0:
    print("RCCL_OK" if same else "RCCL_DIFF", timing)
s.close(); model.close()
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if same else 3)
"""


def test_two_rank_rccl_broadcast_gives_identical_replicas(tmp_path):
    if _lib.device_count() < 2:
        pytest.skip("needs 2 visible GPUs (the driver's multi-GPU node)")
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_checks_its_own_outputs_against_the_reference():
    """bench.py replays the reference's golden decisions on the sessions it times (1-stream headline + the 8-stream
    leg) and prints parity_checked; nothing it reports may come from unverified outputs."""
    full_path = os.path.join(tempfile.mkdtemp(prefix="wlk_bench_"), "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-diarization", "--no-large-v3", "--full-out", full_path], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # the line the driver parses out of a bounded stdout tail: ONE line, last on stdout, <= 8 KB, carrying the judged blocks;
    # everything else is in the full record it names
    printed = r.stdout.strip().splitlines()[-1]
    assert len(printed) <= 8000, len(printed)
    short = json.loads(printed)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_checked", "parity_ok", "full"):
        assert key in short, key
    roof = short["roofline"]
    assert roof["bound"] in ("mfma", "hbm") and 0 < roof["frac"] <= 1.0 and abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-3
    assert all(0 < f["frac"] <= 1.0 for f in roof["families"].values()), roof
    assert short["parity_checked"]["sessions"] == 9 and short["parity_ok"] is True
    line = json.load(open(full_path))
    assert line["value"] == short["value"] and "kernels" in line and "launch_tags" in line
    pc = line["parity_checked"]
    assert pc is not None and pc["sessions"] == 9 and not pc["missing_golden"], pc
    assert pc["mismatches"] == [], pc
    assert pc["decisions"] > 2000
    # ties are re-synchronised (the reference's choice forced at the tied step, the rest of the stream compared): every
    # decision of every session ends up identical and no call stays unchecked
    assert pc["identical"] == pc["decisions"] and pc["words_identical_sessions"] == pc["sessions"], pc
    assert pc["unchecked_calls"] == 0 and pc["forced_decisions"] == len(pc["tie_divergences"]), pc
    # forced steps count as identical, so `identical == decisions` alone could hide a regression that flips many near-ties:
    # bound them.  Today there is ONE tied decision on these streams (seed 5, call 51, frame scores 1.19e-7 apart), met once per
    # pass over seed 5 (headline warm-up is not checked; the 8-stream leg runs it once): allow a second one, not a drift
    assert pc["forced_decisions"] <= 2, pc["tie_divergences"]
    assert all(abs(t[-1]) < 2e-5 for t in pc["tie_divergences"]), pc["tie_divergences"]
    assert short["parity_checked"]["identical"] - pc["forced_decisions"] >= pc["decisions"] - 2
    assert line["n_gpus"] == 1 and line["eight_streams"]["streams"] == 8
    assert line["eight_streams"]["swallowed_errors"] == 0 and line["swallowed_errors"] == 0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_from_test.json"), "w") as fh:
        json.dump(line, fh)


def test_bench_multi_rank_code_path_rehearsal():
    """The N > 1 path of bench.py end to end on ONE GPU: `--gpus 2` re-executes itself under torch.distributed.run, both
    ranks share device 0 and talk over gloo (WLK_BENCH_REHEARSAL=1; RCCL refuses two ranks per device).  Exercises the
    arena broadcast + adoption, barriers, max-over-ranks timing, stream sharding (i mod N) of the 8-stream leg, the
    object gathers and the merged parity report - what the driver's 2/4/8-GPU runs go through."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WLK_BENCH_REHEARSAL"] = "1"
    tmp = tempfile.mkdtemp(prefix="wlk_bench_")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-large-v3", "--full-out", os.path.join(tmp, "weak.json")], capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    short = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert short["n_gpus"] == 2 and len(short["per_rank_audio_s_per_s"]) == 2 and short["asr_plus_diarization_8_sessions"]["sessions"] == 8
    line = json.load(open(os.path.join(tmp, "weak.json")))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and len(line["per_rank_audio_s_per_s"]) == 2
    assert line["weight_broadcast_ms"] is not None and "rehearsal" in line
    assert len(line["weight_finalize_ms_per_rank"]) == 2 and all(t > 0 for t in line["weight_finalize_ms_per_rank"])
    e = line["eight_streams"]
    assert e["gpus"] == 2 and len(e["per_rank_audio_s_per_s"]) == 2 and e["swallowed_errors"] == 0
    # configs[3] under --gpus N: every rank runs the (ASR + diarizer) pairs of its own streams
    c4 = line["asr_plus_diarization_8_sessions"]
    assert c4["gpus"] == 2 and c4["sessions"] == 8 and c4["per_rank_sessions"] == [4, 4] and c4["swallowed_errors"] == 0, c4
    pc = line["parity_checked"]
    assert pc["sessions"] == 2 + 8 + 8 and pc["mismatches"] == [] and not pc["missing_golden"], pc
    assert pc["identical"] == pc["decisions"] and pc["unchecked_calls"] == 0, pc
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "8", "--steps", "1",
                         "--warmup", "0", "--no-cpu-baseline", "--no-diarization", "--no-eight-streams",
                         "--full-out", os.path.join(tmp, "strong.json")],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert json.loads([l for l in r2.stdout.strip().splitlines() if l.startswith("{")][-1])["scaling"] == "strong"
    strong = json.load(open(os.path.join(tmp, "strong.json")))
    assert strong["scaling"] == "strong" and strong["config"]["streams_total"] == 8 and strong["config"]["streams_this_rank"] == 4
    assert strong["parity_checked"]["sessions"] == 8 and strong["parity_checked"]["mismatches"] == []
    assert strong["parity_checked"]["unchecked_calls"] == 0


def test_bench_refuses_more_gpus_than_visible():
    n = _lib.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_finalized_model_is_immutable():
    """include/wlk_hip.h: a wlk_model is immutable after wlk_model_finalize - sessions size their alignment window from
    it and captured step graphs bake the per-layer head counts in.  Late uploads / head changes must be refused."""
    import ctypes as C
    from whisperlivekit_amd.engine import HipWhisperModel
    m = HipWhisperModel.synthetic("micro.en", 0)
    lib = _lib.load()
    w = np.zeros(128, np.float32)
    rc = lib.wlk_model_upload(m._h, b"enc.conv1.b", w.ctypes.data_as(C.c_void_p), w.size)
    assert rc == -3 and b"finalized" in lib.wlk_last_error()
    pairs = np.asarray([1, 0], np.int32)
    rc = lib.wlk_model_set_alignment_heads(m._h, pairs.ctypes.data_as(C.c_void_p), 1)
    assert rc == -3 and b"finalized" in lib.wlk_last_error()
    s = m.new_session()
    s.append(synth.to_pcm16_roundtrip(synth.speech_like(1.0, 0)))
    assert s.encode() == 50
    s.close()
    m.close()


@pytest.mark.gpu
def test_graft_entry_smoke_runs():
    """The driver's round-end check: one tiny hot-path invocation on cuda:0 compared with the oracle."""
    import importlib
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["tiny.en", "base.en"])
def test_stacked_prefills_equal_prefills_alone(model_name):
    """The engine's stacked prefill chain (wlk_prefill_group through the diag hook) against each session's own prefill:
    five sessions with different audio and prompts of ragged length (one too short for the stack, one past a tile
    boundary) - logits of the last and the sot row, the alignment read-out, and the logits of a following single-token
    step (which reads the self-attention caches the stack filled) are bit-identical."""
    import ctypes as C
    from whisperlivekit_amd import _lib, synth
    from whisperlivekit_amd.engine import HipWhisperModel
    model = HipWhisperModel.synthetic(model_name, 0)
    lib = model.lib
    rng = np.random.default_rng(11)
    lengths = [9, 33, 64, 5, 150]
    prompts = [np.concatenate([[50257, 50362], rng.integers(300, 40000, size=n - 2)]).astype(np.int64) for n in lengths]
    sots = [0, 0, 0, 0, 0]
    audios = [synth.speech_like(3.0 + i, seed=70 + i) for i in range(len(lengths))]

    def fresh():
        out = []
        for a in audios:
            s = model.new_session(beam=1, batched=False)
            s.append(a)
            s.encode()
            out.append(s)
        return out

    def read_out(s, cml):
        lp, ids, fr = s.select([], [], [], 2, cml)
        return s.export("logits_last", model.dims.n_vocab).copy(), s.export("logits_sot", model.dims.n_vocab).copy(), lp.copy(), ids.copy(), fr.copy()

    alone, stacked = fresh(), fresh()
    try:
        want = []
        for s, p in zip(alone, prompts):
            s.decode(p[None], first=True, sot_index=0)
            first = read_out(s, 100)
            s.decode(np.asarray([[1234]]), first=False)
            s.sync()
            want.append((first, s.export("logits_last", model.dims.n_vocab).copy()))
        handles = (C.c_void_p * len(stacked))(*[s._h for s in stacked])
        flat = np.ascontiguousarray(np.concatenate(prompts))
        n_tok = np.asarray(lengths, np.int32)
        sot = np.asarray(sots, np.int32)
        taken = np.zeros(len(lengths), np.int32)
        _lib.check(lib.wlk_diag_prefill_stack(handles, flat.ctypes.data_as(C.c_void_p), n_tok.ctypes.data_as(C.c_void_p),
                                              sot.ctypes.data_as(C.c_void_p), len(lengths), taken.ctypes.data_as(C.c_void_p)))
        assert taken.tolist() == [1, 1, 1, 0, 1]          # 5 tokens: the weight-streaming path of a session alone
        stacked[3].decode(prompts[3][None], first=True, sot_index=0)
        for i, s in enumerate(stacked):
            got = read_out(s, 100)
            for a, b, what in zip(got, want[i][0], ("logits_last", "logits_sot", "top log-probs", "top ids", "frames")):
                np.testing.assert_array_equal(a, b, err_msg=f"session {i} ({lengths[i]} tokens): {what}")
            s.decode(np.asarray([[1234]]), first=False)
            s.sync()
            np.testing.assert_array_equal(s.export("logits_last", model.dims.n_vocab), want[i][1],
                                          err_msg=f"session {i}: step after the stacked prefill")
    finally:
        for s in alone + stacked:
            s.close()
        model.close()
