#!/usr/bin/env python
"""Benchmark of the simul_whisper hot path on MI355X (BASELINE.json: real-time factor + p50
committed-token latency, Whisper-base EN, 0.5 s chunks).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one 30 s synthetic 16 kHz stream pushed through the per-session online processor in 60
chunks of 0.5 s (insert_audio_chunk + process_iter per chunk, compute-unaware = back to back), i.e.
60 passes of  log-mel -> encoder -> cross-K/V -> prefill -> AlignAtt decode loop.  Each rank runs
its own stream(s) against its own weight replica (weights reach ranks > 0 by ONE RCCL broadcast);
there is no steady-state collective.  Rank 0 prints one JSON line.

value = audio seconds transcribed per wall second over the whole job (= 1 / RTF for one stream);
`rtf` and the p50/p95 committed-token latency (compute-aware replay of the measured call times) are
extra keys of the same line, as are `roofline` (dominant kernel, HIP-event timed, algorithmic
FLOPs) and `cpu_baseline` (the torch-CPU oracle = port of the reference, on a bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 8000            # 0.5 s
STREAM_SECONDS = 30.0
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA peak
PEAK_HBM_GBPS = 8000.0

# launch tag -> kernel function (what rocprofv3 --stats reports)
KERNEL_OF_TAG = {
    "enc_conv1": "gemm_nt_f32_kernel", "enc_conv2": "gemm_nt_f32_kernel", "enc_qkv": "gemm_nt_f32_kernel",
    "enc_out": "gemm_nt_f32_kernel", "enc_fc1": "gemm_nt_f32_kernel", "enc_fc2": "gemm_nt_f32_kernel",
    "dec_cross_kv": "gemm_nt_f32_kernel", "enc_attention": "flash_attention_kernel",
    "dec_cross_attention": "decoder_cross_attention_kernel",
    "dec_cross_attention_prefill": "flash_attention_kernel",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="base.en")
    ap.add_argument("--streams-per-gpu", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=STREAM_SECONDS)
    ap.add_argument("--cpu-chunks", type=int, default=24, help="max chunks of the stream the CPU baseline replays")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-work budget of the baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads for the baseline (0 = torch default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--audio", default="speech", choices=["speech", "noise"])
    ap.add_argument("--no-diarization", action="store_true",
                    help="skip the separately timed side legs ('diarization': streaming Sortformer; 'vad': Silero gate)")
    return ap.parse_args()


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def make_audio(kind, seconds, seed):
    from whisperlivekit_amd import synth
    gen = synth.speech_like if kind == "speech" else synth.white_noise
    return synth.to_pcm16_roundtrip(gen(seconds, seed))


def run_stream(proc, audio):
    """-> per-call records [(call wall s, insert wall s, [token end times])]."""
    calls = []
    t_end = 0.0
    for lo in range(0, len(audio), CHUNK):
        chunk = audio[lo:lo + CHUNK]
        t_end += len(chunk) / 16000
        t0 = time.perf_counter()
        proc.insert_audio_chunk(chunk, t_end)
        t1 = time.perf_counter()
        tokens, _ = proc.process_iter()
        t2 = time.perf_counter()
        calls.append((t2 - t1, t1 - t0, [float(t.end) for t in tokens], t_end))
    return calls


def committed_latencies(calls):
    """Compute-aware replay: chunk k is available at its stream time; a call starts when both the
    chunk has arrived and the previous call has finished; latency = finish - token.end."""
    lat = []
    free_at = 0.0
    for wall, ins, ends, t_arrive in calls:
        start = max(t_arrive, free_at)
        free_at = start + ins + wall
        lat += [free_at - e for e in ends]
    return lat


def diarization_leg(audio, seconds, device, cpu_check):
    """SURVEY 8 row a12, timed on its own (never part of `value`): the same stream through the streaming
    Sortformer pass in 1.0 s chunks - HIP log-mel, 17 Conformer + 18 Transformer blocks over
    [speaker cache | FIFO | chunk], host speaker-cache update.  Seeded random weights of the 4-speaker v2 geometry."""
    from whisperlivekit_amd.diarization import HipSortformerDiarizationOnline
    from whisperlivekit_amd.sortformer import HipSortformerModel, SortformerDims, synth_sortformer_state_dict
    dims = SortformerDims()
    sd = synth_sortformer_state_dict(dims, 0)
    model = HipSortformerModel(dims, sd, device=device)
    last = {}
    inner = model.step

    def step(feats, ctx):
        a = time.perf_counter()
        out = inner(feats, ctx)
        last.update(device_ms=1e3 * (time.perf_counter() - a), feats=feats, ctx=ctx, preds=out[1], chunk=out[0])
        return out

    model.step = step

    def run():
        online = HipSortformerDiarizationOnline(model)
        per_chunk, dev = [], []
        for lo in range(0, len(audio) - 15999, 16000):
            online.insert_audio_chunk(audio[lo:lo + 16000])
            a = time.perf_counter()
            online.diarize_sync()
            per_chunk.append(1e3 * (time.perf_counter() - a))
            dev.append(last["device_ms"])
        return online, per_chunk, dev

    run()                                   # warm-up: code paths, allocator, cold caches
    t0 = time.perf_counter()
    online, per_chunk, dev = run()
    wall = time.perf_counter() - t0
    n = len(per_chunk)
    T = online.streaming_state.spkcache_len + online.streaming_state.fifo_len
    out = dict(audio_s_per_s=round(n / wall, 2), rtf=round(wall / n, 6), chunks=n, chunk_s=1.0,
               p50_chunk_ms=round(statistics.median(per_chunk), 3), max_chunk_ms=round(max(per_chunk), 3),
               p50_device_call_ms=round(statistics.median(dev), 3), last_chunk_ms=round(per_chunk[-1], 3),
               context_frames_at_end=int(T), weights="seeded random, diar_streaming_sortformer_4spk-v2 geometry",
               parity="unpinned (NeMo absent); HIP == torch oracle checked in tests/test_gpu_sortformer.py")
    if cpu_check:
        import torch
        from oracle import sortformer_oracle as so
        tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
        od = so.SortformerDims()
        with torch.no_grad():
            a = time.perf_counter()
            embs = torch.cat(([torch.from_numpy(last["ctx"])] if last["ctx"] is not None else []) +
                             [so.pre_encode(tsd, od, torch.from_numpy(last["feats"]))], 0)
            ref = so.forward_embeddings(tsd, od, embs).numpy()
            cpu_ms = 1e3 * (time.perf_counter() - a)
        out.update(cpu_oracle_last_chunk_ms=round(cpu_ms, 1), cpu_threads=torch.get_num_threads(),
                   max_abs_err_vs_oracle_last_chunk=float(np.abs(ref - last["preds"]).max()))
    model.step = inner
    model.close()
    return out


def vad_leg(audio, device, cpu_check):
    """SURVEY 8f rank 3, timed on its own: the Silero VAD gate over the same stream in 0.5 s chunks (real weights:
    the reference's vendored checkpoint, carried as tests/golden/vad_weights_16k.npz)."""
    from whisperlivekit_amd import vad as V
    w = dict(np.load(os.path.join(ROOT, "tests", "golden", "vad_weights_16k.npz")))
    weights = V.HipSileroVADWeights(w, device=device)
    model = V.HipSileroVAD(weights)
    it = V.HipFixedVADIterator(model)
    out = {}
    for rep in range(2):
        it.reset_states()
        ms, n_events = [], 0
        for lo in range(0, len(audio), CHUNK):
            a = time.perf_counter()
            n_events += len(it(audio[lo:lo + CHUNK]))
            ms.append(1e3 * (time.perf_counter() - a))
        out = dict(p50_chunk_ms=round(statistics.median(ms), 4), max_chunk_ms=round(max(ms), 4), chunks=len(ms),
                   chunk_s=0.5, events=n_events, weights="reference checkpoint silero_vad.jit (16 kHz sub-model)",
                   parity="pinned: tests/golden/vad_cases.npz produced by the reference (tests/test_gpu_vad.py)")
    if cpu_check:
        import torch
        from oracle import vad_oracle as vo
        n = (len(audio) // 512) * 512
        model.reset_states()
        probs = model.probs(audio[:n])
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        om = vo.OracleSileroVAD(w)
        a = time.perf_counter()
        ref = np.array([float(om(audio[i:i + 512])) for i in range(0, 64 * 512, 512)], np.float32)
        cpu_ms = 1e3 * (time.perf_counter() - a) / 64
        torch.set_num_threads(threads)
        out.update(cpu_oracle_ms_per_chunk=round(cpu_ms * CHUNK / 512, 3), cpu_threads=1,
                   max_abs_prob_err_vs_oracle=float(np.abs(probs[:64] - ref).max()))
    model.close()
    weights.close()
    return out


def main():
    args = parse_args()
    from whisperlivekit_amd import _lib, sharding, synth
    from whisperlivekit_amd.backend import HipSimulStreamingASR, HipSimulStreamingOnlineProcessor
    from whisperlivekit_amd.dims import ALIGNMENT_HEADS, MODEL_DIMS
    from whisperlivekit_amd.engine import HipWhisperModel, pack_state_dict

    rank, world, local = sharding.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    import torch
    dims = MODEL_DIMS[args.model]
    heads = ALIGNMENT_HEADS[args.model]
    if world > 1:
        import torch.distributed as dist
        sharding.init_process_group("nccl")
        packed = pack_state_dict(dims, synth.synth_state_dict(dims, 0)) if rank == 0 else None
        model = sharding.replicated_model(dims, packed, heads, device=local)
    else:
        dist = None
        model = HipWhisperModel.from_state_dict(dims, synth.synth_state_dict(dims, 0), heads, device=local)
    asr = HipSimulStreamingASR(args.model, hip_model=model)

    n_local = args.streams_per_gpu
    audios = [make_audio(args.audio, args.seconds, rank * n_local + i) for i in range(n_local)]
    total_steps = args.warmup + args.steps
    procs = [[HipSimulStreamingOnlineProcessor(asr) for _ in range(n_local)] for _ in range(total_steps)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(step_procs):
        if n_local == 1:
            return [run_stream(step_procs[0], audios[0])]
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(n_local) as ex:      # sessions are independent: one host thread each
            return list(ex.map(lambda pa: run_stream(*pa), zip(step_procs, audios)))

    log(f"model + {total_steps * n_local} sessions ready")
    for w in range(args.warmup):
        one_step(procs[w])
        log(f"warmup step {w} done")
    barrier()
    t0 = time.perf_counter()
    records = []
    for k in range(args.steps):
        records.append(one_step(procs[args.warmup + k]))
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"{args.steps} timed step(s): {elapsed:.3f} s")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- derived metrics on rank 0's own streams -------------------------------------------------
    all_calls = [c for step in records for stream in step for c in stream]
    asr_time = sum(c[0] for c in all_calls)
    audio_s = args.seconds * n_local * args.steps
    lat = [l for step in records for stream in step for l in committed_latencies(stream)]
    errors = sum(1 for step in procs for p in step if getattr(p, "last_error", None) is not None)

    # ---- roofline: one extra profiled replay of the same step (HIP events on the session stream) ---
    prof_proc = HipSimulStreamingOnlineProcessor(asr)
    prof_proc.model.session.prof_begin()
    run_stream(prof_proc, audios[0])
    prof = prof_proc.model.session.prof_end(cap=64)
    log("profiled replay done")
    by_kernel = {}
    for tag, r in prof.items():
        kname = KERNEL_OF_TAG.get(tag, tag)
        k = by_kernel.setdefault(kname, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
        for f in ("ms", "launches", "flops", "bytes"):
            k[f] += r[f]
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
    gpu_ms = sum(k["ms"] for k in by_kernel.values())
    if dom["flops"] > 0 and dom_name in ("gemm_nt_f32_kernel", "flash_attention_kernel"):
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel=dom_name, achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=None)
    else:
        achieved = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=dom_name, achieved=round(achieved, 2), peak=PEAK_HBM_GBPS, unit="GB/s",
                    frac=round(achieved / PEAK_HBM_GBPS, 4), traffic=None)
    # HBM traffic of that kernel from the committed PMC passes over this same command (bench.py cannot collect
    # counters itself): profiles/r01_pmc_bench.json, produced by scripts/gpu_job_pmc.sh + scripts/export_pmc.py
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_bench.json")))
        hit = next((v for k, v in pmc.items() if k.startswith(dom_name)), None)
        if hit and "hbm_read_bytes_per_launch" in hit:
            roof["traffic"] = round(hit["hbm_read_bytes_per_launch"] + hit.get("hbm_write_bytes_per_launch", 0.0))
            roof["traffic_unit"] = "bytes per launch (HBM read + write)"
            roof["traffic_source"] = ("profiles/r01_pmc_bench.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                      "`bench.py --steps 1 --warmup 1`, FETCH_SIZE x2 gfx950 correction")
            if "mfma_util" in hit:
                roof["mfma_util_pmc"] = round(hit["mfma_util"], 4)
    except (OSError, ValueError):
        pass
    roof["algorithmic_per_launch"] = round((dom["flops"] if roof["bound"] == "mfma" else dom["bytes"]) / max(dom["launches"], 1))
    roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / max(dom["launches"], 1))
    roof.update(avg_launch_us=round(1e3 * dom["ms"] / max(dom["launches"], 1), 2), launches=dom["launches"],
                share_of_gpu_time=round(dom["ms"] / gpu_ms, 3),
                timing="HIP events around every launch, separate profiled replay of one step")
    tags = {k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                    tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] and v["ms"] else None)
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"],
                       tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3) if v["flops"] and v["ms"] else None,
                       gbps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] and v["ms"] else None)
               for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- CPU baseline: the oracle (torch-CPU port of the reference path) on a bounded sample --------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import whisper_oracle as wo
        import helpers
        # Thread count: torch's default on a many-core host (one thread per vCPU) is far from the best for
        # these small fp32 ops, so give the CPU its best shot: time one encoder pass at a few thread
        # counts and keep the fastest (or take --cpu-threads as given).
        if args.cpu_threads > 0:
            torch.set_num_threads(args.cpu_threads)
        else:
            from oracle import whisper_oracle as _wo
            sd_t = _wo.to_torch_state_dict(synth.synth_state_dict(dims, 0))
            mel0, _ = _wo.encoder_input_from_audio(torch.from_numpy(audios[0][:CHUNK * 4].copy()),
                                                   torch.from_numpy(np.array(helpers.mel_filterbank(dims.n_mels))))
            best = None
            ncpu = os.cpu_count() or 1
            for nthr in sorted({min(ncpu, n) for n in (8, 16, 32, 64, ncpu)}):
                torch.set_num_threads(nthr)
                with torch.no_grad():
                    _wo.encoder_forward(sd_t, dims, mel0)
                    a = time.perf_counter()
                    _wo.encoder_forward(sd_t, dims, mel0)
                    dt = time.perf_counter() - a
                log(f"cpu baseline calibration: {nthr} threads -> encoder {dt * 1e3:.0f} ms")
                if best is None or dt < best[1]:
                    best = (nthr, dt)
            torch.set_num_threads(best[0])
        sess = wo.OracleAlignAtt(wo.to_torch_state_dict(synth.synth_state_dict(dims, 0)), dims, heads,
                                 prof_proc.model.tokenizer, np.array(helpers.mel_filterbank(dims.n_mels)),
                                 wo.OracleConfig())
        oproc = wo.OracleOnlineProcessor(sess)
        n_max = min(args.cpu_chunks, len(audios[0]) // CHUNK)
        t_cpu, n = 0.0, 0
        while n < n_max and t_cpu < args.cpu_seconds:
            oproc.insert_audio_chunk(audios[0][n * CHUNK:(n + 1) * CHUNK], (n + 1) * 0.5)
            a = time.perf_counter()
            oproc.process_iter()
            t_cpu += time.perf_counter() - a
            n += 1
            log(f"cpu baseline chunk {n}: {t_cpu:.1f} s so far")
        cpu = dict(value=round(n * 0.5 / t_cpu, 4), unit="audio_s/s", cores=torch.get_num_threads(), kind="port",
                   rtf=round(t_cpu / (n * 0.5), 4),
                   sample=f"first {n} chunks ({n * 0.5:.1f} s) of the same {args.model} stream, "
                          f"torch {torch.__version__} CPU fp32 oracle, {t_cpu:.1f} s of CPU work")

    diar = None
    if rank == 0 and not args.no_diarization:
        try:
            diar = diarization_leg(audios[0], args.seconds, local, cpu_check=not args.no_cpu_baseline)
            log(f"diarization leg done: {diar['p50_chunk_ms']} ms per 1 s chunk")
        except Exception as e:          # never let the side leg take the headline line down
            diar = {"error": f"{type(e).__name__}: {e}"}

    vad = None
    if rank == 0 and not args.no_diarization:
        try:
            vad = vad_leg(audios[0], local, cpu_check=not args.no_cpu_baseline)
        except Exception as e:
            vad = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        timed = [p for step in procs[args.warmup:] for p in step]
        n_enc = sum(p.model.counters["encode"] for p in timed)
        n_dec = sum(p.model.counters["decode"] for p in timed)
        n_pre = sum(p.model.counters["prefill_tokens"] for p in timed)
        out = {
            "metric": "audio seconds transcribed per second (1/RTF), Whisper-base EN simul_whisper AlignAtt, "
                      "0.5 s chunks; rtf and p50 committed-token latency alongside",
            "value": round(args.seconds * n_local * world * args.steps / elapsed, 3),
            "unit": "audio_s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} simul_whisper, {n_local} stream(s)/GPU, {args.seconds:g} s "
                                   f"16 kHz {args.audio}-like stream, 0.5 s chunks, beam 1, frame_threshold 25, "
                                   "seeded random weights, synthetic vocabulary",
                       "model": args.model, "streams_per_gpu": n_local, "chunk_s": 0.5},
            "rtf": round(asr_time / audio_s, 5),
            "rtf_wall_inclusive": round(elapsed * 1.0 / (args.seconds * args.steps), 5),
            "p50_committed_token_latency_ms": round(1e3 * statistics.median(lat), 2) if lat else None,
            "p95_committed_token_latency_ms": round(1e3 * float(np.percentile(lat, 95)), 2) if lat else None,
            "committed_tokens": len(lat),
            "calls": len(all_calls),
            "decode_steps_per_call": round(n_dec / max(n_enc, 1), 2),
            "prefill_tokens_per_call": round(n_pre / max(n_enc, 1), 2),
            "p50_call_ms": round(1e3 * statistics.median(c[0] for c in all_calls), 3),
            "p50_insert_ms": round(1e3 * statistics.median(c[1] for c in all_calls), 3),
            "swallowed_errors": errors,
            "roofline": roof,
            "gpu_kernel_ms_per_step": round(gpu_ms, 2),
            "kernels": kernels,
            "launch_tags": tags,
            "cpu_baseline": cpu,
            "diarization": diar,
            "vad": vad,
            "host_cores": os.cpu_count(),
        }
        print(json.dumps(out))
    for step in procs:
        for p in step:
            p.close()
    prof_proc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
