"""ctypes binding of libwlk_hip.so (include/wlk_hip.h).  Fails loudly: there is no CPU fallback -
if the HIP library is missing or cannot be loaded every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

# Importing this module changes nothing in the process environment (the host process may run other HIP users - torch
# diarization, translation models - whose runtime settings are not this backend's to retune).

_LIB_NAME = "libwlk_hip.so"
_lock = threading.Lock()
_lib: Optional[C.CDLL] = None


class WlkError(RuntimeError):
    """Raised for every non-zero status of the C ABI (message from wlk_last_error())."""


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class SfDims(C.Structure):
    """wlk_sf_dims (include/wlk_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "sub_channels", "fc_d_model", "fc_layers", "fc_heads", "fc_ff", "conv_kernel",
        "tf_d_model", "tf_layers", "tf_heads", "tf_inner", "n_spk", "max_frames", "max_feat_frames")] + [
        ("xscale", C.c_float)]


class NllbDims(C.Structure):
    """wlk_nllb_dims (include/wlk_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "vocab", "d_model", "heads", "ffn", "enc_layers", "dec_layers", "max_src", "max_tgt", "pad_id", "n_positions")] + [
        ("embed_scale", C.c_float)]


class LoopParams(C.Structure):
    """wlk_loop_params (include/wlk_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "sot_index", "is_last", "frame_threshold", "rewind_threshold", "last_attend_frame", "max_text_len", "budget",
        "eot", "dec_pad", "no_speech_token")] + [("no_speech_threshold", C.c_float), ("content_mel_len", C.c_int32),
                                                   ("n_force", C.c_int32), ("force_step", C.c_int32 * 4),
                                                   ("force_token", C.c_int32 * 4), ("force_frame", C.c_int32 * 4)]
    MAX_FORCED = 4

    def force(self, entries) -> None:
        """entries: [(step, token or -1, frame or -1)] - teacher forcing of tied decisions (parity harnesses only)."""
        entries = list(entries)
        if len(entries) > self.MAX_FORCED:
            raise ValueError(f"at most {self.MAX_FORCED} forced decisions per loop")
        self.n_force = len(entries)
        for i, (step, token, frame) in enumerate(entries):
            self.force_step[i], self.force_token[i], self.force_frame[i] = int(step), int(token), int(frame)


class LoopResult(C.Structure):
    """wlk_loop_result (include/wlk_hip.h)"""
    _fields_ = [("n_steps", C.c_int32), ("n_new_tokens", C.c_int32), ("stop_reason", C.c_int32),
                ("last_attend_frame", C.c_int32), ("no_speech_prob", C.c_float), ("sum_logprob", C.c_float),
                ("decode_calls", C.c_int32)]


STOP_NONE, STOP_CONTEXT_FULL, STOP_BUDGET, STOP_NO_SPEECH, STOP_COMPLETED, STOP_REWIND, STOP_FRAME = range(7)


def configure_hw_queues(n: int) -> bool:
    """Deployment knob, never applied implicitly: map the process's HIP streams onto ``n`` hardware queues
    (GPU_MAX_HW_QUEUES; HIP's default is 4).  Serving 8 sessions of one GPU from one process measured 2 queues +2.5-3.6 %
    over 4, 8 queues -4 %, 16 queues -8 % (one session: no difference): GPU-filling encoder kernels of different streams
    slow each other down more than they overlap.  The HIP runtime reads the variable once, when it initialises, and it
    applies to EVERY HIP user of the process - so this must be called by the process owner before anything touches HIP
    (`HipSimulStreamingASR(hw_queues=2)` forwards here; `bench.py` exports the variable itself for its single-process
    runs).  An exported value wins.  Returns True when the value was set by this call."""
    if n < 1:
        raise ValueError("hw_queues must be >= 1")
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    if _lib is not None:
        raise WlkError("configure_hw_queues: the HIP library is already loaded in this process - export "
                       "GPU_MAX_HW_QUEUES before start-up instead")
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


def lib_path() -> str:
    return os.environ.get("WLK_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME))


def _declare(lib: C.CDLL) -> None:
    p, i32, u64, f32p, i32p = C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    cint = C.c_int
    sig = {
        "wlk_last_error": (C.c_char_p, []),
        "wlk_abi_version": (cint, []),
        "wlk_device_count": (cint, []),
        "wlk_arena_floats": (cint, [C.POINTER(Dims), C.POINTER(u64)]),
        "wlk_tensor_lookup": (cint, [C.POINTER(Dims), C.c_char_p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_tensor_name": (cint, [C.POINTER(Dims), cint, C.POINTER(C.c_char_p)]),
        "wlk_model_create": (cint, [C.POINTER(Dims), cint, p, C.POINTER(p)]),
        "wlk_model_arena": (cint, [p, C.POINTER(p), C.POINTER(u64)]),
        "wlk_model_upload": (cint, [p, C.c_char_p, p, u64]),
        "wlk_model_set_alignment_heads": (cint, [p, p, cint]),
        "wlk_model_finalize": (cint, [p]),
        "wlk_model_destroy": (cint, [p]),
        "wlk_session_create": (cint, [p, cint, cint, C.POINTER(p)]),
        "wlk_session_destroy": (cint, [p]),
        "wlk_session_set_debug": (cint, [p, cint]),
        "wlk_audio_append": (cint, [p, p, cint]),
        "wlk_audio_append_pcm16": (cint, [p, p, cint]),
        "wlk_audio_append_zeros": (cint, [p, cint]),
        "wlk_audio_drop_front": (cint, [p, cint]),
        "wlk_audio_clear": (cint, [p]),
        "wlk_audio_len": (cint, [p, C.POINTER(cint)]),
        "wlk_encode": (cint, [p, C.POINTER(i32)]),
        "wlk_decode": (cint, [p, p, cint, cint, cint, cint]),
        "wlk_no_speech_prob": (cint, [p, cint, p]),
        "wlk_rules_set": (cint, [p, p, cint, p, cint]),
        "wlk_pick_greedy": (cint, [p, p, p, p]),
        "wlk_select": (cint, [p, p, p, p, cint, cint, cint, p, p, p]),
        "wlk_kv_reorder": (cint, [p, p, cint]),
        "wlk_sync": (cint, [p]),
        "wlk_decode_until_stop": (cint, [p, p, cint, C.POINTER(LoopParams), p, cint, p, cint, C.POINTER(LoopResult),
                                         p, p, p, p, cint]),
        "wlk_engine_attach": (cint, [p]),
        "wlk_engine_detach": (cint, [p]),
        "wlk_engine_encode_stats": (cint, [p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_engine_prefill_stats": (cint, [p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_diag_prefill_stack": (cint, [p, p, p, p, i32, p]),
        "wlk_engine_stats": (cint, [p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "wlk_job_create": (cint, [C.POINTER(LoopParams), p, cint, p, cint, p, cint, C.POINTER(p)]),
        "wlk_job_begin_step": (cint, [p, C.POINTER(i32)]),
        "wlk_job_no_speech": (cint, [p, C.c_float, C.POINTER(i32)]),
        "wlk_job_adjustments": (cint, [p, C.POINTER(i32p), C.POINTER(f32p), C.POINTER(i32)]),
        "wlk_job_consume": (cint, [p, p, p, cint, C.POINTER(i32)]),
        "wlk_job_result": (cint, [p, C.POINTER(LoopResult), p, p, p, p, cint]),
        "wlk_job_destroy": (cint, [p]),
        "wlk_export": (cint, [p, C.c_char_p, p, u64, C.POINTER(u64)]),
        "wlk_session_step_stats": (cint, [p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "wlk_prof_begin": (cint, [p]),
        "wlk_prof_end": (cint, [p, cint, C.POINTER(C.c_char_p), p, p, p, p, C.POINTER(i32)]),
        "wlk_melspec_create": (cint, [cint, cint, cint, cint, cint, p, p, C.c_float, C.c_float, cint, C.POINTER(p)]),
        "wlk_melspec_run": (cint, [p, p, cint, p, cint, C.POINTER(cint)]),
        "wlk_melspec_destroy": (cint, [p]),
        "wlk_sf_arena_floats": (cint, [C.POINTER(SfDims), C.POINTER(u64)]),
        "wlk_sf_tensor_lookup": (cint, [C.POINTER(SfDims), C.c_char_p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_sf_tensor_name": (cint, [C.POINTER(SfDims), cint, C.POINTER(C.c_char_p)]),
        "wlk_sf_create": (cint, [C.POINTER(SfDims), cint, C.POINTER(p)]),
        "wlk_sf_upload": (cint, [p, C.c_char_p, p, u64]),
        "wlk_sf_finalize": (cint, [p]),
        "wlk_sf_step": (cint, [p, p, cint, p, cint, p, cint, C.POINTER(cint), p, cint]),
        "wlk_sf_step_pcm": (cint, [p, p, p, cint, cint, p, cint, p, cint, C.POINTER(cint), p, cint, p, cint, C.POINTER(cint), p, cint]),
        "wlk_sf_stats": (cint, [p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "wlk_sf_export": (cint, [p, C.c_char_p, p, u64, C.POINTER(u64)]),
        "wlk_sf_destroy": (cint, [p]),
        "wlk_nllb_arena_floats": (cint, [C.POINTER(NllbDims), C.POINTER(u64)]),
        "wlk_nllb_tensor_lookup": (cint, [C.POINTER(NllbDims), C.c_char_p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_nllb_tensor_name": (cint, [C.POINTER(NllbDims), cint, C.POINTER(C.c_char_p)]),
        "wlk_nllb_create": (cint, [C.POINTER(NllbDims), cint, C.POINTER(p)]),
        "wlk_nllb_upload": (cint, [p, C.c_char_p, p, u64]),
        "wlk_nllb_finalize": (cint, [p]),
        "wlk_nllb_destroy": (cint, [p]),
        "wlk_nllb_session_create": (cint, [p, cint, C.POINTER(p)]),
        "wlk_nllb_session_destroy": (cint, [p]),
        "wlk_nllb_encode": (cint, [p, p, i32]),
        "wlk_nllb_decode": (cint, [p, p, i32, i32, i32]),
        "wlk_nllb_step": (cint, [p, p, i32, i32, p, p]),
        "wlk_nllb_kv_reorder": (cint, [p, p, i32]),
        "wlk_nllb_topk": (cint, [p, i32, p, p]),
        "wlk_nllb_export": (cint, [p, C.c_char_p, p, u64, C.POINTER(u64)]),
        "wlk_nllb_sync": (cint, [p]),
        "wlk_vad_weights_floats": (cint, [C.POINTER(u64)]),
        "wlk_vad_tensor_lookup": (cint, [C.c_char_p, C.POINTER(u64), C.POINTER(u64)]),
        "wlk_vad_tensor_name": (cint, [cint, C.POINTER(C.c_char_p)]),
        "wlk_vad_create": (cint, [cint, p, u64, C.POINTER(p)]),
        "wlk_vad_destroy": (cint, [p]),
        "wlk_vad_stream_create": (cint, [p, cint, C.POINTER(p)]),
        "wlk_vad_stream_reset": (cint, [p]),
        "wlk_vad_stream_run": (cint, [p, p, cint, p]),
        "wlk_vad_stream_state": (cint, [p, p, p]),
        "wlk_vad_stream_destroy": (cint, [p]),
        "wlk_diag_last_error": (C.c_char_p, []),
        "wlk_diag_linear": (cint, [p, C.c_int64, C.c_int64, p, p, p, C.c_int64, cint, cint, cint, cint, C.c_float,
                                   cint, cint, p]),
        "wlk_dtw": (cint, [cint, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_int8)]),
        "wlk_encode_mel": (cint, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
        "wlk_log_mel": (cint, [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.POINTER(C.c_float), C.c_uint64,
                               C.POINTER(C.c_int32)]),
        "wlk_find_alignment": (cint, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                      C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_float)]),
        "wlk_diag_linear_time": (cint, [cint, cint, cint, cint, cint, cint, C.POINTER(C.c_float)]),
        "wlk_diag_linear_ln": (cint, [p, p, p, p, p, cint, cint, cint, cint, p]),
        "wlk_diag_layernorm": (cint, [p, p, p, cint, cint, p]),
        "wlk_diag_encoder_attention_time": (cint, [cint, cint, cint, cint, cint, C.POINTER(C.c_float)]),
        "wlk_diag_encoder_attention": (cint, [p, cint, cint, cint, p]),
        "wlk_diag_wave_ops": (cint, [p, p, p]),
        "wlk_diag_env_refresh": (cint, []),
        "wlk_diag_linear_x3": (cint, [p, p, p, cint, cint, cint, cint, C.c_float, cint, p]),
        "wlk_diag_linear_x3_time": (cint, [cint, cint, cint, cint, cint, C.POINTER(C.c_float)]),
        "wlk_diag_layernorm_x3": (cint, [p, p, p, cint, cint, p]),
        "wlk_diag_encoder_attention_x3": (cint, [p, cint, cint, cint, p]),
        "wlk_diag_encoder_attention_x3_time": (cint, [cint, cint, cint, cint, C.POINTER(C.c_float)]),
        "wlk_diag_qkv_x3_attention": (cint, [p, p, p, cint, cint, cint, C.c_float, p, p]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)      # AttributeError here = the library lacks a declared symbol
        except AttributeError:
            # an explicitly named library (WLK_HIP_LIB: A/B runs against an older build kept beside the tree) may predate a
            # diagnostic entry point; the in-tree library must export everything
            if "WLK_HIP_LIB" in os.environ and name.startswith("wlk_diag_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args


EXPORTED_SYMBOLS = (
    "wlk_last_error", "wlk_abi_version", "wlk_device_count", "wlk_arena_floats", "wlk_tensor_lookup",
    "wlk_tensor_name", "wlk_model_create", "wlk_model_arena", "wlk_model_upload",
    "wlk_model_set_alignment_heads", "wlk_model_finalize", "wlk_model_destroy", "wlk_session_create",
    "wlk_session_destroy", "wlk_session_set_debug", "wlk_audio_append", "wlk_audio_append_pcm16", "wlk_audio_append_zeros",
    "wlk_audio_drop_front", "wlk_audio_clear", "wlk_audio_len", "wlk_encode", "wlk_decode",
    "wlk_no_speech_prob", "wlk_rules_set", "wlk_pick_greedy", "wlk_select", "wlk_kv_reorder", "wlk_sync", "wlk_decode_until_stop", "wlk_engine_attach", "wlk_engine_detach",
    "wlk_engine_stats", "wlk_engine_encode_stats", "wlk_engine_prefill_stats", "wlk_diag_prefill_stack", "wlk_job_create",
    "wlk_job_begin_step", "wlk_job_no_speech", "wlk_job_adjustments", "wlk_job_consume", "wlk_job_result",
    "wlk_job_destroy", "wlk_export", "wlk_session_step_stats", "wlk_prof_begin",
    "wlk_prof_end", "wlk_melspec_create", "wlk_melspec_run", "wlk_melspec_destroy",
    "wlk_sf_arena_floats", "wlk_sf_tensor_lookup", "wlk_sf_tensor_name", "wlk_sf_create", "wlk_sf_upload",
    "wlk_sf_finalize", "wlk_sf_step", "wlk_sf_step_pcm", "wlk_sf_stats", "wlk_sf_export", "wlk_sf_destroy",
    "wlk_vad_weights_floats", "wlk_vad_tensor_lookup", "wlk_vad_tensor_name", "wlk_vad_create", "wlk_vad_destroy",
    "wlk_vad_stream_create", "wlk_vad_stream_reset", "wlk_vad_stream_run", "wlk_vad_stream_state",
    "wlk_vad_stream_destroy",
    "wlk_dtw", "wlk_encode_mel", "wlk_log_mel", "wlk_find_alignment",
    "wlk_nllb_arena_floats", "wlk_nllb_tensor_lookup", "wlk_nllb_tensor_name", "wlk_nllb_create", "wlk_nllb_upload",
    "wlk_nllb_finalize", "wlk_nllb_destroy", "wlk_nllb_session_create", "wlk_nllb_session_destroy", "wlk_nllb_encode",
    "wlk_nllb_decode", "wlk_nllb_step", "wlk_nllb_kv_reorder", "wlk_nllb_topk", "wlk_nllb_export", "wlk_nllb_sync",
    "wlk_diag_last_error", "wlk_diag_linear", "wlk_diag_linear_time", "wlk_diag_linear_ln", "wlk_diag_layernorm",
    "wlk_diag_encoder_attention", "wlk_diag_encoder_attention_time", "wlk_diag_wave_ops", "wlk_diag_env_refresh",
    "wlk_diag_linear_x3", "wlk_diag_linear_x3_time", "wlk_diag_layernorm_x3",
    "wlk_diag_encoder_attention_x3", "wlk_diag_encoder_attention_x3_time", "wlk_diag_qkv_x3_attention",
)


def load() -> C.CDLL:
    """Load (once) and return the library.  Raises WlkError if it is not built / not loadable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            path = lib_path()
            if not os.path.exists(path):
                raise WlkError(
                    f"{path} not found: build it with `python -m whisperlivekit_amd.build` "
                    "(hipcc, gfx950). There is no CPU fallback for the HIP backend.")
            try:
                # torch ships its own libamdhip64.so.7; importing it first makes torch and this
                # library share ONE HIP runtime (same soname) when both are in the process.
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch is optional for the C ABI itself
                pass
            try:
                lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError as e:
                raise WlkError(f"cannot load {path}: {e}") from e
            _declare(lib)
            _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().wlk_last_error()
        raise WlkError(f"wlk_hip error {rc}: {msg.decode('utf-8', 'replace') if msg else '?'}")


def device_count() -> int:
    return int(load().wlk_device_count())
