/*
 * wlk_hip.h - C ABI of libwlk_hip.so, the MI355X (gfx950) backend for WhisperLiveKit's
 * simul_whisper hot path.
 *
 * The reference has no FFI for this path: its plugin boundary is the set of abstract tensor
 * hooks of AlignAttBase (whisperlivekit/simul_whisper/align_att_base.py:541-649), implemented
 * in PyTorch by AlignAtt (whisperlivekit/simul_whisper/simul_whisper.py:108-462).  Each entry
 * point below names the hook(s)/reference code it replaces.  Everything is plain pointers and
 * sizes; `float*` arguments marked "host" are host memory, "dev" are device (HBM) pointers.
 *
 * Conventions
 *  - every function returns 0 (WLK_OK) or a negative error code; wlk_last_error() returns the
 *    message of the last failure on the calling thread (the Python shim raises RuntimeError,
 *    matching the reference's exception convention, simul_whisper/backend.py:266-268).
 *  - a wlk_model is immutable after wlk_model_finalize() and may be shared by any number of
 *    sessions / threads; a wlk_session is used by one thread at a time (the reference's
 *    threading contract).
 *  - all arithmetic is fp32 (whisper/model.py:39-50 keeps fp32 end to end on this path).
 */
#ifndef WLK_HIP_H
#define WLK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WLK_OK 0
#define WLK_ERR_ARG (-1)
#define WLK_ERR_HIP (-2)
#define WLK_ERR_STATE (-3)
#define WLK_ERR_CAPACITY (-4)

typedef struct wlk_model wlk_model;
typedef struct wlk_session wlk_session;

/* ModelDimensions, whisperlivekit/whisper/model.py:26-37 */
typedef struct wlk_dims {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wlk_dims;

const char* wlk_last_error(void);
int wlk_abi_version(void);
/* number of visible HIP devices (0 if none / runtime unusable); never fails */
int wlk_device_count(void);

/* ---- weights: one packed fp32 arena per GPU ------------------------------------------------
 * Replaces load_model()'s `.to(device)` of an nn.Module (whisper/__init__.py:466-596): tensors
 * keep the reference checkpoint names; conv weights are stored tap-major and q/k/v (k/v for
 * cross attention) are stored concatenated so each projection is one GEMM - the repacking is
 * done by the host shim, the layout is defined by wlk_tensor_lookup(). */
int wlk_arena_floats(const wlk_dims* dims, uint64_t* n_floats);
int wlk_tensor_lookup(const wlk_dims* dims, const char* packed_name, uint64_t* offset_floats,
                      uint64_t* numel);
/* n-th packed tensor name (0 <= index < count); returns WLK_ERR_ARG past the end */
int wlk_tensor_name(const wlk_dims* dims, int index, const char** name);

/* arena_dev may be NULL (the library hipMallocs it) or a caller-owned device buffer of
 * wlk_arena_floats() floats (e.g. a torch tensor that torch.distributed broadcasts over RCCL) */
int wlk_model_create(const wlk_dims* dims, int device, float* arena_dev, wlk_model** out);
int wlk_model_arena(wlk_model* m, float** arena_dev, uint64_t* n_floats);
int wlk_model_upload(wlk_model* m, const char* packed_name, const float* host, uint64_t numel);
/* (decoder layer, head) pairs in alignment-rank order; Whisper.set_alignment_heads,
 * whisper/model.py:363-370 + simul_whisper.py:151-159 */
int wlk_model_set_alignment_heads(wlk_model* m, const int32_t* layer_head_pairs, int n_pairs);
int wlk_model_finalize(wlk_model* m);
int wlk_model_destroy(wlk_model* m);

/* ---- per-stream session --------------------------------------------------------------------
 * Owns what DecoderState owns (simul_whisper/decoder_state.py:7-91) that is tensor-valued:
 * the rolling audio buffer, KV caches, the 16-step cross-attention window, workspaces, and a
 * HIP stream. */
int wlk_session_create(wlk_model* m, int beam, int max_audio_samples, wlk_session** out);
int wlk_session_destroy(wlk_session* s);
/* 1 = keep pre-softmax cross-attention QK of every layer/head and raw logits for wlk_export */
int wlk_session_set_debug(wlk_session* s, int on);

/* a1: AlignAtt.insert_audio (simul_whisper.py:219-237): append a chunk / evict the oldest
 * samples / drop everything.  Only the new chunk crosses PCIe. */
int wlk_audio_append(wlk_session* s, const float* pcm_host, int n);
/* (f-next, rank 3) the same append for int16 PCM as it arrives on the wire: replaces convert_pcm_to_float
 * (whisperlivekit/audio_processor.py:416-418: int16 -> float32 / 32768.0) + the fp32 upload; half the PCIe bytes */
int wlk_audio_append_pcm16(wlk_session* s, const int16_t* pcm_host, int n);
int wlk_audio_append_zeros(wlk_session* s, int n);
int wlk_audio_drop_front(wlk_session* s, int n);
int wlk_audio_clear(wlk_session* s);
int wlk_audio_len(wlk_session* s, int* n);

/* a2+a3+a4: AlignAtt._encode (simul_whisper.py:344-352) = log_mel_spectrogram(padding=30 s)
 * -> first 3000 frames -> AudioEncoder.forward, plus the cross-attention K/V projection of
 * every decoder layer (MultiHeadAttention.forward's first-use branch, whisper/model.py:117-126).
 * content_mel_len = int((T_padded - 3000) / 2). */
int wlk_encode(wlk_session* s, int32_t* content_mel_len);

/* a5: TextDecoder.forward(tokens, xa, kv_cache, return_cross_attn=True) (whisper/model.py:279-332)
 * via AlignAtt._get_logits_and_cross_attn (simul_whisper.py:357-368).  tokens: host int64
 * [n_rows, n_tok] row-major, n_rows = beam.  first != 0 starts a new `infer` (empty self-attn
 * cache and alignment window, i.e. after DecoderState.clean_cache()).  Logits are produced for
 * the last position and, when first, for position sot_index (the only rows the policy reads,
 * align_att_base.py:226-229); softmaxed cross-attention rows of the alignment heads go to the
 * session's 16-step window (align_att_base.py:221-224). Asynchronous on the session stream. */
int wlk_decode(wlk_session* s, const int64_t* tokens, int n_rows, int n_tok, int first, int sot_index);

/* a6: AlignAtt._check_no_speech (simul_whisper.py:370-377): softmax(logits[:, sot_index])[token] */
int wlk_no_speech_prob(wlk_session* s, int no_speech_token, float* probs_host /* [n_rows] */);

/* f4 (batch `transcribe`): the per-step logit rules of whisper's DecodingTask and the greedy pick at temperature 0 for ONE
 * sequence, on the device - SuppressBlank / SuppressTokens / ApplyTimestampRules (whisper/decoding.py:417-499) and
 * GreedyDecoder.update (whisper/decoding.py:270-287).  wlk_rules_set uploads the two token lists of a DecodingTask
 * (decoding.py:609-636 `_get_suppress_tokens`, :424 blank = encode(" ") + [eot]); wlk_pick_greedy applies the rules to the
 * logits of the last wlk_decode and returns the chosen token and its log-probability (8 bytes instead of the logits row).
 * The host derives the history-dependent fields from the sampled tokens exactly as ApplyTimestampRules.apply does:
 *   ts_mode  0: the last sampled token is no timestamp; 1: the last two are timestamps (or only one token was sampled and
 *            it is one) - timestamps are suppressed; 2: only the last one is - text tokens below eot are suppressed
 *   ts_bound timestamps in [timestamp_begin, ts_bound) are suppressed (timestamp_begin = none)
 *   max_initial  first step only: timestamps above timestamp_begin + max_initial are suppressed (-1 = no limit) */
typedef struct wlk_pick_params {
    int32_t first_step, without_timestamps, timestamp_begin, eot, no_timestamps /* -1 = none */, ts_mode, ts_bound, max_initial;
} wlk_pick_params;
int wlk_rules_set(wlk_session* s, const int32_t* suppressed, int n_suppressed, const int32_t* blank, int n_blank);
int wlk_pick_greedy(wlk_session* s, const wlk_pick_params* p, int32_t* token_host, float* logprob_host);

/* a6+a7+a8 in one launch group and ONE readback:
 *  - logits[row, ids[i]] += deltas[i] (-inf suppresses: SuppressTokens.apply whisper/decoding.py:427-432,
 *    _suppress_blank_tokens simul_whisper.py:379-381, DRY penalty align_att_base.py:492-537);
 *  - log_softmax + top-k (the device part of BeamSearchDecoder.update, whisper/decoding.py:332-338);
 *  - AlignAtt._process_cross_attention + _get_attended_frames (simul_whisper.py:390-437) over the
 *    current window: most attended frame of the newest row, per beam row.
 * adj_row[i] < 0 applies the adjustment to every row. */
int wlk_select(wlk_session* s, const int32_t* adj_row, const int32_t* adj_ids, const float* adj_deltas,
               int n_adj, int k, int content_mel_len,
               float* top_logprobs_host /* [n_rows, k] */, int32_t* top_ids_host /* [n_rows, k] */,
               int32_t* frames_host /* [n_rows] */);

/* BeamPyTorchInference.rearrange_kv_cache (simul_whisper/beam.py:15-19) */
int wlk_kv_reorder(wlk_session* s, const int32_t* source_rows, int n_rows);

int wlk_sync(wlk_session* s);

/* ---- (f-next, rank 1) the AlignAtt decode loop of one infer behind ONE call -------------------------------
 * Replaces the per-token Python loop of AlignAttBase.infer (simul_whisper/align_att_base.py:206-286) for beam 1:
 * decoder forward, no-speech check (first step), blank / special-token suppression, DRY penalty
 * (align_att_base.py:492-537), BeamSearchDecoder.update with beam_size 1 (whisper/decoding.py:317-376), AlignAtt
 * read-out and the stop rules (completed / rewind / frame threshold / token budget) - no Python between tokens.
 * The caller keeps everything before and after the loop (context trimming, word splitting, timestamps). */
typedef struct wlk_loop_params {
    int32_t sot_index;            /* index of <|startoftranscript|> in the fed tokens (no-speech row) */
    int32_t is_last;
    int32_t frame_threshold;      /* cfg.frame_threshold (4 is used when is_last) */
    int32_t rewind_threshold;     /* cfg.rewind_threshold */
    int32_t last_attend_frame;    /* state.last_attend_frame on entry */
    int32_t max_text_len;         /* model.dims.n_text_ctx */
    int32_t budget;               /* max(50, int(seconds * 15 * 1.5)): decoder forwards allowed in this infer */
    int32_t eot;                  /* tokenizer.eot */
    int32_t dec_pad;              /* DEC_PAD = 50257 (align_att_base.py:9) */
    int32_t no_speech_token;      /* < 0: no check */
    float no_speech_threshold;    /* cfg.nonspeech_prob */
    int32_t content_mel_len;      /* from wlk_encode, unclipped */
    /* Teacher forcing of single decisions, for parity harnesses only (0 entries in production): when a decision of
     * this library and the reference's differ inside an fp32 tie (two candidates closer than rounding), the harness
     * replays the stream with the reference's choice forced at that step so that every later decision can still be
     * compared.  Entry f applies to decode step force_step[f] of this loop (0 = the prefill's read-out): the attended
     * frame becomes force_frame[f] (if >= 0); the selected token becomes force_token[f] (if >= 0) provided it is the
     * runner-up of the step's top-2 and the winner is not end-of-text (the two are then swapped, log-probs included). */
    int32_t n_force;              /* 0 .. WLK_MAX_FORCED */
    int32_t force_step[4];
    int32_t force_token[4];
    int32_t force_frame[4];
} wlk_loop_params;
#define WLK_MAX_FORCED 4
enum {
    WLK_STOP_NONE = 0,
    WLK_STOP_CONTEXT_FULL = 1,    /* tokens reached max_text_len */
    WLK_STOP_BUDGET = 2,          /* runaway guard: every token of this infer is dropped */
    WLK_STOP_NO_SPEECH = 3,
    WLK_STOP_COMPLETED = 4,       /* end-of-text won: the last appended token is dropped */
    WLK_STOP_REWIND = 5,          /* attention jumped back: every token of this infer is dropped */
    WLK_STOP_FRAME = 6            /* attention reached the end of the audio: the last appended token is dropped */
};
typedef struct wlk_loop_result {
    int32_t n_steps;              /* decode steps that selected a token (= entries of step_tokens / step_frames) */
    int32_t n_new_tokens;         /* tokens kept (after the drops above) */
    int32_t stop_reason;
    int32_t last_attend_frame;    /* state.last_attend_frame on exit */
    float no_speech_prob;
    float sum_logprob;
    int32_t decode_calls;         /* decoder forwards run */
} wlk_loop_result;
/* tokens: host int64 [n_tok] = context + prompt (what _current_tokens() returns, one row).  suppress_ids: the
 * SuppressTokens list; blank_ids: encode(" ") + [eot].  Outputs (each of capacity `cap`, may be NULL except result):
 * the kept new tokens, and per decode step the selected token, the attended frame and the running log-prob sum.
 * The session must have been encoded; its beam must be 1.  Sessions attached to a batch engine (wlk_engine_attach)
 * run their single-token steps batched with the other attached sessions of the model. */
int wlk_decode_until_stop(wlk_session* s, const int64_t* tokens, int n_tok, const wlk_loop_params* p,
                          const int32_t* suppress_ids, int n_suppress, const int32_t* blank_ids, int n_blank,
                          wlk_loop_result* result, int64_t* new_tokens, int32_t* step_tokens, int32_t* step_frames,
                          float* step_sum_logprobs, int cap);
/* Cross-session batching of those loops: the reference serves N sessions from one model on one device
 * (whisperlivekit/core.py:246-271, audio_processor.py:543-551), each issuing its own launch chain per token.  Sessions
 * attached here hand the single-token steps of wlk_decode_until_stop to ONE worker per GPU that advances every loop
 * currently in its decode phase in one launch chain (rows = sessions; weights are streamed once per step for all of
 * them; per-row arithmetic and its order are those of a session running alone).  The prefills (first decoder pass of an
 * infer) that are waiting at the same time run as one stacked chain as well (csrc/api.hip: wlk_prefill_group), bit-identical
 * to a session's own prefill.  Beam-1 sessions only. */
int wlk_engine_attach(wlk_session* s);
int wlk_engine_detach(wlk_session* s);
/* encode launch chains run by the engine and the sessions encoded in them (concurrent wlk_encode calls of attached
 * sessions are stacked: one launch per encoder operator with grid.y = sessions, shared weights) */
int wlk_engine_encode_stats(wlk_model* m, uint64_t* batches, uint64_t* sessions);
/* engine iterations, rows advanced in them, and the batched (>= 2 rows) iterations / rows among those */
int wlk_engine_stats(wlk_model* m, uint64_t* iterations, uint64_t* rows, uint64_t* batched_steps, uint64_t* batched_rows);
/* The host half of that loop without a GPU (integer logic only), for harnesses that supply the numerics themselves:
 * begin_step -> n_feed tokens to run through the decoder (0 = loop over); no_speech (first step only); adjustments ->
 * unique (id, additive delta) pairs to apply to the last-position logits; consume(top-2 log-probs/ids after the
 * adjustments, attended frame) -> goes_on. */
typedef struct wlk_decode_job wlk_decode_job;
int wlk_job_create(const wlk_loop_params* p, const int64_t* tokens, int n_tok, const int32_t* suppress_ids, int n_suppress,
                   const int32_t* blank_ids, int n_blank, wlk_decode_job** out);
int wlk_job_begin_step(wlk_decode_job* j, int32_t* n_feed);
int wlk_job_no_speech(wlk_decode_job* j, float prob, int32_t* stops);
int wlk_job_adjustments(wlk_decode_job* j, const int32_t** ids, const float** deltas, int32_t* n);
int wlk_job_consume(wlk_decode_job* j, const float* top_logprobs, const int32_t* top_ids, int frame, int32_t* goes_on);
int wlk_job_result(wlk_decode_job* j, wlk_loop_result* result, int64_t* new_tokens, int32_t* step_tokens,
                   int32_t* step_frames, float* step_sum_logprobs, int cap);
int wlk_job_destroy(wlk_decode_job* j);

/* Parity/debug exports to host memory.  what: "mel" [n_mels,3000], "enc" [1500,d],
 * "logits_last" / "logits_sot" [n_rows,V], "attn_last" [n_rows, content_mel_len] (after
 * wlk_select), "cross_qk:<layer>" [rows, H, 1500] of the latest wlk_decode (debug sessions only),
 * "self_k:<layer>" [n_rows, len, d], "xattn_w:<rank>" [window rows, 1500]. */
int wlk_export(wlk_session* s, const char* what, float* host, uint64_t capacity, uint64_t* n_written);

/* Per-kernel timing of the session's stream with HIP events (bench.py roofline leg).
 * wlk_prof_begin arms it; wlk_prof_end fills, per launch tag (up to cap entries): summed event
 * time in ms, launch count, and the summed ALGORITHMIC flops / bytes of those launches. */
int wlk_prof_begin(wlk_session* s);
int wlk_prof_end(wlk_session* s, int cap, const char** names, float* total_ms, int32_t* launches,
                 double* flops, double* bytes, int32_t* n_out);

/* Wall time of the session's single-token decode steps so far (graph-replayed steps of wlk_decode_until_stop outside
 * the batch engine): number of steps, nanoseconds from entering the step to holding its result (graph launch + the 46
 * kernels + the flag wait), and the share of that spent inside hipGraphLaunch.  What bench.py's roofline.step divides the
 * step's algorithmic bytes by. */
int wlk_session_step_stats(wlk_session* s, uint64_t* steps, uint64_t* wall_ns, uint64_t* launch_ns);

/* ---- a12 front end: log-mel features of the streaming Sortformer diarizer ----------------------
 * Replaces NeMo's AudioToMelSpectrogramPreprocessor.get_features as called at
 * whisperlivekit/diarization/sortformer_backend.py:181-188,273-275 (window 25 ms, stride 10 ms, n_fft 512,
 * 128 mel bins, normalize "NA", pre-emphasis 0.97, log(x + 2^-24)).  filters: [n_mels, n_fft/2+1];
 * window: [win_length] (symmetric hann).  Output is time-major [n_frames, n_mels], n_frames = n/hop + 1. */
typedef struct wlk_melspec wlk_melspec;
int wlk_melspec_create(int device, int n_fft, int win_length, int hop, int n_mels, const float* filters,
                       const float* window, float preemph, float log_guard, int max_samples, wlk_melspec** out);
int wlk_melspec_run(wlk_melspec* m, const float* pcm_host, int n, float* out_host, int capacity_frames,
                    int* n_frames);
int wlk_melspec_destroy(wlk_melspec* m);

/* ---- a12 network: the streaming Sortformer diarizer ------------------------------------------------
 * Replaces what the reference gets from NeMo's SortformerEncLabelModel at
 * whisperlivekit/diarization/sortformer_backend.py:108-131 (load, streaming configuration) and :293-300
 * (forward_streaming_step): ConvSubsampling(dw_striding, x8) -> 17 Conformer blocks (relative-position
 * attention) -> Linear -> 18 post-LN Transformer blocks -> sigmoid speaker head.  NeMo is a third-party
 * dependency that is not in the reference tree: packed tensor names/layout are defined by wlk_sf_tensor_name /
 * wlk_sf_tensor_lookup, the host shim (whisperlivekit_amd/sortformer.py) maps NeMo state-dict names onto them.
 * The streaming speaker-cache / FIFO bookkeeping (SortformerModules.streaming_update_async) is host logic. */
typedef struct wlk_sortformer wlk_sortformer;
typedef struct wlk_sf_dims {
    int32_t n_mels;        /* 128 */
    int32_t sub_channels;  /* 256: conv channels of the sub-sampling stem */
    int32_t fc_d_model, fc_layers, fc_heads, fc_ff, conv_kernel;   /* 512, 17, 8, 2048, 9 */
    int32_t tf_d_model, tf_layers, tf_heads, tf_inner;             /* 192, 18, 8, 768 */
    int32_t n_spk;         /* 4 */
    int32_t max_frames;    /* longest [spkcache | fifo | chunk] sequence, <= 512 */
    int32_t max_feat_frames; /* longest feature chunk handed to wlk_sf_step */
    float xscale;          /* sqrt(fc_d_model) when the encoder was trained with xscaling, else 1 */
} wlk_sf_dims;
int wlk_sf_arena_floats(const wlk_sf_dims* dims, uint64_t* n_floats);
int wlk_sf_tensor_lookup(const wlk_sf_dims* dims, const char* packed_name, uint64_t* offset_floats, uint64_t* numel);
int wlk_sf_tensor_name(const wlk_sf_dims* dims, int index, const char** name);
int wlk_sf_create(const wlk_sf_dims* dims, int device, wlk_sortformer** out);
int wlk_sf_upload(wlk_sortformer* m, const char* packed_name, const float* host, uint64_t numel);
/* precomputes linear_pos(pos_emb) of every Conformer block for max_frames (input independent) */
int wlk_sf_finalize(wlk_sortformer* m);
/* One streaming step (the device part of forward_streaming_step): feats_host [n_feat, n_mels] time-major log-mel
 * -> pre_encode -> chunk embeddings [n_chunk, fc_d_model] (returned in chunk_embs_host, n_chunk in *n_chunk);
 * ctx_embs_host [n_ctx, fc_d_model] = the caller's valid speaker-cache rows followed by its valid FIFO rows;
 * the network runs over [ctx | chunk] and preds_host receives [n_ctx + n_chunk, n_spk] sigmoid activities.
 * n_feat == 0 runs the network over ctx only; n_ctx == 0 over the chunk only.  Thread-safe; sessions keep their state on
 * the host.  Calls of several threads that wait for the model at the same time run as ONE stacked launch chain (their rows
 * one after the other, up to 8 sessions); a session's outputs are bit for bit those of its step alone. */
int wlk_sf_step(wlk_sortformer* m, const float* feats_host, int n_feat, const float* ctx_embs_host, int n_ctx,
                float* chunk_embs_host, int chunk_capacity_rows, int* n_chunk, float* preds_host,
                int preds_capacity_rows);
/* The same step with its front end inside (what sortformer_backend.py:273-300 does in three calls): pcm_host [n_pcm] is the new
 * audio chunk; its log-mel rows (n_pcm / hop + 1 of them, rows from valid_frames on zeroed - FilterbankFeatures' seq_len rule -
 * unless valid_frames < 0) are computed by `mel`'s kernel on the step's own stream, placed behind prev_feats_host [n_prev, n_mels]
 * (the rows the caller kept from its previous chunk, :279-283) and handed to the stem; feats_out_host receives the new rows
 * (*n_feats_out of them) so the caller can keep them for its next chunk.  Bit-identical to wlk_melspec_run + wlk_sf_step. */
int wlk_sf_step_pcm(wlk_sortformer* m, wlk_melspec* mel, const float* pcm_host, int n_pcm, int valid_frames,
                    const float* prev_feats_host, int n_prev, float* feats_out_host, int feats_capacity_rows, int* n_feats_out,
                    const float* ctx_embs_host, int n_ctx, float* chunk_embs_host, int chunk_capacity_rows, int* n_chunk,
                    float* preds_host, int preds_capacity_rows);
/* stacked launch chains run so far, the session steps inside them, their rows (batching statistics of the benchmark) */
int wlk_sf_stats(wlk_sortformer* m, uint64_t* stacked_steps, uint64_t* session_steps, uint64_t* rows);
/* parity/debug export of the last step (its first session): "fc_out" [T, fc_d_model] (Conformer output), "tf_out" [T, tf_d_model] */
int wlk_sf_export(wlk_sortformer* m, const char* what, float* host, uint64_t capacity, uint64_t* n_written);
int wlk_sf_destroy(wlk_sortformer* m);

/* ---- (f-next, rank 3) Silero VAD gate -------------------------------------------------------------------
 * Replaces the TorchScript model the reference evaluates on the CPU once per 512-sample window
 * (whisperlivekit/silero_vad_iterator.py:163-184 load_jit_vad; called at :254 from VADIterator.__call__, driven by
 * FixedVADIterator :288-319 from audio_processor.py:1189-1190).  16 kHz model only.  The packed weight layout is
 * defined by wlk_vad_tensor_name / wlk_vad_tensor_lookup (the host shim transposes the archive's tensors).
 * A wlk_vad holds the weights of one GPU; a wlk_vad_stream holds what the TorchScript wrapper keeps per stream:
 * the 64-sample context and the LSTM (h, c).  wlk_vad_stream_run consumes n_windows x 512 samples and returns one
 * speech probability per window (two launches whatever n_windows is). */
typedef struct wlk_vad wlk_vad;
typedef struct wlk_vad_stream wlk_vad_stream;
int wlk_vad_weights_floats(uint64_t* n_floats);
int wlk_vad_tensor_lookup(const char* packed_name, uint64_t* offset_floats, uint64_t* numel);
int wlk_vad_tensor_name(int index, const char** name);
int wlk_vad_create(int device, const float* packed_host, uint64_t n_floats, wlk_vad** out);
int wlk_vad_destroy(wlk_vad* m);
int wlk_vad_stream_create(wlk_vad* m, int max_windows, wlk_vad_stream** out);
int wlk_vad_stream_reset(wlk_vad_stream* s);                       /* model.reset_states() */
int wlk_vad_stream_run(wlk_vad_stream* s, const float* pcm_host, int n_windows, float* probs_host);
int wlk_vad_stream_state(wlk_vad_stream* s, float* h_host /* [128] */, float* c_host /* [128] */);
int wlk_vad_stream_destroy(wlk_vad_stream* s);

/* ---- diagnostics: one kernel on host data (used by the GPU parity tests only) ---------------- */
const char* wlk_diag_last_error(void);
/* environment switches the library caches on first use (WLK_PREFILL_LN_FUSE) are read again at their next use: for tests
 * that flip one inside a process */
int wlk_diag_env_refresh(void);
/* c[m,n] = epilogue(a[m,k](row stride lda, a_floats floats in total) . w[n,k]^T + bias); flags:
 * 1 = exact-erf GELU, 2 = add r[m, ldr] after the activation, 4 = scale columns < scale_cols,
 * 8 = ReLU, 16 = Swish.  force_gemv: 0 = shape-based choice among the MFMA kernels, 1 = weight-streaming kernel
 * (m <= 8), 2 = the k-wave MFMA kernel of under-filled grids (four waves split K of one 32x32 tile), 3 = the 64x64
 * kernel, 4 = the one-tile-per-CU k-pipe kernel, 5 = the "kp" family of the stacked Sortformer steps (k-wave tiles below
 * 512 rows, k-pipe tiles from there on: one per-element arithmetic, so a row's result does not depend on m) */
int wlk_diag_linear(const float* a, int64_t lda, int64_t a_floats, const float* w, const float* bias,
                    const float* r, int64_t ldr, int m, int n, int k, int flags, float scale, int scale_cols,
                    int force_gemv, float* c);
/* ---- NLLB-200 / M2M-100 translation network --------------------------------------------
 * The reference loads it through the third-party `nllw` package (whisperlivekit/core.py:320-329 `load_model`,
 * translation.py / core.py:483-493 `OnlineTranslation`), which wraps the `transformers` M2M-100 (or its CTranslate2
 * conversion); neither is in the reference tree, so these entry points replace what that backend executes: the encoder
 * over a source sentence, decoder steps with a KV cache, the tied vocabulary projection.  Layout and arithmetic:
 * csrc/nllb.hip.  Packed tensor names: "shared.emb" [vocab][d], "pos.table" [n_positions][d] (the sinusoidal table of
 * M2M100SinusoidalPositionalEmbedding, row = position id), "enc.<i>." / "dec.<i>." + ln1 / qkv (q | k | v rows) / out /
 * [lnx / xq / xkv (k | v rows) / xout] / ln2 / fc1 / fc2 + ".w" / ".b", "enc.ln", "dec.ln". */
typedef struct wlk_nllb_dims {
    int32_t vocab, d_model, heads, ffn, enc_layers, dec_layers;
    int32_t max_src, max_tgt;      /* longest source sentence / target sequence a session is sized for (<= 512 each) */
    int32_t pad_id;                /* padding_idx: position ids start at pad_id + 1 */
    int32_t n_positions;           /* rows of "pos.table" */
    float embed_scale;             /* sqrt(d_model) when config.scale_embedding, else 1 */
} wlk_nllb_dims;
typedef struct wlk_nllb wlk_nllb;
typedef struct wlk_nllb_session wlk_nllb_session;
int wlk_nllb_arena_floats(const wlk_nllb_dims* dims, uint64_t* n_floats);
int wlk_nllb_tensor_lookup(const wlk_nllb_dims* dims, const char* packed_name, uint64_t* offset_floats, uint64_t* numel);
int wlk_nllb_tensor_name(const wlk_nllb_dims* dims, int index, const char** name);
int wlk_nllb_create(const wlk_nllb_dims* dims, int device, wlk_nllb** out);
int wlk_nllb_upload(wlk_nllb* m, const char* packed_name, const float* host, uint64_t numel);
int wlk_nllb_finalize(wlk_nllb* m);
int wlk_nllb_destroy(wlk_nllb* m);
/* rows = hypotheses decoded side by side against one source sentence (1 = greedy, n = beams); own HIP stream */
int wlk_nllb_session_create(wlk_nllb* m, int rows, wlk_nllb_session** out);
int wlk_nllb_session_destroy(wlk_nllb_session* s);
/* M2M100Encoder.forward over one unpadded sentence (n ids, host int64) + the cross-attention K/V of every decoder layer */
int wlk_nllb_encode(wlk_nllb_session* s, const int64_t* src_ids, int32_t n);
/* M2M100Decoder.forward + lm_head for tokens [n_rows][n_tok] (host int64); first != 0 empties the self-attention cache
 * (the decoder prompt, e.g. [</s>, target language]); afterwards one token per row.  Logits of the last position of
 * every row stay on the device for wlk_nllb_topk / wlk_nllb_export.  Asynchronous on the session stream. */
int wlk_nllb_decode(wlk_nllb_session* s, const int64_t* tokens, int32_t n_rows, int32_t n_tok, int32_t first);
/* One single-token step per row after the prompt, with its read-out: wlk_nllb_decode(tokens, n_tok = 1, first = 0) +
 * wlk_nllb_topk(k) as ONE graph replay (inputs read from / results written to a host-coherent block by the kernels
 * themselves: no copy nodes).  Synchronous. */
int wlk_nllb_step(wlk_nllb_session* s, const int64_t* tokens, int32_t n_rows, int32_t k, float* logprobs, int32_t* ids);
/* beam bookkeeping: row i of the self-attention cache becomes the old row source_rows[i] */
int wlk_nllb_kv_reorder(wlk_nllb_session* s, const int32_t* source_rows, int32_t n_rows);
/* log_softmax(logits) of the latest decode, the k (<= 8) best per row, descending: [rows][k] */
int wlk_nllb_topk(wlk_nllb_session* s, int32_t k, float* logprobs, int32_t* ids);
/* parity exports: "logits" [rows][vocab] of the latest decode, "enc" [src_len][d_model] */
int wlk_nllb_export(wlk_nllb_session* s, const char* what, float* host, uint64_t capacity, uint64_t* n_written);
int wlk_nllb_sync(wlk_nllb_session* s);

/* ---- word-timestamp alignment ------------------------------------------------------------------
 * Dynamic time warping of a [n_rows tokens, n_cols frames] fp32 cost matrix on `device`: replaces `dtw_cpu`
 * (whisperlivekit/whisper/timing.py:82-105; CUDA counterpart `dtw_cuda` :108-138) under `find_alignment` (:163-243).
 * `trace` receives (n_rows + 1) x (n_cols + 1) step codes, row-major: 0 = diagonal, 1 = up (previous token), 2 = left
 * (previous frame), -1 in row 0 / column 0 - the array dtw_cpu hands to `backtrace` (:58-79).  Same strict comparisons
 * (ties go left) and one fp32 add per cell.  n_rows <= 1024.  Host pointers; synchronous. */
int wlk_dtw(int device, const float* x, int32_t n_rows, int32_t n_cols, int8_t* trace);
/* `log_mel_spectrogram(audio, n_mels, padding)` of a whole recording (whisperlivekit/whisper/audio.py:110-157) as
 * `whisper.transcribe()` computes it once per call (transcribe.py:126: padding = 30 s of zeros): STFT 400 / hop 160 with
 * reflection at both ends of the padded signal, mel filterbank, log10, clamp at (maximum over ALL frames) - 8, (x + 4) / 4.
 * `mel` receives [n_mels][n_frames], n_frames = (n_samples + padding) / 160; mel == NULL only reports n_frames.  Host
 * pointers; the work runs on the session's stream, synchronously.  n_samples + padding must exceed 200. */
int wlk_log_mel(wlk_session* s, const float* pcm, int64_t n_samples, int32_t padding, float* mel, uint64_t capacity_floats,
                int32_t* n_frames);
/* Encode a log-mel SEGMENT instead of the session's audio: `mel` is [n_mels][3000] as `whisper.transcribe()` slices it
 * out of the file's log-mel and hands it to the model / to `find_alignment` (whisperlivekit/whisper/transcribe.py,
 * timing.py:163-171).  Runs encoder + cross-K/V; the session then decodes against it like after wlk_encode.  Host
 * pointer, synchronous upload. */
int wlk_encode_mel(wlk_session* s, const float* mel, int32_t n_frames);
/* The device half of `find_alignment` (whisperlivekit/whisper/timing.py:163-218) on an encoded beam-1 session:
 * `tokens` = [sot sequence (n_sot ids), <|notimestamps|>, text tokens, <|endoftext|>] (n_tokens ids).  One decoder pass
 * over them; token_probs[i] = softmax(logits[n_sot + i, :eot])[text token i]; the alignment heads' cross-attention
 * scores -> softmax over the first num_frames / 2 positions (x qk_scale) -> z-score over the token axis -> median-7
 * along the frames -> head mean -> rows [n_sot, n_tokens - 1) negated = the cost matrix [n_text + 1][num_frames / 2]
 * (copied to `cost` when not NULL) -> dtw: `trace` receives (n_text + 2) x (num_frames / 2 + 1) step codes as wlk_dtw
 * returns them.  Walking the path back and cutting it into words stays on the host (timing.py:216-243).  The session
 * needs a new prefill afterwards (its alignment window holds this call's scores). */
int wlk_find_alignment(wlk_session* s, const int64_t* tokens, int32_t n_tokens, int32_t n_sot, int32_t eot,
                       int32_t num_frames, float qk_scale, float* cost, int8_t* trace, float* token_probs);

/* kernel-tuning probe: average microseconds per launch over `reps` back-to-back launches of one linear layer on
 * device-resident pseudo-random operands (same `force_gemv` meaning as wlk_diag_linear) */
int wlk_diag_linear_time(int m, int n, int k, int flags, int force_gemv, int reps, float* us_per_launch);
/* c[m,n] = LayerNorm(a[m,k]; gamma, beta, eps 1e-5) . w[n,k]^T + bias, m <= 8: the fused pre-LN projections
 * of the decode-step (weight-streaming) path */
int wlk_diag_linear_ln(const float* a, const float* w, const float* bias, const float* gamma, const float* beta,
                       int m, int n, int k, int force_gemv, float* c);
int wlk_diag_layernorm(const float* x, const float* gamma, const float* beta, int rows, int d, float* y);
/* test hook of the engine's stacked prefills (csrc/api.hip: wlk_prefill_group): the prefills of n encoded beam-1 sessions of
 * one model (tokens concatenated, n_tok[i] each) as ONE launch chain; taken[i] = 0 for a session whose prompt does not
 * qualify for the stack (nothing is done for it).  A stacked session ends up exactly as after wlk_decode(first = 1). */
int wlk_diag_prefill_stack(wlk_session** sessions, const int64_t* tokens, const int32_t* n_tok, const int32_t* sot_index, int32_t n,
                           int32_t* taken);
/* prefill chains run by the engine's prefill lane and the sessions stacked in them */
int wlk_engine_prefill_stats(wlk_model* m, uint64_t* batches, uint64_t* sessions);
/* kernel-tuning probe: average microseconds per encoder self-attention launch (pseudo-random qkv, k_splits key ranges) */
int wlk_diag_encoder_attention_time(int t, int d, int n_head, int k_splits, int reps, float* us_per_launch);
/* qkv [t, 3d] with q and k pre-scaled -> softmax(q k^T) v per 64-wide head, out [t, d] */
int wlk_diag_encoder_attention(const float* qkv, int t, int d, int n_head, float* out);
/* the VALU wave butterflies of csrc/wave_ops.h (DPP / v_permlane{16,32}_swap) against the __shfl_xor loops they replace,
 * on one wave of 64 floats: ten rows of 64 results each (sum, max, 16-lane sum, xor 1 .. 32 exchanges, arg-max index) */
int wlk_diag_wave_ops(const float* in64, float* out640, float* ref640);
/* the X3 path of the encoder's wide projections (csrc/x3.h, gemm_x3.hip: fp32 operands as three bf16 planes, six bf16
 * MFMAs per fp32 product): C = epilogue(A W^T + bias) with both operands packed on the device; its timing probe; and the
 * LayerNorm that writes its result in that format, unpacked again (bit-identical to wlk_diag_layernorm) */
int wlk_diag_linear_x3(const float* a, const float* w, const float* bias, int m, int n, int k, int flags, float scale,
                       int scale_cols, float* c);
int wlk_diag_linear_x3_time(int m, int n, int k, int flags, int reps, float* us_per_launch);
int wlk_diag_layernorm_x3(const float* x, const float* gamma, const float* beta, int rows, int d, float* y);
/* encoder self-attention through the X3 path (csrc/attention_x3.hip; same contract as wlk_diag_encoder_attention) and
 * its timing probe */
int wlk_diag_encoder_attention_x3(const float* qkv, int t, int d, int n_head, float* out);
int wlk_diag_encoder_attention_x3_time(int t, int d, int n_head, int reps, float* us_per_launch);
/* the production operand route of that attention (qkv projection with the X3 epilogue -> attention) against the diagnostic
 * one (fp32 projection -> x3 pack -> attention) on x [t][d], w [3 d][d], bias [3 d]: the two outputs [t][d] must be equal
 * bit for bit */
int wlk_diag_qkv_x3_attention(const float* x, const float* w, const float* bias, int t, int d, int n_head, float scale,
                              float* out_epilogue, float* out_packed);

#ifdef __cplusplus
}
#endif
#endif /* WLK_HIP_H */
