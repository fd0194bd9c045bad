// the AlignAtt decode loop of one `infer` (beam 1) behind one call.
//
// What the reference does per generated token in Python (simul_whisper/align_att_base.py:206-286):
// decoder forward -> no-speech check (first step) -> blank / special-token suppression
// (simul_whisper.py:379-381, :161-172) -> DRY repetition penalty (align_att_base.py:492-537) ->
// BeamSearchDecoder.update with beam_size 1 (whisper/decoding.py:317-376) -> AlignAtt read-out ->
// the stop rules (completed / rewind / frame threshold).  Here the host half of that is `DecodeJob`
// (plain integer logic, no GPU), driven either by wlk_decode_until_stop() over one session's kernels
// or by the cross-session batch engine (engine.hip) over rows = sessions.  The wlk_job_* entry
// points expose the same logic without a GPU so the CPU tests can drive it with oracle numerics.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/wlk_hip.h"
#include "common.h"
#include "internal.h"
#include <cstdlib>

#include "loop.h"

namespace wlk {

DecodeJob::DecodeJob(const wlk_loop_params& p, const int64_t* tokens, int n_tok, const int32_t* suppress, int n_sup,
                     const int32_t* blank, int n_blank)
    : P(p), seq(tokens, tokens + n_tok), n_before(n_tok), suppress_ids(suppress, suppress + n_sup),
      blank_ids(blank, blank + n_blank), last_attend(p.last_attend_frame) {}

bool DecodeJob::begin_step() {
    if (stop != WLK_STOP_NONE) return false;
    if ((int)seq.size() >= P.max_text_len) {          // `while ... current_tokens.shape[1] < self.max_text_len`
        stop = WLK_STOP_CONTEXT_FULL;
        return false;
    }
    produced += 1;
    if (produced > P.budget) {                         // runaway guard, align_att_base.py:208-214
        seq.resize(n_before);
        stop = WLK_STOP_BUDGET;
        return false;
    }
    return true;
}

bool DecodeJob::no_speech(float prob) {
    no_speech_prob = prob;
    if (prob > P.no_speech_threshold) {                // simul_whisper.py:370-377
        stop = WLK_STOP_NO_SPEECH;
        return true;
    }
    return false;
}

void DecodeJob::adjustments(std::vector<int32_t>& ids, std::vector<float>& deltas) const {
    std::map<int32_t, float> adj;                      // one entry per token id (the device applies them in parallel)
    const float ninf = -INFINITY;
    if (fresh)
        for (int32_t t : blank_ids) adj[t] = ninf;     // _suppress_blank_tokens: encode(" ") + [eot]
    for (int32_t t : suppress_ids) adj[t] = ninf;      // SuppressTokens.apply
    // DRY penalty (align_att_base.py:492-537): for every earlier occurrence of the last token, the length of the
    // common suffix ending there; the token that followed the longest such match is penalised by 2^(len-2)
    const int n = (int)seq.size();
    if (n >= 5 && seq[n - 1] < P.eot) {
        const int64_t last = seq[n - 1];
        std::map<int64_t, int> longest;
        for (int i = n - 2; i >= 0; --i) {
            if (seq[i] != last || seq[i + 1] >= P.eot) continue;
            int m = 1;
            while (m < 50) {
                const int j = i - m, k = n - 1 - m;
                if (j < 0 || k <= i || seq[j] != seq[k] || seq[j] >= P.eot) break;
                ++m;
            }
            auto it = longest.find(seq[i + 1]);
            if (it == longest.end() || m > it->second) longest[seq[i + 1]] = m;
        }
        for (const auto& kv : longest)
            if (kv.second >= 2) {
                const float pen = (float)std::ldexp(1.0, kv.second - 2);   // 1.0 * 2.0 ** (m - 2)
                auto it = adj.find((int32_t)kv.first);
                if (it == adj.end()) adj[(int32_t)kv.first] = -pen;
                else it->second -= pen;                                     // -inf stays -inf
            }
    }
    ids.clear();
    deltas.clear();
    for (const auto& kv : adj) {
        ids.push_back(kv.first);
        deltas.push_back(kv.second);
    }
}

bool DecodeJob::consume(const float* top_lp_in, const int32_t* top_ids_in, int frame) {
    fresh = false;
    float top_lp[2] = {top_lp_in[0], top_lp_in[1]};
    int32_t top_ids[2] = {top_ids_in[0], top_ids_in[1]};
    // teacher forcing (parity harnesses; wlk_loop_params): this step takes the reference's side of an fp32 tie
    const int step_index = (int)step_tokens.size();
    for (int f = 0; f < P.n_force && f < WLK_MAX_FORCED; ++f) {
        if (P.force_step[f] != step_index) continue;
        if (P.force_frame[f] >= 0) frame = P.force_frame[f];
        if (P.force_token[f] >= 0 && top_ids[0] != P.eot && top_ids[1] == P.force_token[f]) {
            std::swap(top_ids[0], top_ids[1]);
            std::swap(top_lp[0], top_lp[1]);
        }
    }
    // BeamSearchDecoder.update, beam_size 1: candidates are the two best tokens; an end-of-text candidate goes to the
    // finished list (and completes the search: max_candidates = 1), the best other candidate is the next token
    bool completed = false;
    int32_t next;
    if (top_ids[0] == P.eot) {
        next = top_ids[1];
        sum_logprob = sum_logprob + top_lp[1];
        completed = true;
    } else {
        next = top_ids[0];
        sum_logprob = sum_logprob + top_lp[0];
    }
    seq.push_back(next);
    step_tokens.push_back(next);
    step_frames.push_back(frame);
    step_sums.push_back(sum_logprob);
    if (completed) {                                                // :255-257
        seq.pop_back();
        stop = WLK_STOP_COMPLETED;
        return false;
    }
    if (!P.is_last && last_attend - frame > P.rewind_threshold) {   // :260-276
        const int n = (int)seq.size();
        if (n > 1 && seq[n - 2] >= P.dec_pad) {
            last_attend = frame;
        } else {
            last_attend = -P.rewind_threshold;
            seq.resize(n_before);                                   // _rewind_tokens(): nothing new is kept
            stop = WLK_STOP_REWIND;
            return false;
        }
    } else {
        last_attend = frame;
    }
    if (P.content_mel_len - frame <= (P.is_last ? 4 : P.frame_threshold)) {   // :280-286
        seq.pop_back();
        stop = WLK_STOP_FRAME;
        return false;
    }
    return true;
}

void DecodeJob::fill(wlk_loop_result* r) const {
    r->n_steps = (int32_t)step_tokens.size();
    r->n_new_tokens = (int32_t)seq.size() - n_before;
    r->stop_reason = stop;
    r->last_attend_frame = last_attend;
    r->no_speech_prob = no_speech_prob;
    r->sum_logprob = sum_logprob;
    r->decode_calls = produced - (stop == WLK_STOP_BUDGET ? 1 : 0);
}

}  // namespace wlk

using namespace wlk;

struct wlk_decode_job {
    DecodeJob job;
    std::vector<int32_t> ids;
    std::vector<float> deltas;
    wlk_decode_job(const wlk_loop_params& p, const int64_t* t, int n, const int32_t* s, int ns, const int32_t* b, int nb)
        : job(p, t, n, s, ns, b, nb) {}
};

static int loop_fail(int code, const std::string& msg) {
    set_last_error(msg);
    return code;
}

static int check_loop_args(const wlk_loop_params* p, const int64_t* tokens, int n_tok, const int32_t* suppress, int n_sup,
                           const int32_t* blank, int n_blank) {
    if (!p || !tokens || n_tok < 1) return loop_fail(WLK_ERR_ARG, "decode loop: NULL / empty argument");
    if (n_sup < 0 || n_blank < 0 || (n_sup > 0 && !suppress) || (n_blank > 0 && !blank))
        return loop_fail(WLK_ERR_ARG, "decode loop: bad suppression lists");
    if (p->sot_index < 0 || p->sot_index >= n_tok) return loop_fail(WLK_ERR_ARG, "decode loop: sot_index out of range");
    if (p->max_text_len < 1 || p->budget < 0) return loop_fail(WLK_ERR_ARG, "decode loop: bad limits");
    if (p->n_force < 0 || p->n_force > WLK_MAX_FORCED) return loop_fail(WLK_ERR_ARG, "decode loop: n_force out of range");
    return WLK_OK;
}

static int copy_result(const DecodeJob& j, wlk_loop_result* result, int64_t* new_tokens, int32_t* step_tokens,
                       int32_t* step_frames, float* step_sum_logprobs, int cap) {
    if (!result) return loop_fail(WLK_ERR_ARG, "decode loop: result is NULL");
    j.fill(result);
    if (result->n_steps > cap || result->n_new_tokens > cap) return loop_fail(WLK_ERR_CAPACITY, "decode loop: output capacity too small");
    for (int i = 0; i < result->n_new_tokens; ++i)
        if (new_tokens) new_tokens[i] = j.seq[j.n_before + i];
    for (int i = 0; i < result->n_steps; ++i) {
        if (step_tokens) step_tokens[i] = j.step_tokens[i];
        if (step_frames) step_frames[i] = j.step_frames[i];
        if (step_sum_logprobs) step_sum_logprobs[i] = j.step_sums[i];
    }
    return WLK_OK;
}

extern "C" {

int wlk_job_create(const wlk_loop_params* p, const int64_t* tokens, int n_tok, const int32_t* suppress_ids, int n_suppress,
                   const int32_t* blank_ids, int n_blank, wlk_decode_job** out) {
    if (int rc = check_loop_args(p, tokens, n_tok, suppress_ids, n_suppress, blank_ids, n_blank)) return rc;
    if (!out) return loop_fail(WLK_ERR_ARG, "decode loop: out is NULL");
    *out = new wlk_decode_job(*p, tokens, n_tok, suppress_ids, n_suppress, blank_ids, n_blank);
    return WLK_OK;
}

int wlk_job_begin_step(wlk_decode_job* j, int32_t* n_feed) {
    if (!j || !n_feed) return loop_fail(WLK_ERR_ARG, "decode loop: NULL argument");
    const bool first = j->job.fresh;
    if (!j->job.begin_step()) {
        *n_feed = 0;
        return WLK_OK;
    }
    *n_feed = first ? (int32_t)j->job.seq.size() : 1;
    return WLK_OK;
}

int wlk_job_no_speech(wlk_decode_job* j, float prob, int32_t* stops) {
    if (!j || !stops) return loop_fail(WLK_ERR_ARG, "decode loop: NULL argument");
    *stops = j->job.no_speech(prob) ? 1 : 0;
    return WLK_OK;
}

int wlk_job_adjustments(wlk_decode_job* j, const int32_t** ids, const float** deltas, int32_t* n) {
    if (!j || !ids || !deltas || !n) return loop_fail(WLK_ERR_ARG, "decode loop: NULL argument");
    j->job.adjustments(j->ids, j->deltas);
    *ids = j->ids.data();
    *deltas = j->deltas.data();
    *n = (int32_t)j->ids.size();
    return WLK_OK;
}

int wlk_job_consume(wlk_decode_job* j, const float* top_logprobs, const int32_t* top_ids, int frame, int32_t* goes_on) {
    if (!j || !top_logprobs || !top_ids || !goes_on) return loop_fail(WLK_ERR_ARG, "decode loop: NULL argument");
    *goes_on = j->job.consume(top_logprobs, top_ids, frame) ? 1 : 0;
    return WLK_OK;
}

int wlk_job_result(wlk_decode_job* j, wlk_loop_result* result, int64_t* new_tokens, int32_t* step_tokens,
                   int32_t* step_frames, float* step_sum_logprobs, int cap) {
    if (!j) return loop_fail(WLK_ERR_ARG, "decode loop: job is NULL");
    return copy_result(j->job, result, new_tokens, step_tokens, step_frames, step_sum_logprobs, cap);
}

int wlk_job_destroy(wlk_decode_job* j) {
    delete j;
    return WLK_OK;
}

int wlk_decode_until_stop(wlk_session* s, const int64_t* tokens, int n_tok, const wlk_loop_params* p,
                          const int32_t* suppress_ids, int n_suppress, const int32_t* blank_ids, int n_blank,
                          wlk_loop_result* result, int64_t* new_tokens, int32_t* step_tokens, int32_t* step_frames,
                          float* step_sum_logprobs, int cap) {
    if (!s) return loop_fail(WLK_ERR_ARG, "decode loop: session is NULL");
    if (int rc = check_loop_args(p, tokens, n_tok, suppress_ids, n_suppress, blank_ids, n_blank)) return rc;
    if (s->beam != 1) return loop_fail(WLK_ERR_ARG, "decode loop: the session's beam must be 1");
    DecodeJob job(*p, tokens, n_tok, suppress_ids, n_suppress, blank_ids, n_blank);
    std::vector<int32_t> ids, rows;
    std::vector<float> deltas;
    struct InLoop {   // the engine counts the loops in flight on this GPU (see wlk_engine_wants)
        wlk_session* s;
        explicit InLoop(wlk_session* ss) : s(ss) { wlk_engine_loop_enter(s); }
        ~InLoop() { wlk_engine_loop_exit(s); }
    } in_loop(s);
    while (job.begin_step()) {
        const bool first = job.fresh;
        const int64_t* feed = first ? job.seq.data() : job.seq.data() + job.seq.size() - 1;
        float lp[2];
        int32_t top[2], frame = 0;
        if (!first) {
            // single-token step and its read-out as one graph replay; 1 = this session / step does not qualify
            job.adjustments(ids, deltas);
            const int rc = wlk_step_select(s, feed[0], ids.data(), deltas.data(), (int)ids.size(), p->content_mel_len, lp,
                                           top, &frame);
            if (rc == WLK_OK) {
                if (!job.consume(lp, top, frame)) break;
                if (wlk_engine_wants(s)) {
                    if (int rc2 = wlk_engine_run_job(s, &job)) return rc2;
                    break;
                }
                continue;
            }
            if (rc != 1) return rc;
        }
        // the prefill rides in the engine's stacked chain when other sessions are busy on this GPU (1 = not taken)
        int taken = first ? wlk_engine_prefill(s, feed, (int)job.seq.size(), p->sot_index) : 1;
        if (taken != WLK_OK && taken != 1) return taken;
        if (taken == 1)
            if (int rc = wlk_decode(s, feed, 1, first ? (int)job.seq.size() : 1, first ? 1 : 0, p->sot_index)) return rc;
        job.adjustments(ids, deltas);
        rows.assign(ids.size(), -1);
        static const bool merged_first = getenv("WLK_NO_FIRST_MERGE") == nullptr;   // A/B switch
        if (first && p->no_speech_token >= 0 && !merged_first) {
            float prob = 0.f;
            if (int rc = wlk_no_speech_prob(s, p->no_speech_token, &prob)) return rc;
            if (job.no_speech(prob)) break;
            if (int rc = wlk_select(s, rows.data(), ids.data(), deltas.data(), (int)ids.size(), 2, p->content_mel_len, lp, top,
                                    &frame))
                return rc;
        } else if (first && p->no_speech_token >= 0) {
            // no-speech probability and read-out share one synchronisation; a no-speech stop ignores the read-out
            float prob = 0.f;
            if (int rc = wlk_select_first(s, p->no_speech_token, rows.data(), ids.data(), deltas.data(), (int)ids.size(), 2,
                                          p->content_mel_len, &prob, lp, top, &frame))
                return rc;
            if (job.no_speech(prob)) break;
        } else if (int rc = wlk_select(s, rows.data(), ids.data(), deltas.data(), (int)ids.size(), 2, p->content_mel_len, lp,
                                       top, &frame)) {
            return rc;
        }
        if (!job.consume(lp, top, frame)) break;
        if (wlk_engine_wants(s)) {
            // the prefill is done: the single-token steps of this loop advance together with the loops of the other
            // sessions of this GPU (engine.hip); returns when this loop has stopped
            if (int rc = wlk_engine_run_job(s, &job)) return rc;
            break;
        }
    }
    return copy_result(job, result, new_tokens, step_tokens, step_frames, step_sum_logprobs, cap);
}

}  // extern "C"
