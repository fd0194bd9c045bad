// Host half of one AlignAtt decode loop (beam 1): see loop.hip.  Shared by wlk_decode_until_stop() and the
// cross-session batch engine.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/wlk_hip.h"

namespace wlk {

struct DecodeJob {
    wlk_loop_params P;
    std::vector<int64_t> seq;            // context + prompt + tokens generated so far
    int n_before;                        // length of seq when the loop started
    std::vector<int32_t> suppress_ids, blank_ids;
    std::vector<int32_t> step_tokens, step_frames;   // one entry per decode step that selected a token
    std::vector<float> step_sums;        // running sum of log-probs after each step (BeamSearchDecoder's sum_logprobs)
    int produced = 0;
    bool fresh = true;                   // the next decoder forward is the first of this infer (prefill)
    float sum_logprob = 0.f;
    float no_speech_prob = 0.f;
    int last_attend;
    int stop = WLK_STOP_NONE;

    DecodeJob(const wlk_loop_params& p, const int64_t* tokens, int n_tok, const int32_t* suppress, int n_sup,
              const int32_t* blank, int n_blank);
    bool begin_step();                                   // false: the loop is over (stop is set)
    bool no_speech(float prob);                          // true: the loop stops here
    void adjustments(std::vector<int32_t>& ids, std::vector<float>& deltas) const;   // unique ids, additive deltas
    bool consume(const float* top_lp, const int32_t* top_ids, int frame);            // false: the loop is over
    void fill(wlk_loop_result* r) const;
};

}  // namespace wlk
