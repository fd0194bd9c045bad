"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the reference's dynamic time warping,
``dtw_cpu`` + ``backtrace`` (whisperlivekit/whisper/timing.py:58-105), pinned by tests/golden/dtw_kat.npz, which
scripts/gen_golden_dtw.py produced by calling the reference's own functions.

The reference fills ``cost`` column by column with strict comparisons (``c0 < c1 and c0 < c2`` -> diagonal,
``c1 < c0 and c1 < c2`` -> up, else left); ``cost`` is a float32 array, so every cell is rounded to fp32.  Here one
anti-diagonal is one vectorised numpy step (the cells of a diagonal are independent), same comparisons, same rounding.
"""
import numpy as np


def dtw_trace(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    n, m = x.shape
    cost = np.full((n + 1, m + 1), np.inf, dtype=np.float32)
    trace = -np.ones((n + 1, m + 1), dtype=np.int8)
    cost[0, 0] = 0
    for k in range(2, n + m + 1):                     # cells (i, j) with i + j = k, 1 <= i <= n, 1 <= j <= m
        i = np.arange(max(1, k - m), min(n, k - 1) + 1)
        j = k - i
        c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
        diag = (c0 < c1) & (c0 < c2)
        up = ~diag & (c1 < c0) & (c1 < c2)
        best = np.where(diag, c0, np.where(up, c1, c2))
        cost[i, j] = (x[i - 1, j - 1].astype(np.float64) + best.astype(np.float64)).astype(np.float32)
        trace[i, j] = np.where(diag, 0, np.where(up, 1, 2))
    return trace


def backtrace(trace: np.ndarray) -> np.ndarray:
    trace = trace.copy()
    trace[0, :] = 2
    trace[:, 0] = 1
    i, j = trace.shape[0] - 1, trace.shape[1] - 1
    out = []
    while i > 0 or j > 0:
        out.append((i - 1, j - 1))
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    return np.array(out[::-1], dtype=np.int64).T.reshape(2, -1)


def dtw(x: np.ndarray) -> np.ndarray:
    return backtrace(dtw_trace(x))


# ---------------------------------------------------------------------------------------------------------------
# find_alignment's arithmetic (whisperlivekit/whisper/timing.py:176-218): one decoder pass over
# [sot sequence, <|notimestamps|>, text tokens, <|endoftext|>], the cross-attention scores of the alignment heads,
# softmax over the first num_frames // 2 encoder positions, z-score over the TOKEN axis (no epsilon here, unlike the
# streaming policy), median filter of width 7 along the frames, mean over the heads, the rows of the text tokens.
# Pinned by the matrices the reference's find_alignment handed to dtw (tests/golden/word_timing_kat.json.gz).
# ---------------------------------------------------------------------------------------------------------------
def alignment_cost(sd, dims, align_heads, mel, sot_sequence, no_timestamps, text_tokens, eot, num_frames,
                   medfilt_width: int = 7, qk_scale: float = 1.0):
    """-> (cost matrix [len(text_tokens) + 1, num_frames // 2] as handed to dtw, per-token probabilities)."""
    import torch

    from . import whisper_oracle as wo
    tokens = torch.tensor([[*sot_sequence, no_timestamps, *text_tokens, eot]], dtype=torch.int64)
    with torch.no_grad():
        xa = wo.encoder_forward(sd, dims, mel.unsqueeze(0) if mel.dim() == 2 else mel)
        logits, qks = wo.decoder_forward(sd, dims, tokens, xa, wo.DecoderCache(dims.n_text_layer))
        sampled = logits[0, len(sot_sequence):, :eot]
        probs = sampled.softmax(dim=-1)[np.arange(len(text_tokens)), list(text_tokens)].tolist()
        w = torch.stack([qks[l][0, h] for l, h in align_heads])              # heads x tokens x frames
        w = (w[:, :, : num_frames // 2] * qk_scale).softmax(dim=-1)
        std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
        w = wo.median_filter((w - mean) / std, medfilt_width)
        matrix = w.mean(dim=0)[len(sot_sequence): -1]
    return (-matrix).numpy().astype(np.float32), probs
