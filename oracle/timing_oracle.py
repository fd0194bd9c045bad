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
