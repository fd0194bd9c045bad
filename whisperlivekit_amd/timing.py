"""Word-timestamp alignment pieces of LocalAgreement's batch Whisper on the HIP library.

Mirrors the reference names of ``whisperlivekit/whisper/timing.py``: ``dtw`` (:141-152) returns the warping path as
``[text_indices, time_indices]`` exactly as ``dtw_cpu`` + ``backtrace`` do (:58-105).  The recurrence runs on the GPU
(``wlk_dtw``, csrc/dtw.hip); walking the trace back is a few hundred integer steps and stays on the host.  There is no
CPU fallback: without the library / a GPU this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

MAX_ROWS = 1024


def backtrace(trace: np.ndarray) -> np.ndarray:
    """timing.py:58-79: from the last cell back to the origin; row 0 only moves left, column 0 only moves up."""
    steps = np.array(trace, dtype=np.int8, copy=True)
    steps[0, :] = 2
    steps[:, 0] = 1
    i, j = steps.shape[0] - 1, steps.shape[1] - 1
    path = []
    while i > 0 or j > 0:
        path.append((i - 1, j - 1))
        code = steps[i, j]
        if code == 0:
            i, j = i - 1, j - 1
        elif code == 1:
            i -= 1
        elif code == 2:
            j -= 1
        else:
            raise ValueError(f"unexpected trace code {code} at ({i}, {j})")
    return np.array(path[::-1], dtype=np.int64).T.reshape(2, -1)


def dtw_trace(x: np.ndarray, device: int = 0) -> np.ndarray:
    """The (N + 1) x (M + 1) step-code array of ``dtw_cpu`` for the cost matrix ``x`` [N tokens, M frames]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2 or x.shape[0] < 1 or x.shape[1] < 1:
        raise ValueError(f"dtw: expected a non-empty 2-D cost matrix, got shape {x.shape}")
    n, m = x.shape
    trace = np.empty((n + 1, m + 1), dtype=np.int8)
    lib = _lib.load()
    _lib.check(lib.wlk_dtw(int(device), x.ctypes.data_as(C.POINTER(C.c_float)), n, m,
                           trace.ctypes.data_as(C.POINTER(C.c_int8))))
    return trace


def dtw(x, device: int = 0) -> np.ndarray:
    """timing.py:141-152: ``text_indices, time_indices = dtw(-matrix)``."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return backtrace(dtw_trace(np.asarray(x), device))
