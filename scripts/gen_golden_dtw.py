#!/usr/bin/env python
"""Known answers for the DTW of the word-timestamp path, produced by the REFERENCE's own `dtw_cpu` / `backtrace`
(whisperlivekit/whisper/timing.py:58-105, numba replaced by an identity decorator: scripts/ref_stubs.py).

Runs in the build container only (needs /root/reference or WLK_REFERENCE_ROOT); writes tests/golden/dtw_kat.npz:
for every case the generator parameters of its cost matrix (tests rebuild it with `dtw_case_matrix`), and the warping path.  Cases: the shapes find_alignment produces (tens to 224 tokens x up to 1500
frames of a negated z-scored attention map), degenerate shapes (1 x 1, one row, one column), exact ties (constant and
quantised matrices - the strict comparisons decide), a ridge the path has to follow.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402

CASES = [  # name, kind, n, m, seed
    ("one_cell", "normal", 1, 1, 0), ("one_row", "normal", 1, 40, 1), ("one_col", "normal", 33, 1, 2),
    ("small", "normal", 7, 13, 3), ("tall", "normal", 40, 9, 4), ("medium", "attention", 64, 300, 5),
    ("ties_const", "const", 5, 9, 6), ("ties_quant", "quant", 24, 50, 7), ("ridge", "ridge", 30, 200, 8),
    ("window_30s", "attention", 224, 1500, 9), ("max_rows", "normal", 448, 700, 10),
]


def dtw_case_matrix(kind, n, m, seed):
    """Cost matrices as find_alignment hands them to dtw (`-matrix`, float32)."""
    rng = np.random.default_rng(seed)
    if kind == "normal":
        return rng.standard_normal((n, m)).astype(np.float32)
    if kind == "const":
        return np.full((n, m), 0.25, dtype=np.float32)
    if kind == "quant":
        return (rng.integers(-2, 3, size=(n, m)) * 0.5).astype(np.float32)
    if kind == "ridge":
        x = rng.standard_normal((n, m)).astype(np.float32) * 0.1
        for i in range(n):
            x[i, int(i * (m - 1) / max(n - 1, 1))] -= 5.0
        return x
    if kind == "attention":   # negated z-scored attention: a monotone band of strong negative cost + noise
        x = rng.standard_normal((n, m)).astype(np.float32)
        centre = np.linspace(0, m - 1, n)
        jj = np.arange(m)[None, :]
        x -= 4.0 * np.exp(-0.5 * ((jj - centre[:, None]) / 6.0) ** 2).astype(np.float32)
        return x
    raise ValueError(kind)


def main():
    ref_stubs.install()
    from whisperlivekit.whisper import timing as ref_timing

    out = {"names": np.array([c[0] for c in CASES]), "kinds": np.array([c[1] for c in CASES]),
           "shapes": np.array([[c[2], c[3], c[4]] for c in CASES], dtype=np.int64)}
    for name, kind, n, m, seed in CASES:
        x = dtw_case_matrix(kind, n, m, seed)
        # what dtw() does on the CPU path: dtw_cpu(x.double().cpu().numpy()) (timing.py:152)
        path = ref_timing.dtw_cpu(x.astype(np.float64))
        out[f"path_{name}"] = np.asarray(path, dtype=np.int64)
        print(name, x.shape, "path length", path.shape[1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dtw_kat.npz"), **out)


if __name__ == "__main__":
    main()
