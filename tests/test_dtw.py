"""SURVEY 8(f) rank 4, the DTW of the word-timestamp path (whisper/timing.py:58-152).

CPU: the oracle restatement and the product's host-side `backtrace` against known answers produced by the reference's
own `dtw_cpu` (scripts/gen_golden_dtw.py).  GPU: the HIP wavefront kernel through the C ABI against the same answers
(bit-exact: the path is integer work) and against the oracle's full trace array."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import helpers as H  # noqa: E402
from gen_golden_dtw import dtw_case_matrix  # noqa: E402
from oracle import timing_oracle  # noqa: E402

KAT = H.golden_npz("dtw_kat.npz")
CASES = [(str(n), str(k), *map(int, s)) for n, k, s in zip(KAT["names"], KAT["kinds"], KAT["shapes"])]


@pytest.mark.parametrize("name,kind,n,m,seed", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_dtw(name, kind, n, m, seed):
    x = dtw_case_matrix(kind, n, m, seed)
    path = timing_oracle.dtw(x)
    np.testing.assert_array_equal(path, KAT[f"path_{name}"])


@pytest.mark.parametrize("name,kind,n,m,seed", CASES[:9], ids=[c[0] for c in CASES[:9]])
def test_host_backtrace_matches_reference(name, kind, n, m, seed):
    """The product's host half (walking the step codes back) on the oracle's trace."""
    from whisperlivekit_amd import timing
    trace = timing_oracle.dtw_trace(dtw_case_matrix(kind, n, m, seed))
    np.testing.assert_array_equal(timing.backtrace(trace), KAT[f"path_{name}"])
    assert trace[0, 0] == -1        # backtrace works on a copy: the caller's array is untouched


def test_host_rejects_bad_input():
    from whisperlivekit_amd import timing
    with pytest.raises(ValueError):
        timing.dtw_trace(np.zeros((0, 5), dtype=np.float32))
    with pytest.raises(ValueError):
        timing.backtrace(np.full((3, 3), 7, dtype=np.int8))


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,n,m,seed", CASES, ids=[c[0] for c in CASES])
def test_hip_dtw_matches_reference(name, kind, n, m, seed):
    from whisperlivekit_amd import timing
    x = dtw_case_matrix(kind, n, m, seed)
    trace = timing.dtw_trace(x)
    want = timing_oracle.dtw_trace(x)
    np.testing.assert_array_equal(trace[1:, 1:], want[1:, 1:])          # every step code, not only the path
    assert (trace[0, :] == -1).all() and (trace[:, 0] == -1).all()
    np.testing.assert_array_equal(timing.dtw(x), KAT[f"path_{name}"])


@pytest.mark.gpu
def test_hip_dtw_limits():
    from whisperlivekit_amd import _lib, timing
    x = np.random.default_rng(0).standard_normal((1024, 40)).astype(np.float32)    # the row limit itself
    np.testing.assert_array_equal(timing.dtw(x), timing_oracle.dtw(x))
    with pytest.raises(_lib.WlkError):
        timing.dtw_trace(np.zeros((1025, 4), dtype=np.float32))
