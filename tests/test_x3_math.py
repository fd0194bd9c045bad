"""The arithmetic behind csrc/x3.h, checked on the CPU: an fp32 value is EXACTLY the sum of three round-to-nearest bf16
planes, every bf16 x bf16 product is exact in fp32, and the six plane products the X3 kernels keep reproduce an fp32
product to below fp32's own half-ulp - so a GEMM built from them (fp32 accumulation) is as accurate as an fp32 GEMM.
(The kernels themselves are compared with float64 and with the fp32-MFMA kernels on the GPU: tests/test_gpu_x3.py.)"""
import numpy as np


def bf16_round(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even of fp32 values to bf16, returned as fp32 (what `(__bf16)x` does on gfx950)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def x3_split(x):
    x = np.asarray(x, np.float32)
    h = bf16_round(x)
    r = (x - h).astype(np.float32)
    m = bf16_round(r)
    r2 = (r - m).astype(np.float32)
    lo = bf16_round(r2)
    return h, m, lo


PA = (2, 0, 1, 1, 0, 0)      # plane of a / plane of b of the six products, small terms first (gemm_x3.hip, attention_x3.hip)
PB = (0, 2, 1, 0, 1, 0)


def test_three_bf16_planes_are_the_fp32_value_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200_000).astype(np.float32),
                        (rng.standard_normal(200_000) * np.exp(rng.uniform(-60, 60, 200_000))).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 65504.0, 1.0 + 2.0 ** -23], np.float32)])
    h, m, lo = x3_split(x)
    # the residuals are exact in fp32 (Sterbenz-style: each plane takes the leading 8 bits of what is left) ...
    assert np.array_equal(((h.astype(np.float64) + m) + lo).astype(np.float32), x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64)), x.astype(np.float64))
    # ... and the planes shrink by 2^-8 each (round to nearest: |rest| <= half an ulp of the plane above)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_bf16_products_are_exact_in_fp32_and_six_of_them_are_an_fp32_product():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(300_000).astype(np.float32) * np.float32(3.0)
    b = (rng.standard_normal(300_000) / 20).astype(np.float32)
    pa, pb = x3_split(a), x3_split(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = np.zeros_like(exact)
    for t in range(6):
        p32 = pa[PA[t]] * pb[PB[t]]                                              # fp32 multiply of two bf16 values ...
        assert np.array_equal(p32.astype(np.float64), pa[PA[t]].astype(np.float64) * pb[PB[t]].astype(np.float64))   # ... is exact
        six += p32.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -24          # the three dropped terms: below half an ulp of the fp32 product
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -26


def test_a_gemm_from_six_plane_products_is_as_accurate_as_an_fp32_gemm():
    rng = np.random.default_rng(2)
    M, N, K = 48, 64, 512
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    pa, pw = x3_split(a), x3_split(w)
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):                                   # one MFMA K-step at a time, fp32 accumulation, small terms first
        for t in range(6):
            acc = (acc + (pa[PA[t]][:, k0:k0 + 16].astype(np.float64) @ pw[PB[t]][:, k0:k0 + 16].astype(np.float64).T)
                   .astype(np.float32)).astype(np.float32)
    f32 = np.zeros((M, N), np.float32)
    for k in range(K):                                           # a plain fp32 fma chain, as v_mfma_f32_32x32x2_f32 runs it
        f32 = (f32.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * w[:, k].astype(np.float64)[None, :]).astype(np.float32)
    scale = np.abs(ref).mean()
    err_x3 = np.abs(acc - ref).mean() / scale
    err_f32 = np.abs(f32 - ref).mean() / scale
    assert err_x3 < 1.5 * err_f32 + 1e-9, (err_x3, err_f32)
    assert err_x3 < 2e-6
    bf = bf16_round(a).astype(np.float64) @ bf16_round(w).astype(np.float64).T   # what a plain bf16 GEMM would give
    assert np.abs(bf - ref).mean() / scale > 300 * err_x3
