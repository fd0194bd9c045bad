"""The X3 path (csrc/x3.h, gemm_x3.hip): the encoder's wide projections on the bf16 matrix cores at fp32 accuracy -
operands held as three bf16 planes (hi + mid + lo == the fp32 value), six bf16 MFMAs per fp32 product.  The claim to
check is "fp32 accuracy": against float64 the result must be as close as the fp32-MFMA kernel's (and as torch's own fp32
GEMM, which is what the reference computes with on the CPU)."""
import ctypes as C

import numpy as np
import pytest
import torch

from whisperlivekit_amd import _lib

pytestmark = pytest.mark.gpu
vp = lambda a: a.ctypes.data_as(C.c_void_p)


def _err(x, ref):
    scale = np.abs(ref).mean()
    d = np.abs(x.astype(np.float64) - ref)
    return d.max() / scale, d.mean() / scale


@pytest.mark.parametrize("M,N,K,flags", [(1500, 2048, 512, 1), (1500, 1536, 512, 4), (1500, 6144, 512, 4), (333, 1152, 384, 0),
                                          (1500, 1280 * 3, 1280, 4), (257, 1024, 64, 1),
                                          (1500, 1280, 1280, 0), (700, 1024, 256, 1)])     # (64-row tiles by the launch rule)
def test_x3_gemm_is_as_close_to_float64_as_the_fp32_mfma_gemm(M, N, K, flags):
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    a[:, :5] *= 40.0                                   # a few loud channels, as a residual stream has
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if flags & 4:
        ref[:, :N // 2] *= 0.5
    if flags & 1:
        ref = 0.5 * ref * (1.0 + torch.erf(torch.from_numpy(ref) * 0.7071067811865476).numpy())
    c3 = np.empty((M, N), np.float32)
    assert lib.wlk_diag_linear_x3(vp(a), vp(w), vp(bias), M, N, K, flags, 0.5, N // 2, vp(c3)) == 0, lib.wlk_diag_last_error()
    c32 = np.empty((M, N), np.float32)
    assert lib.wlk_diag_linear(vp(a), K, M * K, vp(w), vp(bias), None, N, M, N, K, flags, 0.5, N // 2, 0, vp(c32)) == 0
    t = torch.from_numpy(a) @ torch.from_numpy(w).T + torch.from_numpy(bias)
    if flags & 4:
        t[:, :N // 2] *= 0.5
    if flags & 1:
        t = torch.nn.functional.gelu(t)
    e3, e32, et = _err(c3, ref), _err(c32, ref), _err(t.numpy(), ref)
    print(f"M{M} N{N} K{K}: x3 max/mean {e3[0]:.2e} {e3[1]:.2e} | fp32 mfma {e32[0]:.2e} {e32[1]:.2e} | torch cpu {et[0]:.2e} {et[1]:.2e}")
    assert e3[1] <= 2.0 * max(e32[1], et[1]) and e3[0] <= 3.0 * max(e32[0], et[0]), (e3, e32, et)
    assert e3[0] < 5e-5


def test_x3_layernorm_round_trip_is_the_fp32_layernorm():
    """The planes are an exact image: LayerNorm -> X3 -> (hi + mid) + lo equals the fp32 LayerNorm kernel's output bit for bit."""
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for rows, d in ((1500, 512), (77, 384), (300, 1280), (64, 768)):
        x = (rng.standard_normal((rows, d)) * 3.0).astype(np.float32)
        x[:, 7] += 100.0
        g = (1 + 0.2 * rng.standard_normal(d)).astype(np.float32)
        b = (0.3 * rng.standard_normal(d)).astype(np.float32)
        y = np.empty((rows, d), np.float32)
        y3 = np.empty((rows, d), np.float32)
        assert lib.wlk_diag_layernorm(vp(x), vp(g), vp(b), rows, d, vp(y)) == 0
        assert lib.wlk_diag_layernorm_x3(vp(x), vp(g), vp(b), rows, d, vp(y3)) == 0, lib.wlk_diag_last_error()
        assert np.array_equal(y.view(np.uint32), y3.view(np.uint32)), (rows, d, np.abs(y - y3).max())


@pytest.mark.parametrize("T,d,H", [(1500, 512, 8), (1500, 384, 6), (200, 128, 2), (1500, 1280, 20)])
def test_x3_encoder_attention_matches_float64_like_the_fp32_kernel(T, d, H):
    """softmax(Q K^T) V per 64-wide head through the bf16 matrix cores (three planes per operand, fp32 softmax): against a
    float64 reference the output must be as close as the fp32-MFMA attention kernel's."""
    lib = _lib.load()
    rng = np.random.default_rng(T + d)
    qkv = rng.standard_normal((T, 3 * d)).astype(np.float32)
    qkv[:, :2 * d] *= 0.6                                   # pre-scaled q and k: scores of a few units, as in the model
    qkv[:, 2 * d + 3] += 5.0
    q = qkv[:, :d].astype(np.float64).reshape(T, H, 64).transpose(1, 0, 2)
    k = qkv[:, d:2 * d].astype(np.float64).reshape(T, H, 64).transpose(1, 0, 2)
    v = qkv[:, 2 * d:].astype(np.float64).reshape(T, H, 64).transpose(1, 0, 2)
    s = q @ k.transpose(0, 2, 1)
    s -= s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(1, 0, 2).reshape(T, d)
    o3 = np.empty((T, d), np.float32)
    o32 = np.empty((T, d), np.float32)
    assert lib.wlk_diag_encoder_attention_x3(vp(qkv), T, d, H, vp(o3)) == 0, lib.wlk_diag_last_error()
    assert lib.wlk_diag_encoder_attention(vp(qkv), T, d, H, vp(o32)) == 0, lib.wlk_diag_last_error()
    e3, e32 = _err(o3, ref), _err(o32, ref)
    print(f"T{T} d{d}: x3 max/mean {e3[0]:.2e} {e3[1]:.2e} | fp32 mfma {e32[0]:.2e} {e32[1]:.2e}")
    assert e3[1] <= 2.0 * e32[1] + 1e-7 and e3[0] <= 3.0 * e32[0] + 1e-6, (e3, e32)


@pytest.mark.parametrize("T,d,H", [(1500, 512, 8), (1500, 384, 6), (1437, 128, 2), (97, 128, 2), (1500, 1280, 20)])
def test_production_operand_route_of_the_x3_attention_equals_the_diagnostic_one(T, d, H):
    """In the product the attention's operand image (q | k as X3 rows, V transposed in the lane order of the kernel) is
    written by the qkv projection's X3 epilogue; the accuracy tests above feed the kernel through x3_pack_qkv instead.
    Both routes on the same x, w: the attention outputs must be bit-identical (T = 1500 and tile counts that are not whole:
    1437 = 14 x 96 + 93 rows, 97 = one tile + 1)."""
    lib = _lib.load()
    rng = np.random.default_rng(T + d)
    x = rng.standard_normal((T, d)).astype(np.float32)
    w = (rng.standard_normal((3 * d, d)) / np.sqrt(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(3 * d)).astype(np.float32)
    o_epi = np.empty((T, d), np.float32)
    o_pack = np.empty((T, d), np.float32)
    rc = lib.wlk_diag_qkv_x3_attention(vp(x), vp(w), vp(b), T, d, H, 64 ** -0.25, vp(o_epi), vp(o_pack))
    assert rc == 0, lib.wlk_diag_last_error()
    assert np.isfinite(o_epi).all() and float(np.abs(o_epi).max()) > 1e-3
    assert np.array_equal(o_epi.view(np.uint32), o_pack.view(np.uint32)), float(np.abs(o_epi - o_pack).max())


@pytest.mark.parametrize("M,N,K,flags", [(1500, 6144, 512, 4), (1500, 5120, 1280, 1), (1500, 2048, 512, 3), (2999, 1536, 384, 0),
                                         (700, 1152, 64, 0)])
def test_persistent_walk_is_bit_identical_to_one_workgroup_per_tile(M, N, K, flags, monkeypatch):
    """Round 5: a workgroup of the wide kernel walks several tiles with ONE slab stream through its LDS ring (the next tile's
    first slabs land under the previous tile's epilogue).  Same MFMA sequence per tile, so the results must equal, bit for
    bit, the launch with one workgroup per tile (WLK_X3_PERSIST=0): 768 tiles = 3 per workgroup, 640 = ragged 3 / 2,
    GELU + residual-free epilogue, M that is no multiple of 96, K = two slabs (the ring never holds a whole tile ahead)."""
    lib = _lib.load()
    rng = np.random.default_rng(M + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    out = {}
    for persist in ("1", "0"):
        monkeypatch.setenv("WLK_X3_PERSIST", persist)
        assert lib.wlk_diag_env_refresh() == 0
        c = np.full((M, N), np.nan, np.float32)
        assert lib.wlk_diag_linear_x3(vp(a), vp(w), vp(bias), M, N, K, flags & 5, 0.5, N // 2, vp(c)) == 0, lib.wlk_diag_last_error()
        out[persist] = c
    monkeypatch.delenv("WLK_X3_PERSIST")
    assert lib.wlk_diag_env_refresh() == 0
    assert np.isfinite(out["1"]).all()
    assert np.array_equal(out["1"].view(np.uint32), out["0"].view(np.uint32)), float(np.abs(out["1"] - out["0"]).max())
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if not flags & 1:
        ref[:, :N // 2] *= 0.5 if flags & 4 else 1.0
        assert float(np.abs(out["1"] - ref).max()) < 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("M,N,K,flags", [(1500, 1280, 1280, 0), (1500, 1280, 5120, 4), (1500, 2048, 512, 1), (333, 1152, 384, 0),
                                         (2999, 1536, 384, 5), (64, 1024, 128, 0)])
def test_tile_height_and_wave_split_do_not_change_an_element(M, N, K, flags, monkeypatch):
    """Round 6: the two-wave kernel (a slab's k-steps split between the two compute waves of a SIMD) takes 64-row tiles where
    96-row tiles leave CUs idle (WLK_X3_BM forces either).  An element's arithmetic does not depend on the tile height: both
    heights give bit-identical results; against the one-wave kernel of round 5 (WLK_X3_KSPLIT=0), which groups a tile's K sum
    differently, the difference stays at rounding level."""
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    out = {}
    for name, env in (("bm96", {"WLK_X3_BM": "96"}), ("bm64", {"WLK_X3_BM": "64"}), ("one_wave", {"WLK_X3_KSPLIT": "0"})):
        for k in ("WLK_X3_BM", "WLK_X3_KSPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert lib.wlk_diag_env_refresh() == 0
        c = np.full((M, N), np.nan, np.float32)
        assert lib.wlk_diag_linear_x3(vp(a), vp(w), vp(bias), M, N, K, flags, 0.5, N // 2, vp(c)) == 0, lib.wlk_diag_last_error()
        out[name] = c
    for k in ("WLK_X3_BM", "WLK_X3_KSPLIT"):
        monkeypatch.delenv(k, raising=False)
    assert lib.wlk_diag_env_refresh() == 0
    assert np.isfinite(out["bm96"]).all()
    assert np.array_equal(out["bm96"].view(np.uint32), out["bm64"].view(np.uint32)), float(np.abs(out["bm96"] - out["bm64"]).max())
    scale = float(np.abs(out["one_wave"]).max())
    assert float(np.abs(out["bm96"] - out["one_wave"]).max()) <= 4e-6 * max(1.0, scale)
