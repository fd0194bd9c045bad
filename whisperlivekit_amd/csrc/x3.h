// "X3": an fp32 tensor held as three bf16 planes, x = hi + mid + lo EXACTLY (each plane is the round-to-nearest bf16 of
// what the planes before it left over; bf16 has fp32's exponent range, so nothing under- or overflows on the way).
//
// Why: gfx950's matrix cores run bf16 at 16x the fp32-input rate (v_mfma_f32_32x32x16_bf16: 32768 flop per 32 cycles per
// SIMD against v_mfma_f32_32x32x2_f32's 4096 per 64), and an fp32 product is six bf16 products away:
//     a b = (ah + am + al)(bh + bm + bl) ~ ah bh + ah bm + am bh + ah bl + al bh + am bm
// The three dropped terms are below 2^-25 |a b| (|am| <= 2^-9 |a|, |al| <= 2^-18 |a|), under fp32's own half-ulp; every
// bf16 product is exact in the fp32 accumulator.  Six MFMAs at 16x the rate = 2.67x the fp32 matrix peak, at fp32
// accuracy: measured against float64, the six-term GEMM's error is that of the fp32 MFMA GEMM it replaces (and of the
// reference's own fp32 CPU GEMM) - tests/test_gpu_x3.py.  Round 4: the encoder's GEMMs (69 of an encode's 87 GFLOP).
//
// Layout: the planes are interleaved per 8-element chunk so that one K-slab of one row is ONE contiguous span:
//     element (row, k), plane p  ->  16-byte unit ((k / 8) * 3 + p) of the row, position k % 8 inside it
// A row of `ld` fp32 elements is 3 * ld bf16 = 6 * ld bytes.  A 16-byte unit is exactly one lane's A / B fragment of
// v_mfma_f32_32x32x16_bf16 (8 consecutive k of one row), so fragments are single ds_read_b128s of the LDS image.
#pragma once
#include <hip/hip_runtime.h>

namespace wlk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));

struct X3Triple {
    __bf16 h, m, l;
};
__device__ __forceinline__ X3Triple x3_split(float x) {
    X3Triple t;
    t.h = (__bf16)x;
    const float r = x - (float)t.h;
    t.m = (__bf16)r;
    const float r2 = r - (float)t.m;
    t.l = (__bf16)r2;
    return t;
}
__device__ __forceinline__ unsigned x3_bits(__bf16 v) { return (unsigned)__builtin_bit_cast(unsigned short, v); }

// eight consecutive fp32 values -> the chunk's three 16-byte units (hi | mid | lo), 48 contiguous bytes at dst
__device__ __forceinline__ void x3_store_chunk(unsigned short* dst, const float (&v)[8]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const X3Triple a = x3_split(v[2 * i]), b = x3_split(v[2 * i + 1]);
        h[i] = x3_bits(a.h) | (x3_bits(b.h) << 16);
        m[i] = x3_bits(a.m) | (x3_bits(b.m) << 16);
        l[i] = x3_bits(a.l) | (x3_bits(b.l) << 16);
    }
    x3_u32x4* d = reinterpret_cast<x3_u32x4*>(dst);
    d[0] = x3_u32x4{h[0], h[1], h[2], h[3]};
    d[1] = x3_u32x4{m[0], m[1], m[2], m[3]};
    d[2] = x3_u32x4{l[0], l[1], l[2], l[3]};
}

}  // namespace wlk
