// Wavefront (64-lane) butterfly exchanges without the LDS crossbar.  gfx950 only.
//
// `v += __shfl_xor(v, off, 64)` compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): ~100 cycles per level, six
// dependent levels per reduction, and the decode-step kernels are nothing but a memory round trip followed by two or
// three such reductions.  The same exchange pattern exists in the VALU:
//     off = 32 : v_permlane32_swap_b32   (gfx950)        off = 8 : DPP row_ror:8
//     off = 16 : v_permlane16_swap_b32   (gfx950)        off = 4 : DPP row_half_mirror, then quad_perm [3,2,1,0]
//     off = 2  : DPP quad_perm [2,3,0,1]                 off = 1 : DPP quad_perm [1,0,3,2]
// Every level still combines lane i with lane i ^ off, and the levels run in the same order (32, 16, 8, 4, 2, 1),
// so a sum / max folded here is bit-identical to the __shfl_xor loop it replaces (a + b == b + a in IEEE arithmetic;
// only the operand order inside a level changes).  Checked lane by lane on the GPU by wlk_diag_wave_ops
// (tests/test_gpu_parity.py).
//
// The two swap instructions are written as inline asm: with both operands holding the same value, the builtins
// (__builtin_amdgcn_permlane{16,32}_swap) of ROCm 7.2's hipcc fold the two results into one register.
#pragma once
#include <hip/hip_runtime.h>

namespace wlk {

// A pointer that was itself loaded from memory (per-session tables) has no known address space: hipcc emits flat_load,
// which cannot be counted by vmcnt(N) and waits for everything at once.  Device buffers are global memory: kernels keep
// such pointers in the global address space explicitly (a cast back to a generic pointer would lose it again).
typedef const __attribute__((address_space(1))) float* gcf_ptr;
typedef float wlk_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) wlk_v4f* gcf4_ptr;
__device__ __forceinline__ gcf_ptr to_global(const float* p) { return (gcf_ptr)p; }
__device__ __forceinline__ float4 ldg4(gcf_ptr p) {
    const wlk_v4f t = *(gcf4_ptr)p;
    return make_float4(t.x, t.y, t.z, t.w);
}

// "This value exists here": an empty asm that reads and rewrites the registers.  hipcc sinks a load whose only uses sit
// inside a branch INTO that branch (a dependent memory round trip behind whatever the kernel computes first) and hoists
// single uses between the loads of a block (a wait in the middle of the issue stream); a pin after the last load of the
// block, followed by __builtin_amdgcn_sched_barrier(0), keeps the whole block in front of the first wait.
__device__ __forceinline__ void pin_loaded(float4& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
__device__ __forceinline__ void pin_loaded(float& a) { asm volatile("" : "+v"(a)); }

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// a = [lo | lo], b = [hi | hi] of v's two 32-lane halves
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// rows of 16 lanes: a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// value of lane (i ^ OFF), any OFF in {1, 2, 4, 8, 16, 32}
template <int OFF>
__device__ __forceinline__ float wave_xor(float v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "butterfly offsets only");
    if constexpr (OFF == 32) {
        float a, b;
        swap32(v, a, b);
        return (__lane_id() & 32) ? a : b;
    } else if constexpr (OFF == 16) {
        float a, b;
        swap16(v, a, b);
        return (__lane_id() & 16) ? a : b;
    } else if constexpr (OFF == 8) {
        return dpp_f32<0x128>(v);
    } else if constexpr (OFF == 4) {
        return dpp_f32<0x1B>(dpp_f32<0x141>(v));
    } else if constexpr (OFF == 2) {
        return dpp_f32<0x4E>(v);
    } else {
        return dpp_f32<0xB1>(v);
    }
}
template <int OFF>
__device__ __forceinline__ int wave_xor(int v) {
    return __builtin_bit_cast(int, wave_xor<OFF>(__builtin_bit_cast(float, v)));
}

// == for (off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);   every lane ends with the total
__device__ __forceinline__ float wave_sum(float v) {
    float a, b;
    swap32(v, a, b);
    v = a + b;
    swap16(v, a, b);
    v = a + b;
    v += dpp_f32<0x128>(v);
    v += dpp_f32<0x1B>(dpp_f32<0x141>(v));
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0xB1>(v);
    return v;
}

// == for (off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
__device__ __forceinline__ float wave_max(float v) {
    float a, b;
    swap32(v, a, b);
    v = fmaxf(a, b);
    swap16(v, a, b);
    v = fmaxf(a, b);
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, dpp_f32<0x1B>(dpp_f32<0x141>(v)));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0xB1>(v));
    return v;
}

// the four low levels in ASCENDING order, == v += __shfl_xor(v, 1); ... 2; ... 4; ... 8: totals of each 16-lane row
// (the attention kernels fold their 16-lane dot products in this order; the order is part of the result's rounding)
__device__ __forceinline__ float row16_sum_1248(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x1B>(dpp_f32<0x141>(v));
    v += dpp_f32<0x128>(v);
    return v;
}

// (value, index) arg-max with ties to the lowest index, == the __shfl_xor loop of select.hip
__device__ __forceinline__ void wave_argmax(float& bv, int& bi) {
#define WLK_ARGMAX_LEVEL(OFF)                                              \
    {                                                                      \
        const float ov = wave_xor<OFF>(bv);                                \
        const int oi = wave_xor<OFF>(bi);                                  \
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }        \
    }
    WLK_ARGMAX_LEVEL(32)
    WLK_ARGMAX_LEVEL(16)
    WLK_ARGMAX_LEVEL(8)
    WLK_ARGMAX_LEVEL(4)
    WLK_ARGMAX_LEVEL(2)
    WLK_ARGMAX_LEVEL(1)
#undef WLK_ARGMAX_LEVEL
}

}  // namespace wlk
