// Where does the fp32 MFMA GEMM lose its cycles?  Same 64x64x32 LDS-tiled inner loop as
// gemm_nt_f32_kernel, with pieces switched off:
//   V0: MFMAs only (operands in registers)            V1: + LDS fragment reads
//   V2: + LDS stores of a register tile + barrier      V3: + global loads (full kernel body)
// Reports TFLOP/s for a 4096x4096x1024-shaped launch (16384 workgroups) and a 192-workgroup launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 36;
template <int V>
__global__ __launch_bounds__(256) void probe(const float* A, const float* W, float* C, int K, int lda) {
    __shared__ __attribute__((aligned(16))) float As[2][64 * LD];
    __shared__ __attribute__((aligned(16))) float Ws[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int row = tid >> 3, c4 = tid & 7;
    for (int i = tid; i < 64 * LD; i += 256) { As[0][i] = As[1][i] = 0.001f * i; Ws[0][i] = Ws[1][i] = 0.002f * i; }
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int a_off = (wr * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const int w_off = (wc * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* ap = A + (long)(blockIdx.y * 64 + row) * lda + c4 * 4;
    const float* wp = W + (long)(blockIdx.x * 64 + row) * lda + c4 * 4;
    float4 ra0 = make_float4(1, 2, 3, 4), ra1 = ra0, rw0 = ra0, rw1 = ra0;
    float4 a4[4], b4[4];
    for (int s = 0; s < 4; ++s) { a4[s] = make_float4(lane, 1, 2, 3); b4[s] = make_float4(1, lane, 2, 3); }
    int buf = 0;
    for (int kt = 0; kt < K / 32; ++kt) {
        if (V >= 3) {
            ra0 = *reinterpret_cast<const float4*>(ap + kt * 32);
            ra1 = *reinterpret_cast<const float4*>(ap + kt * 32 + 32 * lda);
            rw0 = *reinterpret_cast<const float4*>(wp + kt * 32);
            rw1 = *reinterpret_cast<const float4*>(wp + kt * 32 + 32 * lda);
        }
        if (V >= 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a4[s] = *reinterpret_cast<const float4*>(&As[buf][a_off + s * 8]);
                b4[s] = *reinterpret_cast<const float4*>(&Ws[buf][w_off + s * 8]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].x, b4[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].y, b4[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].z, b4[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].w, b4[s].w, acc, 0, 0, 0);
        }
        if (V >= 2) {
            *reinterpret_cast<float4*>(&As[buf ^ 1][row * LD + c4 * 4]) = ra0;
            *reinterpret_cast<float4*>(&As[buf ^ 1][(row + 32) * LD + c4 * 4]) = ra1;
            *reinterpret_cast<float4*>(&Ws[buf ^ 1][row * LD + c4 * 4]) = rw0;
            *reinterpret_cast<float4*>(&Ws[buf ^ 1][(row + 32) * LD + c4 * 4]) = rw1;
            __syncthreads();
            buf ^= 1;
        }
    }
    float sacc = 0.f;
    for (int i = 0; i < 16; ++i) sacc += acc[i];
    C[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = sacc;
}
template <int V>
void run(const char* name, int gx, int gy, int K, float* A, float* W, float* C) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<V>, dim3(gx, gy), dim3(256), 0, 0, A, W, C, K, K);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<V>, dim3(gx, gy), dim3(256), 0, 0, A, W, C, K, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 64 * 64 * K * (double)gx * gy * reps;
    printf("%-28s grid %4dx%-3d K=%4d  %8.1f us/launch  %6.1f TFLOP/s\n", name, gx, gy, K, 1e3 * ms / reps, fl / (ms * 1e-3) / 1e12);
}
int main() {
    float *A, *W, *C;
    hipMalloc(&A, 4096L * 4096 * 4); hipMalloc(&W, 4096L * 4096 * 4); hipMalloc(&C, 64L * 64 * 256 * 4 * 4);
    hipMemset(A, 0, 4096L * 4096 * 4); hipMemset(W, 0, 4096L * 4096 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const int gx = pass ? 8 : 64, gy = pass ? 24 : 64, K = pass ? 2048 : 1024;
        run<0>("V0 mfma only", gx, gy, K, A, W, C);
        run<1>("V1 + lds reads", gx, gy, K, A, W, C);
        run<2>("V2 + lds writes + barrier", gx, gy, K, A, W, C);
        run<3>("V3 + global loads", gx, gy, K, A, W, C);
    }
    return 0;
}
