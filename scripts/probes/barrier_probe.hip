// Probe for a persistent decode-step kernel: cost and correctness of a fence-free grid barrier on gfx950.
// Protocol under test (cdna_hip_programming.md G16 "write-through sc1 payload + drain, no fence"):
//   producer: agent-scope relaxed atomic stores (global_store sc1) of the payload into a FRESH address range,
//             s_waitcnt vmcnt(0), __syncthreads, lane 0: relaxed agent atomicAdd on the barrier word
//   consumers: lane 0 polls the word with relaxed agent loads (bounded), __syncthreads, PLAIN loads of the payload
// Every phase uses a payload region no workgroup has touched before in this launch, so no L1/L2 can hold a stale line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int VEC = 512;

__device__ __forceinline__ bool grid_barrier(unsigned* word, unsigned target, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
    return true;
}

// mode 0: sc1 (atomic relaxed agent) payload stores; mode 1: plain payload stores (expected to be stale sometimes)
__global__ __launch_bounds__(256) void probe(float* payload, unsigned* word, int phases, int mode, int* bad, int* err,
                                             float* sink) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* out = payload + (size_t)(p + 1) * VEC;
        if (b == p % G) {   // this phase's producer
            for (int i = tid; i < VEC; i += 256) {
                const float v = (float)(p + 1) + 0.001f * i;
                if (mode == 0) __hip_atomic_store(out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else out[i] = v;
            }
        }
        grid_barrier(word, (unsigned)(p + 1) * G, err);
        // everyone consumes with plain loads
        int wrong = 0;
        for (int i = tid; i < VEC; i += 256) {
            const float v = out[i];
            acc += v;
            if (v != (float)(p + 1) + 0.001f * i) ++wrong;
        }
        if (wrong) atomicAdd(bad, wrong);
    }
    if (acc == -1.f) sink[0] = acc;
}

// mode 2: the hand-off stays inside ONE XCD.  256 workgroups are launched (32 per XCD, observed block b -> XCD b % 8);
// only those whose HW_REG_XCC_ID is 0 take part (the others report and leave), so every hand-off is served by that
// XCD's shared L2: plain payload stores (L1 is write-through), s_waitcnt vmcnt(0), relaxed agent atomics for the
// barrier, plain loads of a fresh address range.  Placement is used for SPEED only: membership is counted at run time.
__global__ __launch_bounds__(256) void probe_one_xcd(float* payload, unsigned* words, int phases, int* bad, int* err,
                                                      float* sink) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    __shared__ unsigned members_s;
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (xcc == 0) __hip_atomic_fetch_add(words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(words + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (xcc != 0) return;
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
        members_s = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned members = members_s;
    // rank among the members: take a ticket
    __shared__ unsigned rank_s;
    if (tid == 0) rank_s = __hip_atomic_fetch_add(words + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned rank = rank_s;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* out = payload + (size_t)(p + 1) * VEC;
        if (rank == (unsigned)p % members)
            for (int i = tid; i < VEC; i += 256) out[i] = (float)(p + 1) + 0.001f * i;
        grid_barrier(words, (unsigned)(p + 1) * members, err);
        int wrong = 0;
        for (int i = tid; i < VEC; i += 256) {
            const float v = out[i];
            acc += v;
            if (v != (float)(p + 1) + 0.001f * i) ++wrong;
        }
        if (wrong) atomicAdd(bad, wrong);
    }
    if (acc == -1.f) sink[0] = acc;
    if (rank == 0 && tid == 0) sink[1] = (float)members;
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 400;
    float *payload, *sink; unsigned* word; int *bad, *err;
    CK(hipMalloc(&payload, (size_t)(phases + 2) * VEC * sizeof(float)));
    CK(hipMalloc(&sink, 8)); CK(hipMalloc(&word, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {32, 64, 128, 256, 512}) {
            float best = 1e9f; int hbad = 0, herr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(payload, 0, (size_t)(phases + 2) * VEC * sizeof(float)));
                CK(hipMemset(word, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(err, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, payload, word, phases, mode, bad, err, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                int b2, r2; CK(hipMemcpy(&b2, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&r2, err, 4, hipMemcpyDeviceToHost));
                hbad += b2; herr += r2;
            }
            printf("mode %d (%s payload) G=%3d: %.2f us per phase (barrier + 2 KB hand-off), stale values %d, timeouts %d\n",
                   mode, mode == 0 ? "sc1" : "plain", G, 1e3f * best / phases, hbad, herr);
        }
    {
        unsigned* words; CK(hipMalloc(&words, 16));
        float best = 1e9f; int hbad = 0, herr = 0; float members = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(payload, 0, (size_t)(phases + 2) * VEC * sizeof(float)));
            CK(hipMemset(words, 0, 16)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(sink, 0, 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe_one_xcd, dim3(256), dim3(256), 0, 0, payload, words, phases, bad, err, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            int b2, r2; CK(hipMemcpy(&b2, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&r2, err, 4, hipMemcpyDeviceToHost));
            float sk[2]; CK(hipMemcpy(sk, sink, 8, hipMemcpyDeviceToHost)); members = sk[1];
            hbad += b2; herr += r2;
        }
        printf("mode 2 (one XCD, plain payload through its L2) members=%.0f: %.2f us per phase, stale values %d, timeouts %d\n",
               members, 1e3f * best / phases, hbad, herr);
    }
    return 0;
}
