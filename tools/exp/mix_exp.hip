// mix_exp.hip — does the vector ALU run packed fp32 FMAs in the shadow of fp32 MFMAs on gfx950?  (NOT product code.)
// Every wave issues groups of one v_mfma_f32_16x16x4_f32 followed by NV independent v_pk_fma_f32 (register operands only, no
// memory), 2 waves per SIMD on every CU.  Reported: matrix TFLOP/s, vector TFLOP/s and their sum for NV = 0..8 and for the
// vector-only loop.  If the two pipes overlap, the matrix rate stays near its NV = 0 value while the vector rate grows.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mix_exp mix_exp.hip      Run: ./mix_exp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// matrix-only loops: which fp32 MFMA shape sustains the higher rate, and how many waves per SIMD does it take?
template <int SHAPE, int WPS>      // SHAPE 0: 16x16x4 (8 passes), 1: 32x32x2 (16 passes), 2: 4x4x1 x16 blocks? (not used)
__global__ void __launch_bounds__(256, WPS) k_mat(float* __restrict__ out, int iters, float seed) {
    const int tid = threadIdx.x;
    const float a = seed * (tid & 15), b = 1.0f + seed;
    float s = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int SHAPE, int WPS>
static void run_mat(float* out, int iters) {
    const int blocks = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mat<SHAPE, WPS>), dim3(blocks), dim3(256), 0, 0, out, 64, 1e-3f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mat<SHAPE, WPS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double per_it = SHAPE == 0 ? 8 * 2.0 * 16 * 16 * 4 : 4 * 2.0 * 32 * 32 * 2;
    printf("%s, %d waves/SIMD: %8.3f ms   %7.2f TF\n", SHAPE == 0 ? "16x16x4" : "32x32x2", WPS, ms, (double)blocks * 4 * iters * per_it / ms / 1e9);
    fflush(stdout);
}

#define NACC 8        // independent MFMA accumulators (a dependent chain would stall on the 8-pass latency)
#define NVEC 16       // independent packed accumulators

template <int NV, bool MFMA>
__global__ void __launch_bounds__(256, 2) k_mix(float* __restrict__ out, int iters, float seed) {
    const int tid = threadIdx.x;
    f32x4 acc[NACC];
    f32x2 v[NVEC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NVEC; ++j) v[j] = f32x2{seed * j, seed};
    const float a = seed * (tid & 15), b = 1.0f + seed;
    const f32x2 x = {1.0f + seed, 1.0f - seed}, y = {seed, -seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (MFMA) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int r = (j * NV + t) % NVEC;
                v[r] = __builtin_elementwise_fma(v[r], x, y);
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the groups as written
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) s += v[j][0] + v[j][1];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int NV, bool MFMA>
static void run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<NV, MFMA>), dim3(blocks), dim3(256), 0, 0, out, 64, 1e-3f);        // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mix<NV, MFMA>), dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4, groups = waves * iters * NACC;
    const double mf = MFMA ? groups * 2.0 * 16 * 16 * 4 : 0.0, vf = groups * NV * 2.0 * 2 * 64;
    printf("NV=%d mfma=%d: %8.3f ms   matrix %7.2f TF   vector %7.2f TF   sum %7.2f TF\n", NV, (int)MFMA, ms, mf / ms / 1e9, vf / ms / 1e9,
           (mf + vf) / ms / 1e9);
    fflush(stdout);
}

// workgroup turnover: how long does a CU slot stay empty between two workgroups of a launch?  Each workgroup spins for `spin`
// matrix instructions per wave and exits; LDS per workgroup sets the number of slots per CU.
__global__ void __launch_bounds__(256) k_turn(float* __restrict__ out, int spin, float seed) {
    extern __shared__ float lds[];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = seed * threadIdx.x, b = 1.0f + seed;
    for (int i = 0; i < spin; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) { lds[threadIdx.x] = acc[1]; out[blockIdx.x] = lds[threadIdx.x ^ 1]; }
}
static void run_turn(float* out, int lds_kb, int spin) {
    const int slots = 160 / lds_kb, nwg = 256 * slots * 64;
    hipFuncSetAttribute((const void*)k_turn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_turn, dim3(nwg), dim3(256), lds_kb * 1024, 0, out, spin, 1e-3f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_turn, dim3(nwg), dim3(256), lds_kb * 1024, 0, out, spin, 1e-3f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    // a dependent chain of `spin` 8-pass MFMAs takes >= spin * 32 cycles (~2.4 GHz); 64 rounds of workgroups per slot
    const double ideal_us = 64.0 * spin * 40.0 / 2400.0;       // 40 cycles per dependent 16x16x4 (measured single-wave rate)
    printf("turnover: LDS %3d KB (%d slots/CU), spin %5d: %8.1f us for 64 rounds, %6.2f us per round, compute alone ~%6.2f us per round\n",
           lds_kb, slots, spin, ms * 1e3, ms * 1e3 / 64, ideal_us / 64);
    fflush(stdout);
}

int main() {
    const int blocks = 256 * 2, iters = 20000;       // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    float* out; if (hipMalloc(&out, (size_t)1024 * 256 * sizeof(float)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    if (getenv("MIX_TURNOVER")) {
        for (int kb : {64, 48, 32}) for (int spin : {0, 256, 1024, 4096}) run_turn(out, kb, spin);
        hipFree(out); return 0;
    }
    run<0, true>(out, blocks, iters);
    run<1, true>(out, blocks, iters);
    run<2, true>(out, blocks, iters);
    run<3, true>(out, blocks, iters);
    run<4, true>(out, blocks, iters);
    run<5, true>(out, blocks, iters);
    run<6, true>(out, blocks, iters);
    run<7, true>(out, blocks, iters);
    run<8, true>(out, blocks, iters);
    run<4, false>(out, blocks, iters);
    run<8, false>(out, blocks, iters);
    run_mat<0, 1>(out, 40000); run_mat<0, 2>(out, 40000); run_mat<0, 3>(out, 40000); run_mat<0, 4>(out, 40000);
    run_mat<1, 1>(out, 40000); run_mat<1, 2>(out, 40000); run_mat<1, 3>(out, 40000); run_mat<1, 4>(out, 40000);
    run_mat<0, 2>(out, 400000); run_mat<1, 2>(out, 400000);        // ~100 ms each: sustained clocks
    hipFree(out);
    return 0;
}
