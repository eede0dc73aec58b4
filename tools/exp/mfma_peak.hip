// bare MFMA loops on registers: what the matrix pipes of this chip sustain (no LDS, no memory), for pricing the f16 kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int NACC>
__global__ void __launch_bounds__(512) k_peak(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i)); }
    float s = 0.f;
    if constexpr (KIND == 3) {                                      // random operands, eight different fragments of each side: 4 x 8... (toggling data: power)
        constexpr int NA = NACC / 4;
        h8 ar[NA], br[4];
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (_Float16)(((int)(x >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f); };
        for (int i = 0; i < NA; ++i) for (int e = 0; e < 8; ++e) ar[i][e] = rnd();
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) br[j][e] = rnd();
        f32x4 acc[NA][4];
        for (int i = 0; i < NA; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(br[j], ar[i], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < NA; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    } else if constexpr (KIND == 0) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0];
    } else if constexpr (KIND == 1) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float fa = 0.001f * threadIdx.x, fb = 0.002f * threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NACC>
static void run(const char* name, double flop_per_mfma, int threads, int blocks_per_cu) {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    const int iters = 40000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_peak<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_peak<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64, fl = waves * iters * NACC * flop_per_mfma;
    printf("%-34s waves/SIMD %.0f  %8.2f ms  %8.1f TFLOP/s\n", name, waves / 1024.0, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0, 16>("f16 16x16x32, 16 acc, 1 wave/SIMD", 16384.0, 256, 1);
    run<0, 16>("f16 16x16x32, 16 acc, 2 waves/SIMD", 16384.0, 512, 1);
    run<0, 32>("f16 16x16x32, 32 acc, 2 waves/SIMD", 16384.0, 512, 1);
    run<3, 32>("f16 16x16x32 RANDOM, 8x4, 2 w/SIMD", 16384.0, 512, 1);
    run<3, 32>("f16 16x16x32 RANDOM, 8x4, 2 w/SIMD", 16384.0, 512, 1);
    run<3, 32>("f16 16x16x32 RANDOM, 8x4, 1 w/SIMD", 16384.0, 256, 1);
    run<1, 4>("f16 32x32x16, 4 acc, 1 wave/SIMD", 32768.0, 256, 1);
    run<1, 8>("f16 32x32x16, 8 acc, 2 waves/SIMD", 32768.0, 512, 1);
    run<2, 4>("f32 32x32x2, 4 acc, 2 waves/SIMD", 4096.0, 512, 1);
    run<2, 4>("f32 32x32x2, 4 acc, 4 waves/SIMD", 4096.0, 512, 2);
    return 0;
}
