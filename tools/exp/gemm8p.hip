// gemm8p.hip — experiment (NOT product code): the 256x256x64 fp16 GEMM main loop in the "8-phase" form of cdna_hip_programming.md §5
// (half-tile LDS slots recycled one at a time, one LDS-DMA half-tile per phase, counted vmcnt once per K tile, two barriers per phase, the
// two wave groups of a SIMD one barrier apart so that one multiplies while the other reads), against the shipped 2-phase loop of
// var_amd/csrc/gemm16.hip (k_gemm16p) called through the C ABI in the same process.
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm8p gemm8p.hip -ldl      Run: ./gemm8p [path/to/libvar_hip.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ROWB 128
#define HALFT (128 * ROWB)          // one half-tile: 128 rows x 64 halves = 16 KB

__device__ __forceinline__ void dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void bar() { __builtin_amdgcn_s_barrier(); }

// FLAGS: 1 = no global stores (K loop only), 2 = no stagger between the wave groups, 4 = no setprio
template <int FLAGS>
__global__ void __launch_bounds__(512, 2) k8p(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C,
                                               int M, int N, int K, int tilesM, int tilesN) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int tm_, tn_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 8, width = GM * tilesN, group = lin / width, first = group * GM;
        const int gsz = (tilesM - first) < GM ? (tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz; tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * 256, n0 = tn_ * 256;
    const int drow = lane >> 3, dslot = lane & 7, r16 = lane & 15, kq = lane >> 4;
    // staging: half-tile h of an operand = rows 128 h .. + 127; wave w stages rows 16 w .. + 15 of it (two 1 KiB pieces)
    uint32_t aoff[2][2], boff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 128 * h + (wave * 2 + i) * 8 + drow;
            aoff[h][i] = (uint32_t)((int64_t)(m0 + r) * K * 2 + ((dslot ^ drow) << 4));
            boff[h][i] = (uint32_t)((int64_t)(n0 + r) * K * 2 + ((dslot ^ drow) << 4));
        }
    auto slot = [&](int buf, int op, int h) -> char* { return sm + ((buf * 2 + op) * 2 + h) * HALFT; };
    auto stage = [&](int op, int h, int kt, int buf) {
        char* d = slot(buf, op, h) + wave * 2048;
        const char* base = (const char*)(op ? W : A) + (size_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma16(base, op ? boff[h][i] : aoff[h][i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(d + i * 1024));
    };
    const int nk = K / 64;
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h8 afr[4][2], bfr[2][2][2];                    // A fragments of one half (block, k-step); W fragments of both halves (half, block, k-step)
    const int sl0 = ((0 + kq) ^ (r16 & 7)) << 4, sl1 = ((4 + kq) ^ (r16 & 7)) << 4;
    auto read_a = [&](int buf, int h) {
        const char* s = slot(buf, 0, h) + (wr * 64 + r16) * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) { afr[i][0] = *(const h8*)(s + i * 16 * ROWB + sl0); afr[i][1] = *(const h8*)(s + i * 16 * ROWB + sl1); }
    };
    auto read_b = [&](int buf, int g) {
        const char* s = slot(buf, 1, g) + (wc * 32 + r16) * ROWB;
#pragma unroll
        for (int j = 0; j < 2; ++j) { bfr[g][j][0] = *(const h8*)(s + j * 16 * ROWB + sl0); bfr[g][j][1] = *(const h8*)(s + j * 16 * ROWB + sl1); }
    };
    auto mma = [&](int h, int g) {
        if (!(FLAGS & 4)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][g][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[g][j][s], afr[i][s], acc[h][g][i][j], 0, 0, 0);
        if (!(FLAGS & 4)) __builtin_amdgcn_s_setprio(0);
    };
    // prologue: all of K tile 0, and A0 / W0 / W1 of K tile 1 (its A1 goes out in phase 1 of tile 0, as in the steady state)
    stage(0, 0, 0, 0); stage(1, 0, 0, 0); stage(1, 1, 0, 0); stage(0, 1, 0, 0);
    if (nk > 1) { stage(0, 0, 1, 1); stage(1, 0, 1, 1); stage(1, 1, 1, 1); wait_vm<6>(); } else wait_vm<0>();
    bar();
    if (!(FLAGS & 2) && wr == 1) bar();            // the second wave group runs one barrier behind the first
#pragma unroll 1
    for (int t = 0; t < nk; ++t) {
        const int b = t & 1;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        // ---- phase 1: quadrant (A0, W0)
        read_b(b, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(b, 0);
        if (more1) stage(0, 1, t + 1, b ^ 1);      // A1 of tile t+1 (its slot was read last in phase 3 of tile t-1)
        wait_lgkm0(); bar();
        mma(0, 0);
        bar();
        // ---- phase 2: quadrant (A0, W1)
        read_b(b, 1);
        if (more2) stage(0, 0, t + 2, b);          // A0 of tile t+2 into the slot phase 1 read
        wait_lgkm0(); bar();
        mma(0, 1);
        bar();
        // ---- phase 3: quadrant (A1, W1)
        read_a(b, 1);
        if (more2) stage(1, 0, t + 2, b);          // W0 of tile t+2 (read in phase 1; its fragments stay in registers for phase 4)
        wait_lgkm0(); bar();
        mma(1, 1);
        bar();
        // ---- phase 4: quadrant (A1, W0); everything of tile t+1 has landed except what went out in phases 2 and 3
        if (more2) wait_vm<4>(); else wait_vm<0>();
        if (more2) stage(1, 1, t + 2, b);          // W1 of tile t+2 (read in phase 2)
        bar();
        mma(1, 0);
        bar();
    }
    if (!(FLAGS & 2) && wr == 0) bar();
    // ---- epilogue (experiment: plain stores from the accumulator layout, 8 bytes per lane)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = acc[h][g][i][j];
                    if (FLAGS & 1) { asm volatile("" :: "v"(v)); continue; }
                    const int m = m0 + 128 * h + wr * 64 + i * 16 + r16, n = n0 + 128 * g + wc * 32 + j * 16 + 4 * kq;
                    h4 o; o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
                    *(h4*)(C + (int64_t)m * N + n) = o;
                }
}

typedef int (*gemm_fn)(const void*, int64_t, const void*, int64_t, const float*, void*, int64_t, int, int, int, int, int,
                       const void*, int64_t, int, const float*, int64_t, int, int, int64_t, int64_t, int64_t, void*);

template <int FLAGS> static float run8p(const _Float16* A, const _Float16* W, _Float16* C, int M, int N, int K, int iters) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k8p<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * HALFT); attr = true; }
    const int tM = M / 256, tN = N / 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k8p<FLAGS>, dim3(tM * tN), dim3(512), 8 * HALFT, 0, A, W, C, M, N, K, tM, tN);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k8p<FLAGS>, dim3(tM * tN), dim3(512), 8 * HALFT, 0, A, W, C, M, N, K, tM, tN);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
    return ms / iters;
}

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "var_amd/libvar_hip.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    gemm_fn lib_gemm = lib ? (gemm_fn)dlsym(lib, "varhip_gemm_nt_f16") : nullptr;
    if (!lib_gemm) printf("(library not loaded: %s)\n", dlerror());
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {32768, 3072, 1024}, {32768, 4096, 1024}, {32768, 1024, 4096}, {32768, 1024, 1024}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<_Float16> hA((size_t)M * K), hW((size_t)N * K);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };      // uniform [-1, 1)
        for (auto& v : hA) v = (_Float16)rnd();
        for (auto& v : hW) v = (_Float16)(rnd() * 0.05f);
        _Float16 *A, *W, *C, *C2;
        hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
        hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        hipMemset(C, 0, (size_t)M * N * 2);
        const int iters = 20;
        const double fl = 2.0 * M * N * (double)K;
        const float t0 = run8p<0>(A, W, C, M, N, K, iters);
        // correctness on 256 sampled elements
        std::vector<_Float16> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int q = 0; q < 256; ++q) {
            const int m = (int)(((uint64_t)q * 2654435761u) % M), n = (int)(((uint64_t)q * 40503u + 17) % N);
            double r = 0; for (int k = 0; k < K; ++k) r += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
            const double e = fabs((double)hC[(size_t)m * N + n] - r) / (fabs(r) + 1.0);
            if (e > maxerr) maxerr = e;
        }
        const float t1 = run8p<1>(A, W, C, M, N, K, iters), t2 = run8p<3>(A, W, C, M, N, K, iters), t3 = run8p<5>(A, W, C, M, N, K, iters);
        float tl = 0;
        if (lib_gemm) {
            for (int i = 0; i < 3; ++i) lib_gemm(A, K, W, K, nullptr, C2, N, 1, M, N, K, 0, nullptr, 0, 0, nullptr, 0, 1, 1, 0, 0, 0, nullptr);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
            for (int i = 0; i < iters; ++i) lib_gemm(A, K, W, K, nullptr, C2, N, 1, M, N, K, 0, nullptr, 0, 0, nullptr, 0, 1, 1, 0, 0, 0, nullptr);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&tl, a, b); tl /= iters;
        }
        printf("M=%6d N=%5d K=%5d  8-phase %8.1f us %7.1f TF | no stores %8.1f us %7.1f TF | no stores, no stagger %8.1f us | no stores, no setprio %8.1f us | library %8.1f us %7.1f TF | max rel err %.2e\n",
               M, N, K, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, t3 * 1e3, tl * 1e3, tl > 0 ? fl / tl / 1e9 : 0.0, maxerr);
        fflush(stdout);
        hipFree(A); hipFree(W); hipFree(C); hipFree(C2);
    }
    return 0;
}
