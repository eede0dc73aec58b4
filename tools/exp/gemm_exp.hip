// gemm_exp.hip — standalone timing experiments on the fp32 MFMA GEMM main loop (NOT product code; results may be numerically
// wrong when a stage is ablated).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gemm_exp gemm_exp.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FLAGS: 1 = skip global loads after the first tile, 2 = skip LDS stores after the first, 4 = skip barrier, 8 = skip LDS reads (reuse regs)
template <int TM, int TN, int WGM, int WGN, int BK, int FLAGS, int MINW>
__global__ void __launch_bounds__(256, MINW) k(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int M, int N, int K) {
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN, LDSW = BK + 4, CPR = BK / 8;
    constexpr int NA = BM * CPR / 256, NB = BN * CPR / 256, STAGE = (BM + BN) * LDSW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tilesN = N / BN, tilesM = M / BM;
    int tm_, tn_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 8, width = GM * tilesN, group = lin / width, first = group * GM;
        const int gsz = (tilesM - first) < GM ? (tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz; tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const float* a_ptr[NA]; const float* b_ptr[NB]; int a_off[NA], b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { int qq = tid + 256 * i; a_ptr[i] = A + (size_t)(m0 + qq / CPR) * K + (qq % CPR) * 8; a_off[i] = (qq / CPR) * LDSW + (qq % CPR) * 8; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { int qq = tid + 256 * i; b_ptr[i] = W + (size_t)(n0 + qq / CPR) * K + (qq % CPR) * 8; b_off[i] = (qq / CPR) * LDSW + (qq % CPR) * 8; }
    f32x4 ra[NA][2], rb[NB][2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { const float* s = a_ptr[i] + kt * BK; ra[i][0] = *(const f32x4*)s; ra[i][1] = *(const f32x4*)(s + 4); }
#pragma unroll
        for (int i = 0; i < NB; ++i) { const float* s = b_ptr[i] + kt * BK; rb[i][0] = *(const f32x4*)s; rb[i][1] = *(const f32x4*)(s + 4); }
    };
    auto store_tile = [&](int st) {
        float* sA = smem + st * STAGE; float* sB = sA + BM * LDSW;
#pragma unroll
        for (int i = 0; i < NA; ++i) { float* d = sA + a_off[i]; f32x4 ev = {ra[i][0][0], ra[i][0][2], ra[i][1][0], ra[i][1][2]}, od = {ra[i][0][1], ra[i][0][3], ra[i][1][1], ra[i][1][3]}; *(f32x4*)d = ev; *(f32x4*)(d + 4) = od; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { float* d = sB + b_off[i]; f32x4 ev = {rb[i][0][0], rb[i][0][2], rb[i][1][0], rb[i][1][2]}, od = {rb[i][0][1], rb[i][0][3], rb[i][1][1], rb[i][1][3]}; *(f32x4*)d = ev; *(f32x4*)(d + 4) = od; }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / BK;
    if (FLAGS & 16) {           // ---- 2-deep register prefetch: tile kt+2 is in flight while tile kt is computed ----
        f32x4 ra2[NA][2], rb2[NB][2];
        auto load_tile2 = [&](int kt) {
#pragma unroll
            for (int i = 0; i < NA; ++i) { const float* s = a_ptr[i] + kt * BK; ra2[i][0] = *(const f32x4*)s; ra2[i][1] = *(const f32x4*)(s + 4); }
#pragma unroll
            for (int i = 0; i < NB; ++i) { const float* s = b_ptr[i] + kt * BK; rb2[i][0] = *(const f32x4*)s; rb2[i][1] = *(const f32x4*)(s + 4); }
        };
        auto store_tile2 = [&](int st) {
            float* sA = smem + st * STAGE; float* sB = sA + BM * LDSW;
#pragma unroll
            for (int i = 0; i < NA; ++i) { float* d = sA + a_off[i]; f32x4 ev = {ra2[i][0][0], ra2[i][0][2], ra2[i][1][0], ra2[i][1][2]}, od = {ra2[i][0][1], ra2[i][0][3], ra2[i][1][1], ra2[i][1][3]}; *(f32x4*)d = ev; *(f32x4*)(d + 4) = od; }
#pragma unroll
            for (int i = 0; i < NB; ++i) { float* d = sB + b_off[i]; f32x4 ev = {rb2[i][0][0], rb2[i][0][2], rb2[i][1][0], rb2[i][1][2]}, od = {rb2[i][0][1], rb2[i][0][3], rb2[i][1][1], rb2[i][1][3]}; *(f32x4*)d = ev; *(f32x4*)(d + 4) = od; }
        };
        auto compute = [&](int cur, auto&& mid) {
            const float* sA = smem + cur * STAGE + (wm * TM * 32 + r) * LDSW + h * 4;
            const float* sB = smem + cur * STAGE + BM * LDSW + (wn * TN * 32 + r) * LDSW + h * 4;
#pragma unroll
            for (int c = 0; c < CPR; ++c) {
                if (c == (CPR + 1) / 2) mid();
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(sA + i * 32 * LDSW + c * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(sB + j * 32 * LDSW + c * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
        };
        // prologue: tile 0 -> LDS stage 0; tile 1 -> regs set 1 (ra); tile 2 -> regs set 2 (ra2)
        load_tile(0); store_tile(0); __syncthreads();
        load_tile(1); if (nk > 2) load_tile2(2);
        for (int kt = 0; kt < nk; kt += 2) {
            // even tile kt (stage 0): mid-tile, write tile kt+1 (regs set 1) into stage 1, then refill set 1 with tile kt+3
            compute(0, [&] { if (kt + 1 < nk) { store_tile(1); if (kt + 3 < nk) load_tile(kt + 3); } });
            __syncthreads();
            if (kt + 1 >= nk) break;
            // odd tile kt+1 (stage 1): mid-tile, write tile kt+2 (regs set 2) into stage 0, then refill set 2 with tile kt+4
            compute(1, [&] { if (kt + 2 < nk) { store_tile2(0); if (kt + 4 < nk) load_tile2(kt + 4); } });
            __syncthreads();
        }
    } else {
    load_tile(0); store_tile(0); store_tile(1); __syncthreads();
    f32x4 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(smem + (wm * TM * 32 + i * 32 + r) * LDSW + h * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(smem + BM * LDSW + (wn * TN * 32 + j * 32 + r) * LDSW + h * 4);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (!(FLAGS & 1) && kt + 1 < nk) load_tile(kt + 1);
        const float* sA = smem + cur * STAGE + (wm * TM * 32 + r) * LDSW + h * 4;
        const float* sB = smem + cur * STAGE + BM * LDSW + (wn * TN * 32 + r) * LDSW + h * 4;
#pragma unroll
        for (int c = 0; c < CPR; ++c) {
            if (!(FLAGS & 2) && c == (CPR + 1) / 2 && kt + 1 < nk) store_tile(cur ^ 1);
            if (!(FLAGS & 8)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(sA + i * 32 * LDSW + c * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(sB + j * 32 * LDSW + c * 8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (!(FLAGS & 4)) __syncthreads();
    }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                out[(size_t)(m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * N + n0 + (wn * TN + j) * 32 + r] = acc[i][j][e];
}

template <int TM, int TN, int WGM, int WGN, int BK, int FLAGS, int MINW>
void run(const char* name, const float* A, const float* W, float* out, int M, int N, int K) {
    constexpr int BM = TM * 32 * WGM, BN = TN * 32 * WGN;
    size_t lds = 2 * (size_t)(BM + BN) * (BK + 4) * 4;
    auto kf = k<TM, TN, WGM, WGN, BK, FLAGS, MINW>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kf, grid, dim3(256), lds, 0, A, W, out, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; if (ms < best) best = ms;
    }
    printf("%-44s %4dx%-4d BK%-2d flags %2d : %8.3f ms  %6.1f TF  (%s)\n", name, BM, BN, BK, FLAGS, best, 2.0 * M * N * K / best / 1e9, hipGetErrorString(hipGetLastError()));
}


// ---- variant B: LDS-DMA staging (global_load_lds 16 B, XOR-swizzled source), MFMA 16x16x4, ds_read_b32 operands, natural k order ----
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int WT /*wave tile = WT x WT*/, int NSTAGE, int BK, int MINW, int PF>
__global__ void __launch_bounds__(256, MINW) kdma(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int M, int N, int K) {
    constexpr int BM = 2 * WT, BN = 2 * WT, T = WT / 16, CH = BK / 4 /*16B chunks per row*/, RPI = 64 / CH /*rows per DMA instr*/;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [NSTAGE][(BM+BN)*BK]
    constexpr int STAGE = (BM + BN) * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN, tilesM = M / BM;
    int tm_, tn_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 8, width = GM * tilesN, group = lin / width, first = group * GM;
        const int gsz = (tilesM - first) < GM ? (tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz; tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    // DMA roles: wave w stages rows [w*BM/4, (w+1)*BM/4) of A and the same range of B; one instruction = 8 rows x 128 B
    constexpr int NI = BM / 4 / RPI;
    const int drow = lane / CH, dslot = lane % CH;
    const float* asrc[NI]; const float* bsrc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = wave * (BM / 4) + i * RPI + drow;
        asrc[i] = A + (size_t)(m0 + row) * K + ((dslot ^ (row & (CH - 1))) << 2);
        bsrc[i] = W + (size_t)(n0 + row) * K + ((dslot ^ (row & (CH - 1))) << 2);
    }
    auto dma_tile = [&](int kt, int st) {
        float* sA = smem + st * STAGE + wave * (BM / 4) * BK;
        float* sB = smem + st * STAGE + BM * BK + wave * (BN / 4) * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sA + i * RPI * BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sB + i * RPI * BK), 16, 0, 0);
        }
    };
    f32x4v acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = K / BK;
    dma_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % NSTAGE;
        if (kt + 1 < nk) dma_tile(kt + 1, (kt + 1) % NSTAGE);
        const float* sA = smem + cur * STAGE + (wm * WT + r16) * BK + kq;
        const float* sB = smem + cur * STAGE + BM * BK + (wn * WT + r16) * BK + kq;
        if (PF == 0) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            float af[T], bf[T];
            const int sl = ((s ^ (r16 & (CH - 1))) << 2);        // rows i*16 + r16: (row & 7) == (r16 & 7)
#pragma unroll
            for (int i = 0; i < T; ++i) af[i] = sA[i * 16 * BK + sl];
#pragma unroll
            for (int j = 0; j < T; ++j) bf[j] = sB[j * 16 * BK + sl];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        } else {
            // fragment prefetch: the operands of step s+1 are read while the MFMAs of step s run
            float af[2][T], bf[2][T];
            {
                const int sl = ((0 ^ (r16 & (CH - 1))) << 2);
#pragma unroll
                for (int i = 0; i < T; ++i) af[0][i] = sA[i * 16 * BK + sl];
#pragma unroll
                for (int j = 0; j < T; ++j) bf[0][j] = sB[j * 16 * BK + sl];
            }
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                if (s + 1 < CH) {
                    const int sl = (((s + 1) ^ (r16 & (CH - 1))) << 2);
#pragma unroll
                    for (int i = 0; i < T; ++i) af[(s + 1) & 1][i] = sA[i * 16 * BK + sl];
#pragma unroll
                    for (int j = 0; j < T; ++j) bf[(s + 1) & 1][j] = sB[j * 16 * BK + sl];
                }
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
                if (PF == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[(size_t)(m0 + wm * WT + i * 16 + kq * 4 + e) * N + n0 + wn * WT + j * 16 + r16] = acc[i][j][e];
}

template <int WT, int NSTAGE, int BK, int MINW, int PF = 0>
void run_dma(const char* name, const float* A, const float* W, float* out, int M, int N, int K, const float* ref) {
    constexpr int BM = 2 * WT, BN = 2 * WT;
    size_t lds = (size_t)NSTAGE * (BM + BN) * BK * 4;
    auto kf = kdma<WT, NSTAGE, BK, MINW, PF>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kf, grid, dim3(256), lds, 0, A, W, out, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; if (ms < best) best = ms;
    }
    // bit-compare a sample of outputs with the reference kernel's result
    std::vector<float> a(4096), b(4096);
    hipMemcpy(a.data(), out, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ref, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 4096; ++i) bad += (a[i] != b[i]);
    printf("%-44s %4dx%-4d stages %d: %8.3f ms  %6.1f TF  mismatches vs 32x32x2 kernel %d/4096 (%s)\n", name, BM, BN, NSTAGE, best, 2.0 * M * N * K / best / 1e9, bad, hipGetErrorString(hipGetLastError()));
}

// ---- variant C: 8 waves (4 x 2), 256x128 tile, wave tile 64x64, LDS-DMA, fragment prefetch ----
template <int MINW>
__global__ void __launch_bounds__(512, MINW) kdma8(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int M, int N, int K) {
    constexpr int BK = 32, BM = 256, BN = 128, T = 4, STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = N / BN, tilesM = M / BM;
    int tm_, tn_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 4, width = GM * tilesN, group = lin / width, first = group * GM;
        const int gsz = (tilesM - first) < GM ? (tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz; tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const int drow = lane >> 3, dslot = lane & 7;
    const float* asrc[4]; const float* bsrc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) asrc[i] = A + (size_t)(m0 + wave * 32 + i * 8 + drow) * K + ((dslot ^ drow) << 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) bsrc[i] = W + (size_t)(n0 + wave * 16 + i * 8 + drow) * K + ((dslot ^ drow) << 2);
    auto dma_tile = [&](int kt, int st) {
        float* sA = smem + st * STAGE + wave * 32 * BK;
        float* sB = smem + st * STAGE + BM * BK + wave * 16 * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sA + i * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sB + i * 8 * BK), 16, 0, 0);
    };
    f32x4v acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = K / BK;
    dma_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* sA = smem + cur * STAGE + (wm * 64 + r16) * BK + kq;
        const float* sB = smem + cur * STAGE + BM * BK + (wn * 64 + r16) * BK + kq;
        float af[2][T], bf[2][T];
        { const int sl = (r16 & 7) << 2;
#pragma unroll
          for (int i = 0; i < T; ++i) { af[0][i] = sA[i * 16 * BK + sl]; bf[0][i] = sB[i * 16 * BK + sl]; } }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) { const int sl = ((s + 1) ^ (r16 & 7)) << 2;
#pragma unroll
                for (int i = 0; i < T; ++i) { af[(s + 1) & 1][i] = sA[i * 16 * BK + sl]; bf[(s + 1) & 1][i] = sB[i * 16 * BK + sl]; } }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) { if (kt + 1 < nk) dma_tile(kt + 1, cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
            *(f32x4v*)(out + (size_t)(m0 + wm * 64 + i * 16 + r16) * N + n0 + wn * 64 + j * 16 + kq * 4) = acc[i][j];
}
template <int MINW>
void run_dma8(const char* name, const float* A, const float* W, float* out, int M, int N, int K) {
    size_t lds = (size_t)2 * (256 + 128) * 32 * 4;
    auto kf = kdma8<MINW>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / 256) * (N / 128));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kf, grid, dim3(512), lds, 0, A, W, out, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; if (ms < best) best = ms;
    }
    printf("%-44s 256x128 8 waves: %8.3f ms  %6.1f TF (%s)\n", name, best, 2.0 * M * N * K / best / 1e9, hipGetErrorString(hipGetLastError()));
}

// ---- variant D: one wave per workgroup, 64x64 tile, private LDS stages, NO barriers (only the wave's own vmcnt waits) ----
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int NST>
__global__ void __launch_bounds__(64) kwave(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int M, int N, int K) {
    constexpr int BK = 32, BM = 64, BN = 64, T = 4, STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int tilesN = N / BN, tilesM = M / BM;
    int tm_, tn_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 16, width = GM * tilesN, group = lin / width, first = group * GM;
        const int gsz = (tilesM - first) < GM ? (tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz; tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const int drow = lane >> 3, dslot = lane & 7;
    const float* asrc[8]; const float* bsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { asrc[i] = A + (size_t)(m0 + i * 8 + drow) * K + ((dslot ^ drow) << 2); bsrc[i] = W + (size_t)(n0 + i * 8 + drow) * K + ((dslot ^ drow) << 2); }
    auto dma_tile = [&](int kt, int st) {
        float* sA = smem + st * STAGE; float* sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sA + i * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + kt * BK), (__attribute__((address_space(3))) void*)(sB + i * 8 * BK), 16, 0, 0);
    };
    f32x4v acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = K / BK;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) if (t < nk) dma_tile(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % NST;
        // tile kt landed when at most (NST-2) later tiles (16 instrs each) are outstanding
        const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
        if (NST >= 3 && ahead >= 1) wait_vm<16>(); else wait_vm<0>();
        const float* sA = smem + cur * STAGE + r16 * BK + kq;
        const float* sB = smem + cur * STAGE + BM * BK + r16 * BK + kq;
        float af[2][T], bf[2][T];
        { const int sl = (r16 & 7) << 2;
#pragma unroll
          for (int i = 0; i < T; ++i) { af[0][i] = sA[i * 16 * BK + sl]; bf[0][i] = sB[i * 16 * BK + sl]; } }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) { const int sl = ((s + 1) ^ (r16 & 7)) << 2;
#pragma unroll
                for (int i = 0; i < T; ++i) { af[(s + 1) & 1][i] = sA[i * 16 * BK + sl]; bf[(s + 1) & 1][i] = sB[i * 16 * BK + sl]; } }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the stage of tile kt-1 is free once this wave's reads of it are done: they were, before this tile's first MFMA
            if (s == 0) { if (kt + NST - 1 < nk) dma_tile(kt + NST - 1, (kt + NST - 1) % NST); __builtin_amdgcn_sched_barrier(0); }
        }
    }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
            *(f32x4v*)(out + (size_t)(m0 + i * 16 + r16) * N + n0 + j * 16 + kq * 4) = acc[i][j];
}
template <int NST>
void run_wave(const char* name, const float* A, const float* W, float* out, int M, int N, int K, const float* ref) {
    size_t lds = (size_t)NST * 128 * 32 * 4;
    auto kf = kwave<NST>;
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / 64) * (N / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kf, grid, dim3(64), lds, 0, A, W, out, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; if (ms < best) best = ms;
    }
    std::vector<float> a(4096), b(4096);
    hipMemcpy(a.data(), out, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ref, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 4096; ++i) bad += (a[i] != b[i]);
    printf("%-44s 64x64 1 wave/WG %d stages: %8.3f ms  %6.1f TF  mismatches %d (%s)\n", name, NST, best, 2.0 * M * N * K / best / 1e9, bad, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 4096;
    float *A, *W, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&out, (size_t)M * N * 4);
    std::vector<float> h((size_t)(M > N ? M : N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((double)rand() / RAND_MAX * 2 - 1);
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("M=%d N=%d K=%d\n", M, N, K);
    run<2, 2, 2, 2, 32, 0, 1>("baseline 128x128 BK32", A, W, out, M, N, K);
    float* ref; hipMalloc(&ref, (size_t)M * N * 4);
    run<2, 2, 2, 2, 32, 0, 1>("baseline again (reference output)", A, W, ref, M, N, K);
    run_dma<64, 2, 32, 1>("DMA 128x128 BK32 2 stages", A, W, out, M, N, K, ref);
    run_dma<64, 2, 32, 1, 1>("DMA 128x128 BK32 2 stages frag-prefetch", A, W, out, M, N, K, ref);
    run_dma<64, 2, 32, 1, 2>("DMA 128x128 BK32 2 stages frag-prefetch+schedbar", A, W, out, M, N, K, ref);
    run_dma<32, 2, 32, 1>("DMA 64x64 BK32 2 stages", A, W, out, M, N, K, ref);
    run_dma<32, 2, 32, 1, 1>("DMA 64x64 BK32 2 stages frag-prefetch", A, W, out, M, N, K, ref);
    run_dma<32, 2, 32, 1, 2>("DMA 64x64 BK32 2 stages frag-prefetch+schedbar", A, W, out, M, N, K, ref);
    run_dma8<1>("DMA 256x128 8 waves", A, W, out, M, N, K);
    run_wave<2>("one wave per WG", A, W, out, M, N, K, ref);
    run_wave<3>("one wave per WG", A, W, out, M, N, K, ref);
    return 0;
}
