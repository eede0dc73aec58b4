// gemm16.hip — the 16-bit-input throughput mode of the transformer GEMMs: fp16 operands, fp32 accumulation on
// v_mfma_f32_16x16x32_f16 (the mode the reference's own harness runs in: demo_sample.py:66-68 wraps the call in
// torch.autocast('cuda', dtype=torch.float16), under which every F.linear of basic_var.py computes in fp16).
//
// NOT part of the fp32 parity contract: the MFMA-internal reduction over 32 k is not a k-ascending fma chain, so results agree
// with the CPU twin (oracle/var_oracle.py, f16 mode: same rounding points, fp32 chains) to rounding noise, not bit for bit.
// Rounding points (the contract of this mode): A and W are fp16; products are exact in fp32; accumulation fp32; bias, GELU,
// gamma and the residual add in fp32; ONE rounding to fp16 where the output is fp16.
//
// Data path = k_dma_gemm of gemm.hip at the same byte geometry: a K tile is 64 halves = 128 bytes per row, both operand tiles go
// global -> LDS by LDS-DMA (16 B per lane, bank swizzle on the source side: slot c of row r holds chunk c ^ (r & 7)), two LDS
// stages, one barrier per K tile, weights as the A operand so that a lane ends with 4 consecutive n of one row.  A lane's operand
// fragment of a 16x16x32 step is 8 consecutive k = one 16-byte chunk: one ds_read_b128 per 16-row fragment and step
// (conflict-free with that swizzle), two steps per K tile.
#include "common.h"
#include "elem16.h"

namespace VH16_NS {

typedef vh_e16 h8 __attribute__((ext_vector_type(8)));
typedef vh_e16 h4 __attribute__((ext_vector_type(4)));

struct Gemm16P {
    const vh_e16* A; const vh_e16* W; const float* bias; void* out; const void* resid; const float* gamma;
    int64_t lda, ldw, ldo, ldr, ldg, sA, sW, sO;
    int M, N, K, epi, rows_per_group, out_f16, resid_f16;
    int m_base;                // rows of the GEMM in front of this launch (a GEMM may be issued as two launches over row ranges): A / out / resid are pre-offset, gamma's row
                               // group and the q/k/v epilogue's (image, position) use the absolute row m_base + m
    int tilesM, tilesN;
    int dbg;                   // experiments (k_gemm16p): bit 0 no global stores, bit 1 no epilogue, bits 8.. stagger (odd workgroups sleep dbg >> 8 x 8k cycles first)
    // epi == 3: fused q/k/v epilogue (N = 3C, head_dim 64)
    const float* q_smul; vh_e16* q_out; vh_e16* q_kc; vh_e16* q_vc; float q_plain; int q_l2, q_l, q_pos0, q_Lmax;
};

template <int N> __device__ __forceinline__ void vh16_waitcnt_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
__device__ __forceinline__ void vh16_dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

// GELU (tanh form) for the fp16 output of fc1: x * sigmoid(2u), u = sqrt(2/pi)(x + 0.044715 x^3), with the hardware exp2 / reciprocal
// (this mode is not bit-compared; the twin uses the same formula with libm: difference ~1e-7 relative, far below the fp16 rounding).
// exp(-2u) = exp2(x * (c1 + c2 x^2)) with the constants folded: seven instructions per element — the epilogue of the 256x256 tile is bound by
// vector-instruction issue (measured: it costs 6.6 - 10.7 us of a 26 us tile with the stores taken out)
__device__ __forceinline__ float vh16_gelu(float x) {
    const float t = __builtin_fmaf(x * x, -0.10294323958f, -2.3022081981f);      // -2 log2(e) sqrt(2/pi) * (1 + 0.044715 x^2)
    const float e = __builtin_amdgcn_exp2f(x * t);                              // exp(-2u)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// q / k normalisation factor of the fused q/k/v epilogue: scale / max(|v|, 1e-12) as scale * rsq(max(|v|^2, 1e-24)) on the hardware
// reciprocal square root (1 ulp; the fp16 rounding of q and k is 2^-11) — one instruction where sqrt + IEEE division took ~25
__device__ __forceinline__ float vh16_qk_rn(float ss, float scale) { return scale * __builtin_amdgcn_rsqf(__builtin_fmaxf(ss, 1e-24f)); }

// v + (v of lane ^ 16) / (v of lane ^ 32) on the row / half swaps of gfx950 (plain vector instructions; __shfl_xor is a ds_bpermute round
// trip).  permlane16_swap(a, b) exchanges the odd 16-lane rows of a with the even rows of b: called on two copies of v it leaves, in every
// lane, v in one result and the partner row's v in the other — their sum is the same sum in both partners (addition commutes).
__device__ __forceinline__ float vh16_add_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float vh16_add_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// sum over the 16 lanes of a row group (every lane ends with the total), partners lane^1, ^2, ^4, ^8 in that order, on DPP moves (the
// compiler's __shfl_xor goes through ds_bpermute: four dependent LDS round trips per row).  After the first two steps the four lanes of a quad
// agree, so the half-row mirror (i <-> 7 - i) pairs quads exactly as lane^4 would, and the row mirror (i <-> 15 - i) as lane^8.
__device__ __forceinline__ float vh16_sum16(float v) {
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));    // row_mirror
    return v;
}

// WM x WN waves, each owning (TMW*16) x (TNW*16) outputs.  2x2 waves: 128x128 / 64x128 / 64x64 tiles, two workgroups per CU.
// 2x4 waves with TMW = 8, TNW = 4: the 256x256 tile (one workgroup of 8 waves per CU, 128 KB of LDS): a K tile step then moves 64 KB
// for 8.4 MFLOP = 128 FLOP per L2->LDS byte, twice the 128x128 tile's — at the f16 MFMA rate the smaller tile is bound by what the
// L2s can deliver.
// NST: LDS stages.  2: a K tile is requested half a K step before it is waited for (vmcnt(0) + barrier per K tile).  3 / 4 (the 64-row tiles
// of the small and middle scales, whose launches were bound by one request round trip per K tile — 16 of them in a row at K = 1024): NST - 1
// K tiles in flight, counted vmcnt, the request for K tile kt + NST - 1 goes out right behind the barrier that frees its stage.
template <int TMW, int TNW, int WM = 2, int WN = 2, int NST = 2>
__global__ void __launch_bounds__(64 * WM * WN, 2) k_gemm16(Gemm16P p) {
    constexpr int NWAVE = WM * WN, BM = TMW * 16 * WM, BN = TNW * 16 * WN, ROWB = 128, STAGE = (BM + BN) * ROWB;      // bytes
    constexpr int NIA = BM / NWAVE / 8, NIB = BN / NWAVE / 8;                              // DMA instructions (8 rows x 128 B) per wave and K tile
    static_assert(BM % (NWAVE * 8) == 0 && BN % (NWAVE * 8) == 0, "tile rows must split evenly over the waves' DMA instructions");
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tm_, tn_;
    {   // XCD-contiguous, grouped block order (as k_dma_gemm)
        const int nwg = p.tilesM * p.tilesN, bid = blockIdx.x;
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int GM = 8, width = GM * p.tilesN, group = lin / width, first = group * GM;
        const int gsz = (p.tilesM - first) < GM ? (p.tilesM - first) : GM;
        tm_ = first + (lin % width) % gsz;
        tn_ = (lin % width) / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN, bz = blockIdx.z;
    const char* Ab = (const char*)(p.A + (int64_t)bz * p.sA);
    const char* Wb = (const char*)(p.W + (int64_t)bz * p.sW);

    const int drow = lane >> 3, dslot = lane & 7;
    uint32_t aoff[NIA], boff[NIB];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int m = m0 + wave * (BM / NWAVE) + i * 8 + drow; m = m < p.M ? m : p.M - 1;
        aoff[i] = (uint32_t)((int64_t)m * p.lda * 2 + ((dslot ^ drow) << 4));
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        int n = n0 + wave * (BN / NWAVE) + i * 8 + drow; n = n < p.N ? n : p.N - 1;
        boff[i] = (uint32_t)((int64_t)n * p.ldw * 2 + ((dslot ^ drow) << 4));
    }
    auto dma_tile = [&](int kt, int st) {
        char* sA = smem16 + st * STAGE + wave * (BM / NWAVE) * ROWB;
        char* sB = smem16 + st * STAGE + BM * ROWB + wave * (BN / NWAVE) * ROWB;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            vh16_dma16(Ab + (size_t)kt * ROWB, aoff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 8 * ROWB));
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            vh16_dma16(Wb + (size_t)kt * ROWB, boff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sB + i * 8 * ROWB));
    };

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = p.K / 64;
    const bool idle_wave = (n0 + wn * TNW * 16 >= p.N) || (m0 + wm * TMW * 16 >= p.M);
    auto compute = [&](int cur, auto&& mid) {
        const char* sA = smem16 + cur * STAGE + (wm * TMW * 16 + r16) * ROWB;
        const char* sB = smem16 + cur * STAGE + BM * ROWB + (wn * TNW * 16 + r16) * ROWB;
        if constexpr (TMW * TNW <= 16) {
            // small wave tile: the fragments of both k-steps are fetched up front (32 registers at 4x4)
            h8 am[2][TMW], bn[2][TNW];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int sl = ((4 * s + kq) ^ (r16 & 7)) << 4;          // rows i*16 + r16: (row & 7) == (r16 & 7)
#pragma unroll
                for (int i = 0; i < TMW; ++i) am[s][i] = *(const h8*)(sA + i * 16 * ROWB + sl);
#pragma unroll
                for (int j = 0; j < TNW; ++j) bn[s][j] = *(const h8*)(sB + j * 16 * ROWB + sl);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j) acc[i][j] = VH16_MFMA_16x16x32(bn[s][j], am[s][i], acc[i][j]);
                if (s == 0) mid();
            }
        } else {
            // 8x4 wave tile (128 accumulator registers): one k-step's fragments at a time; the weight fragments of step 1 are fetched
            // while the MFMAs of step 0 run, the activation fragments half a step ahead (register budget: 256 at two waves per SIMD)
            h8 bn[2][TNW], am[TMW];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int sl = ((4 * s + kq) ^ (r16 & 7)) << 4;
#pragma unroll
                for (int j = 0; j < TNW; ++j) bn[s][j] = *(const h8*)(sB + j * 16 * ROWB + sl);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int sl = ((4 * s + kq) ^ (r16 & 7)) << 4;
#pragma unroll
                for (int i = 0; i < TMW; ++i) am[i] = *(const h8*)(sA + i * 16 * ROWB + sl);
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j) acc[i][j] = VH16_MFMA_16x16x32(bn[s][j], am[i], acc[i][j]);
                if (s == 0) mid();
            }
        }
    };
    if constexpr (NST > 2) {
        static_assert(TMW * TNW <= 16 && NST <= 4, "deep pipeline: the small wave tiles only");
        constexpr int NDMA = NIA + NIB;                 // requests per wave and K tile
#pragma unroll
        for (int s2 = 0; s2 < NST - 1; ++s2) if (s2 < nk) dma_tile(s2, s2);
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            // K tile kt has landed: everything older than the (up to NST - 2) younger tiles already requested
            const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
            if (ahead == 0) vh16_waitcnt_barrier<0>();
            else if (ahead == 1) vh16_waitcnt_barrier<NDMA>();
            else vh16_waitcnt_barrier<2 * NDMA>();
            // stage (kt - 1) % NST was read in the previous K step, which every wave has left: refill it
            if (kt + NST - 1 < nk) dma_tile(kt + NST - 1, (kt + NST - 1) % NST);
            if (!idle_wave) compute(kt % NST, [] {});
        }
        vh16_waitcnt_barrier<0>();                      // the epilogue stages through the same LDS
    } else {
    dma_tile(0, 0);
    vh16_waitcnt_barrier<0>();
    if (idle_wave) {
        for (int kt = 0; kt < nk; ++kt) { if (kt + 1 < nk) dma_tile(kt + 1, (kt & 1) ^ 1); vh16_waitcnt_barrier<0>(); }
    } else {
        if constexpr (TMW * TNW > 16) {
            // one loop body, the stage a run-time offset: inlined once per stage, the accumulators get renamed between the two copies (232
            // registers; 198 this way, same speed) and any further change to the schedule ends at the 256-register ceiling with a spill
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                compute(cur, [&] { if (kt + 1 < nk) dma_tile(kt + 1, cur ^ 1); });
                vh16_waitcnt_barrier<0>();
            }
        } else {
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) {
                compute(0, [&] { dma_tile(kt + 1, 1); });
                vh16_waitcnt_barrier<0>();
                compute(1, [&] { if (kt + 2 < nk) dma_tile(kt + 2, 0); });
                vh16_waitcnt_barrier<0>();
            }
            if (kt < nk) { compute(0, [] {}); vh16_waitcnt_barrier<0>(); }
        }
    }
    }

    // ---- epilogue: acc[i][j][e] = C[m = tile_m(i) + r16][n = tile_n(j) + 4*kq + e]
    if constexpr (NWAVE == 8 && TNW == 4) {
        // Full 256x256 tiles: staged through LDS so that every global access is a whole row of the tile.  Straight from the accumulators a
        // store instruction touches 16 rows x 64 bytes (32 bytes for fp16 results): at the f16 matrix rate those half- and quarter-line
        // accesses cost more than the K loop of a K = 1024 GEMM.  Pass i: every wave parks its 16 rows x 64 columns, then wave w takes rows
        // 4w .. 4w+3 of the 32 parked rows: lane = 4 consecutive columns, so a row is one 1 KiB (fp32) / 512 B (fp16) access.
        if (m0 + BM <= p.M && n0 + BN <= p.N) {
            constexpr int SROW = BN * 4 + 16;                          // bytes per parked row: 256 floats + 16 (a ds_write_b128 serves 8 lanes = 8 rows at a time: row stride = 4 banks mod 32, conflict-free; + 64 was 4-way)
            const int n = n0 + lane * 4;
            const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            int sect = 0, head = 0, Cq = 0, Hh = 0; float sm = 1.0f;
            if (p.epi == 3) { Cq = p.N / 3; sect = n / Cq; head = (n - sect * Cq) >> 6; Hh = Cq >> 6;
                              sm = (p.q_l2 && sect == 0) ? vm_exp(vm_min(p.q_smul[head], 4.605170249938965f)) : 1.0f; }
            char* Ob = (char*)p.out + (int64_t)bz * p.sO * (p.out_f16 ? 2 : 4);
            // fp32 residual rows: requested one pass ahead (fetched inside the pass, each of the 8 passes waited out a memory latency)
            const bool res32 = p.epi == VARHIP_EPI_RESID && !p.resid_f16;
            f32x4 rcur[4], rnxt[4];
            auto res_rows = [&](int i, f32x4* r) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { const int sr = 4 * wave + rr, m = m0 + (sr >> 4) * (TMW * 16) + i * 16 + (sr & 15);
                                                 r[rr] = *(const f32x4*)((const float*)p.resid + (int64_t)m * p.ldr + n); }
            };
            if (res32) res_rows(0, rcur);
            static_assert(2 * 32 * SROW <= 2 * STAGE, "two parking areas");
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                // two parking areas in turn: pass i + 1 parks into the one pass i - 1 was read from, which every wave left before it arrived at
                // the barrier of pass i — one barrier per pass instead of two
                char* const park = smem16 + (i & 1) * 32 * SROW;
                char* wr = park + (wm * 16 + r16) * SROW + (wn * 64 + kq * 4) * 4;
#pragma unroll
                for (int j = 0; j < TNW; ++j) *(f32x4*)(wr + j * 64) = acc[i][j];
                if (res32 && i + 1 < TMW) res_rows(i + 1, rnxt);
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int sr = 4 * wave + rr, m = m0 + (sr >> 4) * (TMW * 16) + i * 16 + (sr & 15);
                    f32x4 v = *(const f32x4*)(park + sr * SROW + lane * 16) + b4;
                    if (p.epi == 3) {
                        if (p.q_l2 && sect < 2) {
                            const float ss = vh16_sum16((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));      // a head = 16 lanes x 4 columns
                            v = v * vh16_qk_rn(ss, sect == 0 ? sm : 1.0f);
                        } else if (!p.q_l2 && sect == 0) v = v * p.q_plain;
                        vh_e16* dst;
                        const int ma = p.m_base + m;
                        if (sect == 0) dst = p.q_out + (int64_t)ma * Cq + head * 64;
                        else { const int bb = ma / p.q_l, t = ma - bb * p.q_l; dst = (sect == 1 ? p.q_kc : p.q_vc) + (((int64_t)bb * Hh + head) * p.q_Lmax + p.q_pos0 + t) * 64; }
                        h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
                        *(h4*)(dst + (lane & 15) * 4) = o;
                        continue;
                    }
                    if (p.epi == VARHIP_EPI_GELU) { v[0] = vh16_gelu(v[0]); v[1] = vh16_gelu(v[1]); v[2] = vh16_gelu(v[2]); v[3] = vh16_gelu(v[3]); }
                    else if (p.epi == VARHIP_EPI_RESID) {
                        if (p.gamma) v = v * *(const f32x4*)(p.gamma + (int64_t)((p.m_base + m) / p.rows_per_group) * p.ldg + n);
                        if (p.resid_f16) { const h4 r4 = *(const h4*)((const vh_e16*)p.resid + (int64_t)m * p.ldr + n);
                                           v[0] = (float)r4[0] + v[0]; v[1] = (float)r4[1] + v[1]; v[2] = (float)r4[2] + v[2]; v[3] = (float)r4[3] + v[3]; }
                        else v = rcur[rr] + v;
                    }
                    if (p.out_f16) { h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
                                     *(h4*)(Ob + ((int64_t)m * p.ldo + n) * 2) = o; }
                    else *(f32x4*)(Ob + ((int64_t)m * p.ldo + n) * 4) = v;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) rcur[rr] = rnxt[rr];
            }
            return;
        }
    }
    // Every other case (the 2x2-wave tiles, and partial 256x256 tiles): staged through LDS too, 16 rows at a time and private to the wave (no
    // workgroup barrier; the K loop ended on one, so the stages are free).  The accumulator layout gives a lane 4 columns of one row: direct
    // stores are 64-byte (fp32) or 32-byte (fp16) pieces of 16 rows per instruction.  Read back, NC = 4*TNW consecutive lanes cover one row's
    // 16*TNW columns: residual reads and stores are contiguous runs of 64*TNW (fp32) / 32*TNW (fp16) bytes, 64 / NC rows per instruction.
    const int nw0 = n0 + wn * TNW * 16;
    if (nw0 >= p.N || m0 + wm * TMW * 16 >= p.M) return;
    constexpr int SROWW = TNW * 64 + 16, STGW = 16 * SROWW, NC = TNW * 4, RPI = 64 / NC, NIT = 16 / RPI;
    static_assert((size_t)NWAVE * STGW <= (size_t)NST * STAGE, "epilogue staging must fit in the stages");
    char* const stg = smem16 + wave * STGW;
    const int col = lane % NC, rl = lane / NC, n = nw0 + col * 4;
    const bool n_ok = n < p.N;                                      // (N % 4 == 0: host-checked)
    char* Ob = (char*)p.out + (int64_t)bz * p.sO * (p.out_f16 ? 2 : 4);
    if (p.epi == 3) {
        if constexpr (TNW == 4) {
            // fused q/k/v post-processing (basic_var.py:98-109): the wave's 64 columns are one head of q, k or v.  L2 norm, scale and the
            // sum of squares in fp32 on the accumulators; q and the cache rows leave as fp16, one 128-byte head row per 16 lanes.
            const int C = p.N / 3, sect = nw0 / C, head = (nw0 - sect * C) >> 6, Hh = C >> 6;
            f32x4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = p.bias ? *(const f32x4*)(p.bias + nw0 + j * 16 + kq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float sm = (p.q_l2 && sect == 0) ? vm_exp(vm_min(p.q_smul[head], 4.605170249938965f)) : 1.0f;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int mrow = m0 + (wm * TMW + i) * 16;
                if (mrow >= p.M) break;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + b4[j];
                if (p.q_l2 && sect < 2) {
                    // sum of squares over the head's 64 columns in the SAME order as the staged 256x256 epilogue above (there lane c = 4j + kq of a
                    // 16-lane group holds columns 16j + 4kq .. +3 and combines lane ^ 1, ^ 2, ^ 4, ^ 8): four consecutive columns first, then kq ^ 1,
                    // kq ^ 2, j ^ 1, j ^ 2 — so q and the cached k come out bit-identical whichever tile a launch picks (batch-size invariance)
                    float sj[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sj[j] = (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
                        sj[j] = vh16_add_xor16(sj[j]);                             // kq ^ 1  (the staged layout's lane ^ 1)
                        sj[j] = vh16_add_xor32(sj[j]);                             // kq ^ 2  (lane ^ 2)
                    }
                    const float w = (sj[0] + sj[1]) + (sj[2] + sj[3]);             // j ^ 1 (lane ^ 4), then j ^ 2 (lane ^ 8)
                    const float rn = vh16_qk_rn(w, sect == 0 ? sm : 1.0f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * rn;
                } else if (!p.q_l2 && sect == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * p.q_plain;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(stg + r16 * SROWW + (j * 16 + kq * 4) * 4) = v[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rr = rl + RPI * k, m = mrow + rr;
                    if (m >= p.M) continue;
                    const f32x4 w = *(const f32x4*)(stg + rr * SROWW + col * 16);
                    vh_e16* dst;
                    const int ma = p.m_base + m;
                    if (sect == 0) dst = p.q_out + (int64_t)ma * C + head * 64;
                    else { const int bb = ma / p.q_l, t = ma - bb * p.q_l; dst = (sect == 1 ? p.q_kc : p.q_vc) + (((int64_t)bb * Hh + head) * p.q_Lmax + p.q_pos0 + t) * 64; }
                    h4 o; o[0] = (vh_e16)w[0]; o[1] = (vh_e16)w[1]; o[2] = (vh_e16)w[2]; o[3] = (vh_e16)w[3];
                    *(h4*)(dst + col * 4) = o;
                }
                asm volatile("" ::: "memory");
            }
        }
        return;
    }
    f32x4 b4[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) { const int nn = nw0 + j * 16 + kq * 4; b4[j] = (p.bias && nn < p.N) ? *(const f32x4*)(p.bias + nn) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    // the fp32 residual rows of the whole wave tile are requested before the first pass (4 registers per row piece: 64 at 4x4, the 2x2-wave
    // kernels have them): fetched pass by pass, each pass waited out a full memory latency with four loads in flight
    constexpr bool PRE = (TMW * NIT <= 16);
    f32x4 rpre[PRE ? TMW * NIT : 1];
    const bool res32 = p.epi == VARHIP_EPI_RESID && !p.resid_f16;
    if constexpr (PRE) {
        if (res32) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int m = m0 + (wm * TMW + i) * 16 + rl + RPI * k;
                    rpre[i * NIT + k] = (m < p.M && n_ok) ? *(const f32x4*)((const float*)p.resid + (int64_t)m * p.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int mrow = m0 + (wm * TMW + i) * 16;
        if (mrow >= p.M) break;                                     // wave-uniform
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            f32x4 v = acc[i][j] + b4[j];
            if (p.epi == VARHIP_EPI_GELU) { v[0] = vh16_gelu(v[0]); v[1] = vh16_gelu(v[1]); v[2] = vh16_gelu(v[2]); v[3] = vh16_gelu(v[3]); }
            *(f32x4*)(stg + r16 * SROWW + (j * 16 + kq * 4) * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own LDS traffic is in order; this keeps the compiler from moving the reads up
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int rr = rl + RPI * k, m = mrow + rr;
            if (m >= p.M || !n_ok) continue;
            f32x4 v = *(const f32x4*)(stg + rr * SROWW + col * 16);
            if (p.epi == VARHIP_EPI_RESID) {
                if (p.gamma) v = v * *(const f32x4*)(p.gamma + (int64_t)((p.m_base + m) / p.rows_per_group) * p.ldg + n);
                if (p.resid_f16) { const h4 r4 = *(const h4*)((const vh_e16*)p.resid + (int64_t)m * p.ldr + n);
                                   v[0] = (float)r4[0] + v[0]; v[1] = (float)r4[1] + v[1]; v[2] = (float)r4[2] + v[2]; v[3] = (float)r4[3] + v[3]; }
                else if constexpr (PRE) v = rpre[i * NIT + k] + v;
                else v = *(const f32x4*)((const float*)p.resid + (int64_t)m * p.ldr + n) + v;
            }
            if (p.out_f16) { h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
                             *(h4*)(Ob + ((int64_t)m * p.ldo + n) * 2) = o; }
            else *(f32x4*)(Ob + ((int64_t)m * p.ldo + n) * 4) = v;
        }
        asm volatile("" ::: "memory");
    }
}

// ---- k_gemm16p: the 256x256 tile as a PERSISTENT workgroup ---------------------------------------------------------------------------
// One workgroup of 8 waves per CU walks a list of tiles instead of exiting after one.  What that buys (12-30 us per tile sit
// outside the K loop of k_gemm16<8,4,2,4>, a third to 60 % of a K = 1024 tile):
//   * the next tile's first K tile is requested at the middle of the current tile's LAST K step and lands while the epilogue runs: no
//     workgroup launch, no cold first request, no pipeline fill between tiles;
//   * the epilogue's stores are not waited for: the K loop of the next tile starts while they drain (the first wait of the next tile is a
//     counted vmcnt that covers the prefetched tile only);
//   * tiles are dealt so that the 32 workgroups of an XCD always work on 32 consecutive tiles of that XCD's contiguous share of the tile
//     list (8 m-tiles x 4 n-tiles: the panels its 4 MB L2 holds), as the one-tile kernel's dispatch order did.
// LDS (all 160 KB): [stage 0: 64 KB][spare: 32 KB][stage 1: 64 KB].  The epilogue parks in the stage the last K step was read from plus the
// spare (96 KB contiguous, either way round) while the other stage receives the next tile.  Arithmetic, fragment reads, epilogue math:
// k_gemm16<8,4,2,4>'s, statement for statement — the results are bit-identical (tests: every tile on ragged shapes).
#define G16P_STAGE 65536
#define G16P_SPARE 32768
__global__ void __launch_bounds__(512, 2) k_gemm16p(Gemm16P p) {
    constexpr int TMW = 8, TNW = 4, WM = 2, WN = 4, NWAVE = 8, BM = 256, BN = 256, ROWB = 128;
    constexpr int NIA = BM / NWAVE / 8, NIB = BN / NWAVE / 8;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int drow = lane >> 3, dslot = lane & 7, r16 = lane & 15, kq = lane >> 4;
    const int nk = p.K / 64, bz = blockIdx.z;
    const char* Ab = (const char*)(p.A + (int64_t)bz * p.sA);
    const char* Wb = (const char*)(p.W + (int64_t)bz * p.sW);
    // this workgroup's tiles: XCD x = bid & 7 owns the contiguous share [start, start + share) of the tile list (as k_gemm16's block order);
    // its workgroup number `loc` (0 .. per_xcd - 1) takes the tiles start + loc, start + loc + per_xcd, ...
    const int nwg = p.tilesM * p.tilesN, bid = blockIdx.x, per_xcd = gridDim.x >> 3;
    const int xq = nwg >> 3, xrem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int share = xq + (xcd < xrem ? 1 : 0), start = xcd < xrem ? xcd * (xq + 1) : xrem * (xq + 1) + (xcd - xrem) * xq;
    if (loc >= share) return;
    if ((p.dbg >> 8) && (loc & 1)) for (int i = 0; i < (p.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    auto tile_of = [&](int lin, int& m0, int& n0) {
        const int GM = 8, width = GM * p.tilesN, group = lin / width, first = group * GM;
        const int gsz = (p.tilesM - first) < GM ? (p.tilesM - first) : GM;
        m0 = (first + (lin % width) % gsz) * BM; n0 = ((lin % width) / gsz) * BN;
    };
    auto offsets = [&](int m0, int n0, uint32_t* aoff, uint32_t* boff) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) { int m = m0 + wave * (BM / NWAVE) + i * 8 + drow; m = m < p.M ? m : p.M - 1; aoff[i] = (uint32_t)((int64_t)m * p.lda * 2 + ((dslot ^ drow) << 4)); }
#pragma unroll
        for (int i = 0; i < NIB; ++i) { int n = n0 + wave * (BN / NWAVE) + i * 8 + drow; n = n < p.N ? n : p.N - 1; boff[i] = (uint32_t)((int64_t)n * p.ldw * 2 + ((dslot ^ drow) << 4)); }
    };
    auto stage_base = [&](int st) -> char* { return smem16 + (st ? G16P_STAGE + G16P_SPARE : 0); };
    auto dma_tile = [&](int kt, int st, const uint32_t* aoff, const uint32_t* boff) {
        char* sA = stage_base(st) + wave * (BM / NWAVE) * ROWB;
        char* sB = stage_base(st) + BM * ROWB + wave * (BN / NWAVE) * ROWB;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            vh16_dma16(Ab + (size_t)kt * ROWB, aoff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sA + i * 8 * ROWB));
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            vh16_dma16(Wb + (size_t)kt * ROWB, boff[i], (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sB + i * 8 * ROWB));
    };

    int lin = start + loc, m0, n0;
    tile_of(lin, m0, n0);
    uint32_t aoff[NIA], boff[NIB], naoff[NIA], nboff[NIB];
    offsets(m0, n0, aoff, boff);
    dma_tile(0, 0, aoff, boff);
    int st = 0;                                        // stage of the K tile about to be multiplied
#pragma unroll 1
    for (int idx = loc; idx < share; idx += per_xcd) {
        const bool has_next = idx + per_xcd < share;
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_of(lin + per_xcd, nm0, nn0);
        // the prefetched first K tile of this tile has landed: it is older than every store of the previous epilogue (32 per wave when that
        // was the full-tile epilogue: a counted wait leaves them in flight; otherwise drain)
        if (idx != loc && !(p.dbg & 3)) vh16_waitcnt_barrier<32>(); else vh16_waitcnt_barrier<0>();
        f32x4 acc[TMW][TNW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt + 1 == nk;
            if (last && has_next) offsets(nm0, nn0, naoff, nboff);
            auto mid = [&] { if (!last) dma_tile(kt + 1, st ^ 1, aoff, boff); else if (has_next) dma_tile(0, st ^ 1, naoff, nboff); };
            {
                const char* sA = stage_base(st) + (wm * TMW * 16 + r16) * ROWB;
                const char* sB = stage_base(st) + BM * ROWB + (wn * TNW * 16 + r16) * ROWB;
                h8 bn[2][TNW], am[TMW];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int sl = ((4 * s2 + kq) ^ (r16 & 7)) << 4;
#pragma unroll
                    for (int j = 0; j < TNW; ++j) bn[s2][j] = *(const h8*)(sB + j * 16 * ROWB + sl);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int sl = ((4 * s2 + kq) ^ (r16 & 7)) << 4;
#pragma unroll
                    for (int i = 0; i < TMW; ++i) am[i] = *(const h8*)(sA + i * 16 * ROWB + sl);
#pragma unroll
                    for (int i = 0; i < TMW; ++i)
#pragma unroll
                        for (int j = 0; j < TNW; ++j) acc[i][j] = VH16_MFMA_16x16x32(bn[s2][j], am[i], acc[i][j]);
                    if (s2 == 0) mid();
                }
            }
            // every wave is done reading stage st; except behind the last K step (the epilogue comes first) the next K tile must have landed
            if (!last) vh16_waitcnt_barrier<0>(); else asm volatile("s_barrier" ::: "memory");
            st ^= 1;
        }
        // ---- epilogue of tile (m0, n0): parks in the stage just read (st ^ 1) + the spare; stage st holds / receives the next tile
        char* const parkbase = smem16 + ((st ^ 1) ? G16P_STAGE : 0);
        if (p.dbg & 2) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) asm volatile("" :: "v"(acc[i][j]));
        }
        else {   // (full tiles only: the host sends M % 256 rows and GEMMs with N % 256 != 0 to the one-tile kernels)
            constexpr int SROW = BN * 4 + 16;              // (row stride = 4 banks mod 32: the parking ds_write_b128 of 8 rows are conflict-free)
            static_assert(2 * 32 * SROW <= G16P_STAGE + G16P_SPARE, "two parking areas");
            const int n = n0 + lane * 4;
            const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            int sect = 0, head = 0, Cq = 0, Hh = 0; float sm = 1.0f;
            if (p.epi == 3) { Cq = p.N / 3; sect = n / Cq; head = (n - sect * Cq) >> 6; Hh = Cq >> 6;
                              sm = (p.q_l2 && sect == 0) ? vm_exp(vm_min(p.q_smul[head], 4.605170249938965f)) : 1.0f; }
            char* Ob = (char*)p.out + (int64_t)bz * p.sO * (p.out_f16 ? 2 : 4);
            const bool res32 = p.epi == VARHIP_EPI_RESID && !p.resid_f16;
            f32x4 rcur[4], rnxt[4];
            auto res_rows = [&](int i, f32x4* r) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { const int sr = 4 * wave + rr, m = m0 + (sr >> 4) * (TMW * 16) + i * 16 + (sr & 15);
                                                 r[rr] = *(const f32x4*)((const float*)p.resid + (int64_t)m * p.ldr + n); }
            };
            if (res32) res_rows(0, rcur);
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                char* const park = parkbase + (i & 1) * 32 * SROW;
                char* wr = park + (wm * 16 + r16) * SROW + (wn * 64 + kq * 4) * 4;
#pragma unroll
                for (int j = 0; j < TNW; ++j) *(f32x4*)(wr + j * 64) = acc[i][j];
                if (res32 && i + 1 < TMW) res_rows(i + 1, rnxt);
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int sr = 4 * wave + rr, m = m0 + (sr >> 4) * (TMW * 16) + i * 16 + (sr & 15);
                    f32x4 v = *(const f32x4*)(park + sr * SROW + lane * 16) + b4;
                    if (p.epi == 3) {
                        if (p.q_l2 && sect < 2) {
                            const float ss = vh16_sum16((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));      // a head = 16 lanes x 4 columns
                            v = v * vh16_qk_rn(ss, sect == 0 ? sm : 1.0f);
                        } else if (!p.q_l2 && sect == 0) v = v * p.q_plain;
                        vh_e16* dst;
                        const int ma = p.m_base + m;
                        if (sect == 0) dst = p.q_out + (int64_t)ma * Cq + head * 64;
                        else { const int bb = ma / p.q_l, t = ma - bb * p.q_l; dst = (sect == 1 ? p.q_kc : p.q_vc) + (((int64_t)bb * Hh + head) * p.q_Lmax + p.q_pos0 + t) * 64; }
                        h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
                        if (!(p.dbg & 1)) *(h4*)(dst + (lane & 15) * 4) = o;
                        continue;
                    }
                    if (p.epi == VARHIP_EPI_GELU) { v[0] = vh16_gelu(v[0]); v[1] = vh16_gelu(v[1]); v[2] = vh16_gelu(v[2]); v[3] = vh16_gelu(v[3]); }
                    else if (p.epi == VARHIP_EPI_RESID) {
                        if (p.gamma) v = v * *(const f32x4*)(p.gamma + (int64_t)((p.m_base + m) / p.rows_per_group) * p.ldg + n);
                        if (p.resid_f16) { const h4 r4 = *(const h4*)((const vh_e16*)p.resid + (int64_t)m * p.ldr + n);
                                           v[0] = (float)r4[0] + v[0]; v[1] = (float)r4[1] + v[1]; v[2] = (float)r4[2] + v[2]; v[3] = (float)r4[3] + v[3]; }
                        else v = rcur[rr] + v;
                    }
                    if (p.dbg & 1) { asm volatile("" :: "v"(v)); continue; }
                    if (p.out_f16) { h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
                                     *(h4*)(Ob + ((int64_t)m * p.ldo + n) * 2) = o; }
                    else *(f32x4*)(Ob + ((int64_t)m * p.ldo + n) * 4) = v;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) rcur[rr] = rnxt[rr];
            }
        }
        if (has_next) {
            lin += per_xcd; m0 = nm0; n0 = nn0;
#pragma unroll
            for (int i = 0; i < NIA; ++i) aoff[i] = naoff[i];
#pragma unroll
            for (int i = 0; i < NIB; ++i) boff[i] = nboff[i];
        }
    }
}


static int launch16p(Gemm16P& p, int batch, hipStream_t stream) {
    static const int dbg = [] { const char* e = getenv("VARHIP_GEMM16_DBG"); return e ? atoi(e) : 0; }();      // experiments only
    p.dbg = dbg;
    constexpr size_t lds = 2 * (size_t)G16P_STAGE + G16P_SPARE;
    p.tilesM = (p.M + 255) / 256; p.tilesN = (p.N + 255) / 256;
    static int ncu = 0;
    static bool attr_done = false;
    if (!attr_done) {
        int dev = 0; (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
        (void)hipFuncSetAttribute((const void*)k_gemm16p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int T = p.tilesM * p.tilesN;
    int per_xcd = ncu / 8 / batch; if (per_xcd < 1) per_xcd = 1;     // workgroups per XCD (batched launches share the CUs over blockIdx.z)
    const int need = (T + 7) / 8;                                      // the largest XCD share
    if (per_xcd > need) per_xcd = need;
    hipLaunchKernelGGL(k_gemm16p, dim3(per_xcd * 8, 1, batch), dim3(512), lds, stream, p);
    return vh_launch_status();
}

template <int TMW, int TNW, int WM = 2, int WN = 2, int NST = 2>
static int launch16(Gemm16P& p, int batch, hipStream_t stream) {
    constexpr int BM = TMW * 16 * WM, BN = TNW * 16 * WN;
    constexpr size_t lds = NST * (size_t)(BM + BN) * 128;
    static_assert(lds <= 80 * 1024 || NST == 2, "two workgroups per CU");
    p.tilesM = (p.M + BM - 1) / BM; p.tilesN = (p.N + BN - 1) / BN;
    auto kfn = k_gemm16<TMW, TNW, WM, WN, NST>;
    static bool attr_done = false;
    if (!attr_done) { if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
    hipLaunchKernelGGL(kfn, dim3(p.tilesM * p.tilesN, 1, batch), dim3(64 * WM * WN), lds, stream, p);
    return vh_launch_status();
}


static int pick_tile16(int M, int N, int batch, bool resid32 = false) {
    // Which tile: a cost estimate in units of ONE ROUND of 256x256 tiles (one workgroup per CU).  A round of 192x256 tiles (one per CU) costs
    // 0.86 (0.75 of the work at 87 % of the rate), a round of 512 128x128 tiles (two workgroups per CU) 0.62 (half the work at 81 %); a launch
    // costs its rounds, full or not.  Fitted on the d16 / B = 64 shapes (tools/bench_kernels.py gemm16 with GEMM16_TILES=-1,0,1,2,3): e.g.
    // M = 21632, N = 1024: 340 / 452 / 1352 tiles = 2.0 / 1.72 / 1.86 -> 192x256 (measured 261 / 235 / 251 us at K = 4096);
    // M = 4608, N = 3072: 216 / 864 tiles = 1.0 / 1.24 -> 256x256 (35 / 39 us); M = 8192, N = 1024: 128 / 512 = 1.0 / 0.62 -> 128x128 (43 / 30 us).
    // Below 256 tiles of 128x128 the 64-row kernels take over (K cannot be split without a reduction pass; those launches are latency-bound).
    // (resid32 — attn.proj, ffn.fc2: the fp32 residual epilogue favours the large tile slightly: measured at M = 12800, N = 1024.)
    if (vh_g_force_tile16 >= 0) return vh_g_force_tile16;
    const int64_t nb256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256) * batch;
    const int64_t nb192 = (int64_t)((M + 191) / 192) * ((N + 255) / 256) * batch;
    const int64_t nb128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (nb128 < 256) return 1;
    const double t256 = (double)((nb256 + 255) / 256) * (resid32 ? 0.95 : 1.0);
    const double t192 = (batch == 1 && nb192 >= 256) ? (double)((nb192 + 255) / 256) * 0.8625 : 1e30;
    const double t128 = (double)((nb128 + 511) / 512) * 0.62;
    if (t256 <= t128 && t256 <= t192) return 2;
    return t192 < t128 * 0.95 ? 3 : 0;
}

// A launch of the 256x256 kernel takes ceil(tiles / 256) rounds of (K loop + epilogue) whatever the last round's occupancy (one workgroup
// per CU): 1360 tiles cost six full rounds.  When the last round would be less than 70 % full, the GEMM is issued as TWO launches over row
// ranges: the rows that fill whole rounds of 256x256 tiles, then the remaining rows with the smaller tiles (two workgroups per CU, short
// rounds).  Every output element is computed by the same MFMA sequence in either kernel (tests: every tile bit-identical), so the split is
// invisible in the results.  Returns the number of leading rows for the 256x256 kernel (0: no split, use pick_tile16).
static int split_rows16(int M, int N, int batch) {
    if (vh_g_force_tile16 >= 0 || batch != 1) return 0;
    static const int off = [] { const char* e = getenv("VARHIP_GEMM16_NOSPLIT"); return e ? atoi(e) : 0; }();       // experiments only
    if (off) return 0;
    const int tilesN = (N + 255) / 256, tilesM = (M + 255) / 256;
    const int64_t T = (int64_t)tilesM * tilesN;
    const int64_t rounds = T / 256;
    const double frac = (double)T / 256.0 - (double)rounds;
    // (measured, tools/bench_kernels.py gemm16 at the d16 shapes: pays from three full rounds on — fc1 at l = 100 / 169: -8 % / -3 %; with one
    // or two rounds in front the small-tile launch costs what the ragged round did)
    if (rounds < 3 || frac == 0.0 || frac >= 0.7) return 0;
    const int mA = (int)((rounds * 256) / tilesN);                    // m-tiles whose tiles fill `rounds` rounds (252 of 256 slots when tilesN = 12)
    if (mA < 1 || mA >= tilesM) return 0;
    return mA * 256;
}

template <typename F>
static int run_gemm16(Gemm16P& p, int batch, hipStream_t stream, bool qkv, bool resid32, double bytes_per_row, double bytes_fixed, F&& small_launch) {
    // one launch, or two over row ranges: (a) split_rows16, (b) the persistent 256x256 kernel takes whole tiles only — the last M % 256 rows
    // go to a small-tile launch.  Every launch is timed in the family of ITS kernel.
    const int M = p.M;
    const bool persist_ok = vh_g_gemm16_persist && (p.N % 256) == 0;
    struct Seg { int m0, rows, pick; } seg[2];
    int nseg = 1;
    auto small_pick = [&](int rows) { return (int64_t)((rows + 127) / 128) * ((p.N + 127) / 128) >= 384 ? 0 : 1; };
    const int mA = split_rows16(M, p.N, batch);
    if (mA) { seg[0] = {0, mA, 2}; seg[1] = {mA, M - mA, small_pick(M - mA)}; nseg = 2; }
    else {
        const int pick = pick_tile16(M, p.N, batch, resid32);
        seg[0] = {0, M, pick};
        // (M % 256 != 0 with the 256x256 tile: the one-tile kernel takes the partial last row of tiles itself — the persistent kernel plus a
        // small-tile launch for the M % 256 rows measured 140.0 against 133.8 us at M = 21632, N = 3072; `persistent` below is false then)
    }
    const char* A0 = (const char*)p.A; char* O0 = (char*)p.out; const char* R0 = (const char*)p.resid;
    int rc = 0;
    for (int i = 0; i < nseg && !rc; ++i) {
        const int m0 = seg[i].m0, rows = seg[i].rows, pick = seg[i].pick;
        p.m_base = m0; p.M = rows;
        p.A = (const vh_e16*)(A0 + (int64_t)m0 * p.lda * 2);
        if (!qkv) {
            p.out = O0 + (int64_t)m0 * p.ldo * (p.out_f16 ? 2 : 4);
            p.resid = R0 ? R0 + (int64_t)m0 * p.ldr * (p.resid_f16 ? 2 : 4) : nullptr;
        }
        const bool persistent = pick == 2 && persist_ok && (rows % 256) == 0;
        // family "gemm16" = k_gemm16p alone (one symbol: its event average is comparable with a rocprofv3 trace); everything else "gemm16_small"
        VhScope scope(persistent ? VH_FAM_GEMM16 : VH_FAM_GEMM16_SMALL, stream, 2.0 * rows * p.N * (double)p.K * batch,
                      batch * (rows * bytes_per_row + (i ? 0.0 : bytes_fixed)));
        if (pick == 3) rc = launch16<6, 4, 2, 4>(p, batch, stream);
        else if (pick == 2) rc = persistent ? launch16p(p, batch, stream) : launch16<8, 4, 2, 4>(p, batch, stream);
        else rc = small_launch(p, pick, batch, stream);
    }
    p.M = M; p.m_base = 0; p.A = (const vh_e16*)A0; p.out = O0; p.resid = R0;
    return rc;
}

extern "C" int VH16_FN(gemm_nt)(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                  void* out, int64_t ldo, int out_f16, int M, int N, int K, int epi,
                                  const void* resid, int64_t ldr, int resid_f16, const float* gamma, int64_t ldg, int rows_per_group,
                                  int batch, int64_t sA, int64_t sW, int64_t sO, varhip_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || batch < 1 || (K & 63) || (N & 3)) return VARHIP_EINVAL;
    if ((lda & 7) || (ldw & 7) || (sA & 7) || (sW & 7) || (((uintptr_t)A | (uintptr_t)W) & 15)) return VARHIP_EINVAL;
    if (((int64_t)(M > 0 ? M - 1 : 0) * lda + K) * 2 >= (1ll << 32) || ((int64_t)(N - 1) * ldw + K) * 2 >= (1ll << 32)) return VARHIP_EINVAL;
    if (epi < 0 || epi > 2 || (epi == VARHIP_EPI_RESID && !resid) || (batch > 1 && (resid || gamma))) return VARHIP_EINVAL;
    if ((ldo & 3) || (sO & 3) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15)) || (resid && ((ldr & 3) || ((uintptr_t)resid & 15))) ||
        (gamma && ((ldg & 3) || ((uintptr_t)gamma & 15)))) return VARHIP_EINVAL;
    if (M == 0) return 0;
    Gemm16P p{};
    p.A = (const vh_e16*)A; p.W = (const vh_e16*)W; p.bias = bias; p.out = out; p.resid = resid; p.gamma = gamma;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr; p.ldg = ldg; p.sA = sA; p.sW = sW; p.sO = sO;
    p.M = M; p.N = N; p.K = K; p.epi = epi; p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; p.out_f16 = out_f16; p.resid_f16 = resid_f16;
    return run_gemm16(p, batch, (hipStream_t)stream, false, epi == VARHIP_EPI_RESID && !resid_f16,
                      2.0 * K + (out_f16 ? 2.0 : 4.0) * N, 2.0 * (double)N * K,
                      [](Gemm16P& q, int pick, int b, hipStream_t s) { if (pick == 0) return launch16<4, 4>(q, b, s);
                          // fewer than one 64x64 tile per CU (the first scales: M = 128 .. 512 rows): 32x32 tiles, four times the workgroups, each streaming a
                          // quarter of the bytes per K tile — such a launch is bound by what ONE CU can request per K tile, not by the chip
                          if (vh_g_gemm16_deep && (int64_t)((q.M + 63) / 64) * ((q.N + 63) / 64) * b < 256) return launch16<1, 1, 2, 2, 4>(q, b, s);
                          return vh_g_gemm16_deep ? launch16<2, 2, 2, 2, 4>(q, b, s) : launch16<2, 2>(q, b, s); });
}

// mat_qkv in the 16-bit mode: fp16 x fp16 -> fp32 accumulators -> (+bias, q/k L2 norm, scale) in fp32 -> fp16 q and fp16 KV-cache rows
extern "C" int VH16_FN(gemm_qkv)(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int M, int C, int K,
                                   const float* scale_mul, float plain_scale, int l2norm,
                                   void* q_out, void* kcache, void* vcache, int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || pos0 < 0 || pos0 + l > Lmax || (l2norm && !scale_mul)) return VARHIP_EINVAL;
    if (C != H * 64 || M != B2 * l || K <= 0 || (K & 63) || (lda & 7) || (ldw & 7)) return VARHIP_EINVAL;
    if (((int64_t)(M - 1) * lda + K) * 2 >= (1ll << 32) || ((int64_t)(3 * C - 1) * ldw + K) * 2 >= (1ll << 32)) return VARHIP_EINVAL;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)q_out | (uintptr_t)kcache | (uintptr_t)vcache) & 15)) return VARHIP_EINVAL;
    Gemm16P p{};
    p.A = (const vh_e16*)A; p.W = (const vh_e16*)W; p.bias = bias; p.lda = lda; p.ldw = ldw;
    p.M = M; p.N = 3 * C; p.K = K; p.epi = 3; p.rows_per_group = 1;
    p.q_smul = scale_mul; p.q_out = (vh_e16*)q_out; p.q_kc = (vh_e16*)kcache; p.q_vc = (vh_e16*)vcache; p.q_plain = plain_scale;
    p.q_l2 = l2norm; p.q_l = l; p.q_pos0 = pos0; p.q_Lmax = Lmax;
    return run_gemm16(p, 1, (hipStream_t)stream, true, false, 2.0 * K + 2.0 * 3.0 * C, 2.0 * 3.0 * C * (double)K,
                      [](Gemm16P& q, int pick, int b, hipStream_t s) { if (pick == 0) return launch16<4, 4>(q, b, s);
                          if (vh_g_gemm16_deep && (int64_t)((q.M + 63) / 64) * ((q.N + 127) / 128) < 256) return launch16<1, 4, 2, 2, 3>(q, b, s);      // 32 rows x one head per wave
                          return vh_g_gemm16_deep ? launch16<2, 4, 2, 2, 3>(q, b, s) : launch16<2, 4>(q, b, s); });
}

}  // namespace VH16_NS
