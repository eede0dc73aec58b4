// attn.hip — attention of l new queries over curL cached keys, head_dim 64, fp32 MFMA, no mask (the cache only
// holds scales <= the current one, so block-causality is implicit: reference basic_var.py:107-117).
//
// Structure (per workgroup: one (sample, head), NW waves x 32 queries):
//   - "swapped" scores: S^T = K_tile . Q^T with MFMA 32x32x2 (A = keys, B = queries), so each LANE owns one query
//     (column) and its 16 accumulator registers are 16 keys: row max / row sum are per-lane register reductions
//     plus one exchange between the two lane halves (v_permlane32_swap, no LDS round trip).
//   - one pass over 32-key tiles with a running maximum (flash-attention recurrence): O and the row sum are rescaled by
//     exp(m_old - m_new) when the maximum of some query of the wave moved, p = exp(s - m_new), O^T += V^T . P^T.
//     The 32-key tile is part of the arithmetic contract; the oracle walks the same tiles.
//   - an MFMA step contracts two k values, one from each lane half.  Both products take their operands in the order the data
//     already has: lane half h reads 16 contiguous bytes (4 consecutive k) at offset 16h of every 32-byte group, so step u of
//     group c contracts k = 8c + u (half 0) and k = 8c + 4 + u (half 1).  The summation order of a dot product is therefore
//         k = 0, 4, 1, 5, 2, 6, 3, 7,  8, 12, 9, 13, ...            ("4-interleaved": one fma chain in this order)
//     for the 64 channels of q.k and for the 32 keys of a tile in p.v — stated in include/var_hip.h and restated by the oracle.
//     With it the K tile needs no reordering (it goes global -> LDS by LDS-DMA), Q is loaded straight into registers, and P feeds
//     the second MFMA as the accumulator registers stand (C layout: half h, register 4g + j = key 8g + 4h + j): no lane exchange.
//   Vector-ALU work is paid in matrix time next to fp32 MFMAs, so the softmax is written for instruction count:
//   the exponential is vm_exp_le0 (include/var_math.h: clamp, one fma for n, integer exponent add) in plain fp32 instructions;
//   the row sum is kept as four partial sums per query (two accumulators per lane half):
//       S[h][x] = sum over keys with ((key >> 2) & 1) == h and (key & 1) == x, ascending;   l = (S[0][0] + S[0][1]) + (S[1][0] + S[1][1])
//   and the final normalisation is one reciprocal per query and a multiply per element.
#include "common.h"

#define KPIECE 272      // floats per 4-row piece of the K tile in LDS: 4 x 64 + 16 pad (a piece is one 1 KiB LDS-DMA write; the pad
                        // staggers the pieces over the banks: ds_read_b128 of 16 rows from 4 pieces is 4-way instead of 16-way)
#define VLD 36          // V tile is kept TRANSPOSED in LDS: [64 channels][32 keys] (+4 pad): conflict-free ds_read_b128
#define OLD 65

// single instructions the compiler would otherwise surround with canonicalising v_max(x, x) on MFMA results
__device__ __forceinline__ float vh_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vh_clamp_m87(float x) { float r; asm("v_max_f32 %0, 0xc2ae0000, %1" : "=v"(r) : "v"(x)); return r; }

// one LDS-DMA request: 16 bytes per lane from base + voff to LDS address lds + 16 * lane (M0 written and clobbered in the statement)
__device__ __forceinline__ void vh_attn_dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

// vm_exp_le0 (include/var_math.h) with plain fp32 instructions, on four independent elements whose chains alternate: element for
// element the same operations in the same order.  Next to fp32 MFMAs a packed fp32 instruction (v_pk_fma_f32 ...) costs the matrix
// pipe about three plain ones (fitted from three builds of this kernel), so the unpacked form wins although it is
// twice the instruction count.  (attn.hip is compiled with -fno-slp-vectorize: the compiler would re-pack adjacent operations.)
__device__ __forceinline__ void vh_exp_le0_x4(float& a, float& b, float& c, float& d) {
    float x[4] = {vh_clamp_m87(a), vh_clamp_m87(b), vh_clamp_m87(c), vh_clamp_m87(d)}, t[4], r[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = __builtin_fmaf(x[i], 1.44269504088896341f, 12582912.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float n = t[i] - 12582912.0f; r[i] = __builtin_fmaf(n, -0.693145751953125f, x[i]); r[i] = __builtin_fmaf(n, -1.42860682030941723212e-6f, r[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_fmaf(1.9875691500e-4f, r[i], 1.3981999507e-3f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_fmaf(q[i], r[i], 8.3334519073e-3f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_fmaf(q[i], r[i], 4.1665795894e-2f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_fmaf(q[i], r[i], 1.6666665459e-1f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_fmaf(q[i], r[i], 5.0000001201e-1f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float y = __builtin_fmaf(q[i], r[i] * r[i], r[i]) + 1.0f; x[i] = __uint_as_float(__float_as_uint(y) + (__float_as_uint(t[i]) << 23)); }
    a = x[0]; b = x[1]; c = x[2]; d = x[3];
}

__device__ __forceinline__ float vh_exp_le0_s(float x) {
    x = vh_clamp_m87(x);
    const float t = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f), n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float q = __builtin_fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    q = __builtin_fmaf(q, r, 8.3334519073e-3f); q = __builtin_fmaf(q, r, 4.1665795894e-2f);
    q = __builtin_fmaf(q, r, 1.6666665459e-1f); q = __builtin_fmaf(q, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(q, r * r, r) + 1.0f;
    return __uint_as_float(__float_as_uint(y) + (__float_as_uint(t) << 23));
}

// NW waves per workgroup; the register allocation is capped for 4 workgroups per CU (NW waves per SIMD: 128 registers at NW = 4)
template <int NW>
__global__ void __launch_bounds__(NW * 64, NW) k_attn_cached(const float* __restrict__ q, const float* __restrict__ kcache, const float* __restrict__ vcache,
                                                              float* __restrict__ out, int l, int H, int curL, int Lmax) {
    constexpr int NT = NW * 64, NIT = (256 + NT - 1) / NT;       // V staging items (key row, 8-float chunk) per thread and tile
    constexpr int KST = 8 * KPIECE;                               // floats per K stage
    // one LDS array: K stages | V stages; the O transpose at the end aliases it
    __shared__ __attribute__((aligned(16))) float smem[2 * KST + 2 * 64 * VLD];
    float* sK = smem;
    float (*sV)[64 * VLD] = reinterpret_cast<float (*)[64 * VLD]>(smem + 2 * KST);
    float (*sO)[32 * OLD] = reinterpret_cast<float (*)[32 * OLD]>(smem);
    static_assert(4 * 32 * OLD <= 2 * KST + 2 * 64 * VLD, "O staging must fit in the K/V stages");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int C = H * 64;
    const int t0 = (blockIdx.x * NW + wave) * 32;                  // this wave's first query
    const float* Kc = kcache + ((int64_t)b * H + hd) * Lmax * 64;
    const float* Vc = vcache + ((int64_t)b * H + hd) * Lmax * 64;
    const int ntile = (curL + 31) / 32;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- Q fragments straight from global memory: lane (query r, half h) keeps q[8c + 4h + u], c = 0..7, u = 0..3
    float qf[32];
    {
        const int t = t0 + r;
        const float* src = q + ((int64_t)b * l + (t < l ? t : 0)) * C + hd * 64 + h2 * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 a = *(const f32x4*)(src + c * 8);
            qf[c * 4 + 0] = a[0]; qf[c * 4 + 1] = a[1]; qf[c * 4 + 2] = a[2]; qf[c * 4 + 3] = a[3];
        }
    }

    // K tile: 8 pieces of 4 key rows (1 KiB each) by LDS-DMA, piece n of stage st at sK + st*KST + n*KPIECE; wave w issues pieces
    // w, w + NW, ...  Rows past curL repeat the last key (inside the cache; their scores are masked).  The requests are counted by
    // hand: the explicit s_waitcnt vmcnt(0) in front of the barrier that publishes the tile.
    auto dma_k = [&](int kt, int st) {
#pragma unroll
        for (int n = wave; n < 8; n += NW) {
            int key = kt * 32 + n * 4 + (lane >> 4);
            key = key < curL ? key : curL - 1;
            vh_attn_dma16(Kc, (uint32_t)key * 256u + (uint32_t)(lane & 15) * 16u,
                          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sK + st * KST + n * KPIECE));
        }
    };
    // V tile staging through registers, split so the global loads of tile kt+1 are in flight during the math of tile kt and only
    // meet their LDS stores at the end of the tile
    f32x4 gv0[NIT], gv1[NIT];
    auto load_v = [&](int kt) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * NT, sr = item >> 3, sc = item & 7, key = kt * 32 + sr;
            gv0[it] = zero4; gv1[it] = zero4;
            if ((NIT * NT == 256 || item < 256) && key < curL) {
                const float* sv = Vc + (int64_t)key * 64 + sc * 8; gv0[it] = *(const f32x4*)sv; gv1[it] = *(const f32x4*)(sv + 4);
            }
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * NT, sr = item >> 3, sc = item & 7;
            if (NIT * NT != 256 && item >= 256) continue;
            // V^T[c][key]: lane (channel, half h) of a PV step reads the 4 keys 8g + 4h .. + 3 with one ds_read_b128.  (The 8 scalar
            // writes below conflict 8-way on banks; ~64 cycles per tile against 4096 cycles of MFMA.)
            float* dv = &sV[buf][(sc * 8) * VLD + sr];
#pragma unroll
            for (int e = 0; e < 4; ++e) { dv[e * VLD] = gv0[it][e]; dv[(e + 4) * VLD] = gv1[it][e]; }
        }
    };

    // ---- one pass over the key tiles with the running maximum (the flash-attention recurrence, tile = 32 keys):
    //   m' = max(m, max_tile s);  a = exp(m - m');  l = l*a + sum_tile p;  O = O*a + P.V  with p = exp(s - m')
    // (a == 1 exactly when the maximum stands: the rescale is skipped when that holds for every query of the wave — same bits)
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    f32x2 lsum = {0.f, 0.f};
    float mx = -INFINITY;
    dma_k(0, 0); load_v(0); store_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntile) { dma_k(kt + 1, buf ^ 1); load_v(kt + 1); }       // stage buf^1 was last read before the previous barrier
        if (t0 < l) {                                               // (ragged last workgroup: a wave without queries only stages and syncs)
            // S^T tile: rows = keys, col (lane) = query
            f32x16 p;
            {
                const float* kb = sK + buf * KST + (r >> 2) * KPIECE + (r & 3) * 64 + h2 * 4;
                f32x4 kf = *(const f32x4*)kb;
                p = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[0], qf[0], (f32x16)(0.f), 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[1], qf[1], p, 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[2], qf[2], p, 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[3], qf[3], p, 0, 0, 0);
#pragma unroll
                for (int c = 1; c < 8; ++c) {
                    kf = *(const f32x4*)(kb + c * 8);
#pragma unroll
                    for (int u = 0; u < 4; ++u) p = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u], qf[c * 4 + u], p, 0, 0, 0);
                }
            }
            const bool ragged = kt * 32 + 32 > curL;                // only the last tile can hold keys past curL (wave-uniform)
            if (ragged) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                    if (key >= curL) p[e] = -INFINITY;
                }
            }
            // The FIRST read of the score accumulators is compiler-generated code: hipcc pads the MFMA -> VALU read hazard for its own
            // instructions only, never for an inline-asm consumer (an asm v_max3 placed first read the registers before the last MFMA had
            // written them: run-to-run differences of the tile maximum).  The asm reads below depend on it, so they stay behind it.
            float tmax = fmaxf(p[0], p[1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) tmax = vh_max3(tmax, p[e], p[e + 1]);
            {   // both lane halves of a query agree on the tile maximum: after the swap one register holds this half's, the other the other half's
                auto xr = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                const float mnew = vh_max3(mx, __uint_as_float(xr[0]), __uint_as_float(xr[1]));
                if (__any(mnew != mx)) {
                    const float alpha = vh_exp_le0_s(mx - mnew);                   // (first tile: m = -inf, O = l = 0)
                    lsum[0] = lsum[0] * alpha; lsum[1] = lsum[1] * alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] = o0[e] * alpha; o1[e] = o1[e] * alpha; }
                }
                mx = mnew;
            }
            __builtin_amdgcn_sched_barrier(0);
            // four elements at a time: with all sixteen chains interleaved the temporaries do not fit 128 registers (the Q fragments get spilled)
#define VH_EXP4S(E) { float a_ = p[E] - mx, b_ = p[E + 1] - mx, c_ = p[E + 2] - mx, d_ = p[E + 3] - mx; vh_exp_le0_x4(a_, b_, c_, d_); \
                      p[E] = a_; p[E + 1] = b_; p[E + 2] = c_; p[E + 3] = d_; __builtin_amdgcn_sched_barrier(0); }
            VH_EXP4S(0) VH_EXP4S(4) VH_EXP4S(8) VH_EXP4S(12)
#undef VH_EXP4S
            if (ragged) {                                          // keys past curL do not exist: no share of the row sum
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                    if (key >= curL) p[e] = 0.f;
                }
            }
            // PV: register 4g + j of lane half h is key 8g + 4h + j — the B operand of the step that contracts keys 8g + j and 8g + 4 + j
            const float* vb = &sV[buf][r * VLD + h2 * 4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v0 = *(const f32x4*)(vb + g * 8), v1 = *(const f32x4*)(vb + 32 * VLD + g * 8);   // channels r and r+32, keys 8g + 4h ..
                lsum[0] = lsum[0] + p[4 * g]; lsum[1] = lsum[1] + p[4 * g + 1];         // accumulator [x]: keys with (key & 1) == x of this half, ascending
                lsum[0] = lsum[0] + p[4 * g + 2]; lsum[1] = lsum[1] + p[4 * g + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j], p[4 * g + j], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j], p[4 * g + j], o1, 0, 0, 0);
                }
            }
        }
        if (kt + 1 < ntile) store_v(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's K pieces of the next tile have landed
        __syncthreads();
    }
    float inv;
    {
        const float mine = lsum[0] + lsum[1];
        auto xr = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
        const float a = __uint_as_float(xr[0]), c = __uint_as_float(xr[1]);      // in both halves: a = half 0's sum, c = half 1's
        inv = 1.0f / (a + c);
    }

    // ---- O^T accumulators: col (lane&31) = query, row = channel.  Transpose through LDS, store 256-byte rows.
    {
        float* st = sO[wave];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = (e & 3) + 8 * (e >> 2) + 4 * h2;
            st[r * OLD + c] = o0[e] * inv;
            st[r * OLD + 32 + c] = o1[e] * inv;
        }
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
        for (int qi = 0; qi < 32; ++qi) {
            const int t = t0 + qi;
            if (t < l) out[((int64_t)b * l + t) * C + hd * 64 + lane] = st[qi * OLD + lane];
        }
    }
}

// waves per workgroup for l queries: whole 32-query waves, as few idle ones as possible (l = 169 -> 2 workgroups x 3 waves,
// l = 36..64 -> 1 x 2, l <= 32 -> 1 x 1); the K/V tiles are staged once per workgroup, so larger workgroups are preferred on ties
static int attn_waves(int l) {
    const int nq = (l + 31) / 32;
    int best = 4, waste = ((nq + 3) / 4) * 4 - nq;
    for (int nw = 3; nw >= 1; --nw) { const int w = ((nq + nw - 1) / nw) * nw - nq; if (w < waste) { waste = w; best = nw; } }
    return best;
}

extern "C" int varhip_attn_cached_f32(const float* q, const float* kcache, const float* vcache, float* out,
                                      int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || curL <= 0 || curL > Lmax) return VARHIP_EINVAL;
    if (B2 > 65535 || H > 65535) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_ATTN, (hipStream_t)stream, 4.0 * B2 * H * (double)l * curL * 64,
               4.0 * B2 * H * (2.0 * curL * 64 + 2.0 * l * 64));
    static const int forced = [] { const char* e = getenv("VARHIP_ATTN_WAVES"); return e ? atoi(e) : 0; }();   // experiments only
    const int nw = (forced >= 1 && forced <= 4) ? forced : attn_waves(l);
    dim3 grid((l + nw * 32 - 1) / (nw * 32), H, B2);
    hipStream_t s = (hipStream_t)stream;
    switch (nw) {
#define VH_ATTN_LAUNCH(NW_) hipLaunchKernelGGL((k_attn_cached<NW_>), grid, dim3(NW_ * 64), 0, s, q, kcache, vcache, out, l, H, curL, Lmax)
        case 1: VH_ATTN_LAUNCH(1); break;
        case 2: VH_ATTN_LAUNCH(2); break;
        case 3: VH_ATTN_LAUNCH(3); break;
        default: VH_ATTN_LAUNCH(4); break;
#undef VH_ATTN_LAUNCH
    }
    return vh_launch_status();
}
