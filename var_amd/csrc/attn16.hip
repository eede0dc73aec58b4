// attn16.hip — attention of the 16-bit throughput mode: fp16 q / K cache / V cache, fp32 scores, softmax statistics and output
// accumulators, v_mfma_f32_32x32x16_f16 for both products (the reference under its harness' fp16 autocast takes the flash-attention
// path with fp16 q, k, v: basic_var.py:97,113).  Not part of the fp32 parity contract; compared with the CPU twin with a tolerance.
//
// Rounding points of this mode: q, k, v are fp16 (written by varhip_gemm_qkv_f16); scores accumulate in fp32; p = exp(s - m) in fp32
// (hardware exp2), the row sum adds the fp32 p; p is rounded to fp16 for the second product; O accumulates in fp32; out = O / l -> fp16.
//
// Structure as attn.hip: one workgroup per (sample, head), NW waves x 32 queries, "swapped" scores S^T = K_tile . Q^T so that a lane owns
// one query; 32-key tiles with the running maximum.  A 32x32x16 step contracts 16 k: lane (row, half h) supplies k = 8h .. 8h+7 as one
// 16-byte fragment.  The score accumulators of a lane (C layout: register e = key (e & 3) + 8 (e >> 2) + 4h) become the B operand of the
// second product by pairwise conversion to fp16 (registers 8s .. 8s+7 = the fragment of step s); the V^T fragment of that step is stored
// in LDS in the matching key order (cdna_hip_programming.md §3, "an accumulator tile as the next MFMA's operand").
#include <type_traits>
#include "common.h"
#include "elem16.h"

namespace VH16_NS {

typedef vh_e16 h8 __attribute__((ext_vector_type(8)));
typedef vh_e16 h4 __attribute__((ext_vector_type(4)));
typedef vh_e16 h2 __attribute__((ext_vector_type(2)));
typedef short vs4 __attribute__((ext_vector_type(4)));                       // the transposing LDS read's result: four 16-bit elements, whatever their type

#define OLD16 72          // halves per row of the output staging tile (64 + 8 pad)

__device__ __forceinline__ void vh16a_dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ float vh16a_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int NW>
__global__ void __launch_bounds__(NW * 64, NW) k_attn16(const vh_e16* __restrict__ q, const vh_e16* __restrict__ kcache, const vh_e16* __restrict__ vcache,
                                                         vh_e16* __restrict__ out, int l, int H, int curL, int Lmax) {
    constexpr int NT = NW * 64, NIT = (256 + NT - 1) / NT;
    constexpr int KST = 32 * 128, VST = 32 * 128;                 // bytes per K / V stage (32 keys x 64 halves)
    constexpr int OST = NW * 32 * OLD16 * 2;                      // bytes of the output staging tile (reuses the K / V stages)
    __shared__ __attribute__((aligned(16))) char smem[(2 * KST + 2 * VST) > OST ? (2 * KST + 2 * VST) : OST];
    char* sK = smem;
    char* sV = smem + 2 * KST;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int C = H * 64;
    const int t0 = (blockIdx.x * NW + wave) * 32;
    const vh_e16* Kc = kcache + ((int64_t)b * H + hd) * Lmax * 64;
    const vh_e16* Vc = vcache + ((int64_t)b * H + hd) * Lmax * 64;
    const int ntile = (curL + 31) / 32;

    // Q fragments: lane (query r, half h), step t: q[16t + 8h .. + 7]
    h8 qf[4];
    {
        const int t = t0 + r;
        const vh_e16* src = q + ((int64_t)b * l + (t < l ? t : 0)) * C + hd * 64 + h2 * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const h8*)(src + s * 16);
    }

    // K tile by LDS-DMA: 4 pieces of 8 key rows (1 KiB); slot c of row rr holds source chunk c ^ ((rr >> 1) & 7) (conflict-free b128 reads)
    auto dma_k = [&](int kt, int st) {
#pragma unroll
        for (int i = 0; i < (4 + NW - 1) / NW; ++i) {
            const int n = wave + i * NW;
            if (n >= 4) break;
            const int rr = n * 8 + (lane >> 3);
            int key = kt * 32 + rr; key = key < curL ? key : curL - 1;
            vh16a_dma16(Kc, (uint32_t)key * 128u + (uint32_t)(((lane & 7) ^ ((rr >> 1) & 7)) << 4),
                        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sK + st * KST + n * 1024));
        }
    };
    // V tile by LDS-DMA as well, ROW-major as it lies in the cache ([32 keys][64 channels], 128 B per key); the second product wants V^T
    // fragments (lane = channel, 8 keys) — the transposing read ds_read_b64_tr_b16 delivers exactly that: per group of 16 lanes it reads a
    // block of 4 key rows x 16 channels (lane 4q + p supplies the address of row q, channels 4p .. 4p + 3) and gives lane i channel i of the
    // four rows.  Round 2 transposed in registers: 8 two-byte LDS stores per thread and tile, 8-way bank-conflicted.
    // Slot c of row rr holds source chunk c ^ (((rr >> 1) & 1) << 2): the four rows of a block then sit on four different bank quarters.
    auto dma_v = [&](int kt, int st) {
#pragma unroll
        for (int i = 0; i < (4 + NW - 1) / NW; ++i) {
            const int n = wave + i * NW;
            if (n >= 4) break;
            const int rr = n * 8 + (lane >> 3);
            int key = kt * 32 + rr; key = key < curL ? key : curL - 1;      // (keys past curL: their probabilities are exactly 0, any finite row serves)
            vh16a_dma16(Vc, (uint32_t)key * 128u + (uint32_t)(((lane & 7) ^ (((rr >> 1) & 1) << 2)) << 4),
                        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(sV + st * VST + n * 1024));
        }
    };
    // this lane's address inside a V stage for (step s, second half of the step's keys u, channel half c32): block rows = keys
    // 16 s + 8 u + 4 h2 + q, q = (lane >> 2) & 3; channels 16 ((lane >> 4) & 1) + 4 (lane & 3) .. + 3 (+ 32 c32)
    const int vq = (lane >> 2) & 3, vp = lane & 3, vg = (lane >> 4) & 1;
    auto v_addr = [&](int st, int s2, int u, int c32) -> const char* {
        const int row = 16 * s2 + 8 * u + 4 * h2 + vq, chunk = (2 * vg + (vp >> 1) + 4 * c32) ^ (((row >> 1) & 1) << 2);
        return sV + st * VST + row * 128 + (chunk << 4) + (vp & 1) * 8;
    };

    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float lsum = 0.f, mx = -INFINITY;                               // mx in units of log2: (max score) * log2(e)
    const float LOG2E = 1.44269504088896341f;
    dma_k(0, 0); dma_v(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int kxor = (r >> 1) & 7;
    // one key tile; the stage index is a compile-time constant (the loop below runs two tiles per trip): every LDS address of the tile is then a
    // register + immediate instead of ~20 vector adds per tile, and the V^T fragments are requested at the top of the tile, a whole Q.K^T + softmax
    // ahead of their use (16 registers; the reads used to sit right in front of the second product)
    auto tile = [&](int kt, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (kt + 1 < ntile) { dma_k(kt + 1, buf ^ 1); dma_v(kt + 1, buf ^ 1); }
        if (t0 < l) {
            // V^T fragments of step s: elements 0..3 = keys 16s + 4h + 0..3, elements 4..7 = keys 16s + 8 + 4h + 0..3 (the order of pf); [s][2 c32 + u]
            vs4 va[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    va[s][q4] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) vs4*)v_addr(buf, s, q4 & 1, q4 >> 1));
            f32x16 p;
            {
                const char* kb = sK + buf * KST + r * 128;
                h8 kf = *(const h8*)(kb + (((0 + h2) ^ kxor) << 4));
                p = VH16_MFMA_32x32x16(kf, qf[0], (f32x16)(0.f));
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    kf = *(const h8*)(kb + (((2 * s + h2) ^ kxor) << 4));
                    p = VH16_MFMA_32x32x16(kf, qf[s], p);
                }
            }
            const bool ragged = kt * 32 + 32 > curL;
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
            for (int e = 2; e < 16; e += 2) tmax = vh16a_max3(tmax, p[e], p[e + 1]);
            {
                auto xr = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                const float mnew = vh16a_max3(mx, __uint_as_float(xr[0]) * LOG2E, __uint_as_float(xr[1]) * LOG2E);
                if (__any(mnew != mx)) {
                    const float alpha = __builtin_amdgcn_exp2f(mx - mnew);       // first tile: exp2(-inf) = 0, O = l = 0
                    lsum = lsum * alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o0[e] = o0[e] * alpha; o1[e] = o1[e] * alpha; }
                }
                mx = mnew;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[e], LOG2E, -mx)); lsum = lsum + p[e]; }   // masked keys: exp2(-inf) = 0
            // P -> fp16 B fragments: registers 8s .. 8s+7 of this lane = keys 16s + 8(j >> 2) + 4h + (j & 3), j = 0..7
            h8 pf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[s][j] = (vh_e16)p[8 * s + j];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h8 v0 = __builtin_shufflevector(__builtin_bit_cast(h4, va[s][0]), __builtin_bit_cast(h4, va[s][1]), 0, 1, 2, 3, 4, 5, 6, 7);
                const h8 v1 = __builtin_shufflevector(__builtin_bit_cast(h4, va[s][2]), __builtin_bit_cast(h4, va[s][3]), 0, 1, 2, 3, 4, 5, 6, 7);
                o0 = VH16_MFMA_32x32x16(v0, pf[s], o0);      // channels r
                o1 = VH16_MFMA_32x32x16(v1, pf[s], o1);      // channels r + 32
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    {
        int kt = 0;
        for (; kt + 1 < ntile; kt += 2) { tile(kt, std::integral_constant<int, 0>{}); tile(kt + 1, std::integral_constant<int, 1>{}); }
        if (kt < ntile) tile(kt, std::integral_constant<int, 0>{});
    }
    float inv;
    {
        auto xr = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        inv = 1.0f / (__uint_as_float(xr[0]) + __uint_as_float(xr[1]));
    }
    // O^T accumulators: col (lane & 31) = query, row = channel.  fp16, transposed through LDS, stored as 128-byte rows (16 B per lane).
    {
        vh_e16* st = (vh_e16*)smem + wave * 32 * OLD16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            h4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = (vh_e16)(o0[4 * g + e] * inv); c[e] = (vh_e16)(o1[4 * g + e] * inv); }
            *(h4*)(st + r * OLD16 + 8 * g + 4 * h2) = a;               // channels 8g + 4h .. + 3
            *(h4*)(st + r * OLD16 + 32 + 8 * g + 4 * h2) = c;
        }
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int qi = it * 8 + (lane >> 3), t = t0 + qi;
            if (t < l) *(h8*)(out + ((int64_t)b * l + t) * C + hd * 64 + (lane & 7) * 8) = *(const h8*)(st + qi * OLD16 + (lane & 7) * 8);
        }
    }
}

static int attn16_waves(int l) {
    const int nq = (l + 31) / 32;
    int best = 4, waste = ((nq + 3) / 4) * 4 - nq;
    for (int nw = 3; nw >= 1; --nw) { const int w = ((nq + nw - 1) / nw) * nw - nq; if (w < waste) { waste = w; best = nw; } }
    return best;
}

extern "C" int VH16_FN(attn_cached)(const void* q, const void* kcache, const void* vcache, void* out,
                                      int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || curL <= 0 || curL > Lmax) return VARHIP_EINVAL;
    if (B2 > 65535 || H > 65535 || (((uintptr_t)q | (uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)out) & 15)) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_ATTN16, (hipStream_t)stream, 4.0 * B2 * H * (double)l * curL * 64, 2.0 * B2 * H * (2.0 * curL * 64 + 2.0 * l * 64));
    const int nw = attn16_waves(l);
    dim3 grid((l + nw * 32 - 1) / (nw * 32), H, B2);
    hipStream_t s = (hipStream_t)stream;
    const vh_e16 *q_ = (const vh_e16*)q, *k_ = (const vh_e16*)kcache, *v_ = (const vh_e16*)vcache;
    vh_e16* o_ = (vh_e16*)out;
    switch (nw) {
        case 1: hipLaunchKernelGGL((k_attn16<1>), grid, dim3(64), 0, s, q_, k_, v_, o_, l, H, curL, Lmax); break;
        case 2: hipLaunchKernelGGL((k_attn16<2>), grid, dim3(128), 0, s, q_, k_, v_, o_, l, H, curL, Lmax); break;
        case 3: hipLaunchKernelGGL((k_attn16<3>), grid, dim3(192), 0, s, q_, k_, v_, o_, l, H, curL, Lmax); break;
        default: hipLaunchKernelGGL((k_attn16<4>), grid, dim3(256), 0, s, q_, k_, v_, o_, l, H, curL, Lmax); break;
    }
    return vh_launch_status();
}

}  // namespace VH16_NS
