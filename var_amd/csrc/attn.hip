// attn.hip — attention of l new queries over curL cached keys, head_dim 64, fp32 MFMA, no mask (the cache only
// holds scales <= the current one, so block-causality is implicit: reference basic_var.py:107-117).
//
// Structure (per workgroup: one (sample, head), 4 waves x 32 queries):
//   - "swapped" scores: S^T = K_tile . Q^T with MFMA 32x32x2 (A = keys, B = queries), so each LANE owns one query
//     (column) and its 16 accumulator registers are 16 keys: row max / row sum are per-lane register reductions
//     plus one exchange between the two lane halves.
//   - one pass over 32-key tiles with a running maximum (flash-attention recurrence): O and the row sum are rescaled by
//     exp(m_old - m_new) per tile, p = exp(s - m_new), O^T += V^T . P^T along the natural ascending key order.  The
//     32-key tile is part of the arithmetic contract; the oracle walks the same tiles.
//   - P feeds the second MFMA straight from the accumulator registers: v_permlane32_swap on register pairs turns the
//     C-layout (lane half h holds keys 8g+4h+{0..3}) into the B-operand layout (half h holds key 2s+h).
//   Row-sum order (mirrored by oracle/var_oracle.c): (sum over even keys, ascending) + (sum over odd keys, ascending).
#include "common.h"

#define KLD 68          // K tile row stride (floats): 64 + 4 -> conflict-free ds_read_b128 / ds_write_b128
#define VLD 36          // V tile is kept TRANSPOSED in LDS: [64 channels][32 keys, permuted] (+4 pad), see store_kv
#define OLD 65
#ifndef ATTN_WG_PER_CU
#define ATTN_WG_PER_CU 4      // workgroups per CU the register allocation is capped for
#endif

template <int E>
__device__ __forceinline__ void swap_pair(f32x16& p) {      // registers (E, E+1): afterwards E = keys (2s,2s+1), E+1 = keys (2s+4, 2s+5)
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[E]), __float_as_uint(p[E + 1]), false, false);
    p[E] = __uint_as_float(r[0]);
    p[E + 1] = __uint_as_float(r[1]);
}

__global__ void __launch_bounds__(256, ATTN_WG_PER_CU) k_attn_cached(const float* __restrict__ q, const float* __restrict__ kcache, const float* __restrict__ vcache,
                                                     float* __restrict__ out, int l, int H, int curL, int Lmax) {
    // one LDS array: K stages | V stages; the Q staging at the start and the O transpose at the end alias it
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * KLD + 2 * 64 * VLD];
    float (*sK)[32 * KLD] = reinterpret_cast<float (*)[32 * KLD]>(smem);
    float (*sV)[64 * VLD] = reinterpret_cast<float (*)[64 * VLD]>(smem + 2 * 32 * KLD);
    float (*sO)[32 * OLD] = reinterpret_cast<float (*)[32 * OLD]>(smem);
    static_assert(4 * 32 * OLD <= 2 * 32 * KLD + 2 * 64 * VLD, "O staging must fit in the K/V stages");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int C = H * 64;
    const int t0 = (blockIdx.x * 4 + wave) * 32;                   // this wave's first query
    const float* Kc = kcache + ((int64_t)b * H + hd) * Lmax * 64;
    const float* Vc = vcache + ((int64_t)b * H + hd) * Lmax * 64;
    const int ntile = (curL + 31) / 32;

    // staging role of this thread: key row sr (0..31), chunk sc (8 floats)
    const int sr = tid >> 3, sc = tid & 7;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- Q fragments: stage the wave's 32x64 query tile through sO (permuted like a K tile), keep 32 floats per lane
    float qf[32];
    {
        float* st = sO[wave];                                       // 32 rows x 64 (+1) floats, used as [row][KLD-free layout]
        // each lane loads 32 floats of one row: row = lane&31, half (h2) of the 64 dims
        const int t = t0 + r;
        const float* src = q + ((int64_t)b * l + (t < l ? t : 0)) * C + hd * 64 + h2 * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                               // 4 chunks of 8 dims
            f32x4 a = zero4, bq = zero4;
            if (t < l) { a = *(const f32x4*)(src + c * 8); bq = *(const f32x4*)(src + c * 8 + 4); }
            // even k first, then odd k, inside the chunk (see gemm.hip)
            float* d = st + r * OLD + h2 * 32 + c * 8;
            d[0] = a[0]; d[1] = a[2]; d[2] = bq[0]; d[3] = bq[2];
            d[4] = a[1]; d[5] = a[3]; d[6] = bq[1]; d[7] = bq[3];
        }
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[c * 4 + s] = st[r * OLD + c * 8 + h2 * 4 + s];
        __syncthreads();
    }

    // K/V tile staging, split so the global loads of tile kt+1 are in flight during the math of tile kt and only meet their
    // LDS stores at the end of the tile (a fused load->store waits a full memory latency at the top of every tile)
    f32x4 gk0 = zero4, gk1 = zero4, gv0 = zero4, gv1 = zero4;
    auto load_kv = [&](int kt) {
        const int key = kt * 32 + sr;
        gk0 = zero4; gk1 = zero4; gv0 = zero4; gv1 = zero4;
        if (key < curL) {
            const float* sk = Kc + (int64_t)key * 64 + sc * 8; gk0 = *(const f32x4*)sk; gk1 = *(const f32x4*)(sk + 4);
            const float* sv = Vc + (int64_t)key * 64 + sc * 8; gv0 = *(const f32x4*)sv; gv1 = *(const f32x4*)(sv + 4);
        }
    };
    auto store_kv = [&](int buf) {
        float* dk = &sK[buf][sr * KLD + sc * 8];
        const f32x4 ev = {gk0[0], gk0[2], gk1[0], gk1[2]}, od = {gk0[1], gk0[3], gk1[1], gk1[3]};    // even k first, then odd k (see gemm notes)
        *(f32x4*)dk = ev; *(f32x4*)(dk + 4) = od;
        // V^T[c][pos(key)]: inside every 8-key chunk the even keys come first, then the odd keys, so that lane (c, h2) of the PV
        // MFMAs (A operand = V^T, k = key parity h2) gets the keys of 4 consecutive steps with ONE ds_read_b128 — the same trick
        // as the K tile; with a [key][c] tile every step paid its own ds_read_b32 round trip.  (The 8 scalar writes below conflict
        // 8-way on banks; that is ~64 cycles per tile against 4096 cycles of MFMA.)
        const int pk = (sr & ~7) + ((sr & 1) << 2) + ((sr & 7) >> 1);
        float* dv = &sV[buf][(sc * 8) * VLD + pk];
#pragma unroll
        for (int e = 0; e < 4; ++e) { dv[e * VLD] = gv0[e]; dv[(e + 4) * VLD] = gv1[e]; }
    };
    auto scores = [&](int buf, int kt, f32x16& acc) {              // S^T tile: rows = keys, col (lane) = query
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float* kb = &sK[buf][r * KLD + h2 * 4];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 kf = *(const f32x4*)(kb + c * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[c * 4 + s], acc, 0, 0, 0);
        }
        if (kt * 32 + 32 > curL) {                                  // only the last tile can hold keys past curL
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                if (key >= curL) acc[e] = -INFINITY;
            }
        }
    };

    // ---- one pass over the key tiles with the running maximum (the flash-attention recurrence, tile = 32 keys):
    //   m' = max(m, max_tile s);  a = exp(m - m');  l = l*a + sum_tile p;  O = O*a + P.V  with p = exp(s - m')
    // The tile size is part of the arithmetic contract (oracle/var_oracle.c walks the same 32-key tiles), so GPU == oracle bit
    // for bit; the exact two-pass form this replaced spent a third of its MFMAs recomputing the scores.
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float lsum = 0.f, mx = -INFINITY;
    load_kv(0); store_kv(0);
    __syncthreads();
    for (int kt = 0; kt < ntile; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntile) load_kv(kt + 1);
        if (t0 >= l) { if (kt + 1 < ntile) store_kv(buf ^ 1); __syncthreads(); continue; }   // ragged last workgroup: a wave without queries only stages and syncs
        f32x16 p;
        scores(buf, kt, p);
        float tmax = p[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, p[e]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));              // both lane halves of a query agree on the tile maximum
        const float mnew = fmaxf(mx, tmax);
        const float alpha = vm_exp(mx - mnew);                      // 0 on the first tile (m = -inf), 1 when the maximum stands
        mx = mnew;
        lsum = lsum * alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] = o0[e] * alpha; o1[e] = o1[e] * alpha; }
#pragma unroll
        for (int e = 0; e < 16; e += 2) { const f32x2 pe = vh_exp_pair(f32x2{p[e] - mx, p[e + 1] - mx}); p[e] = pe[0]; p[e + 1] = pe[1]; }
        swap_pair<0>(p); swap_pair<2>(p); swap_pair<4>(p); swap_pair<6>(p);
        swap_pair<8>(p); swap_pair<10>(p); swap_pair<12>(p); swap_pair<14>(p);
        // natural key order of the registers: groups of 4 regs (4g..4g+3) hold steps 4g..4g+3 in the order {0,2,1,3}
        const float* vb = &sV[buf][r * VLD + h2 * 4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v0 = *(const f32x4*)(vb + g * 8), v1 = *(const f32x4*)(vb + 32 * VLD + g * 8);   // channels r and r+32, steps 4g..4g+3
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * g + ((u & 1) << 1) + (u >> 1);     // u=0,1,2,3 -> reg 4g+{0,2,1,3}; MFMA step s = 4g+u: keys 2s, 2s+1
                const float pv = p[e];
                lsum = lsum + pv;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[u], pv, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[u], pv, o1, 0, 0, 0);
            }
        }
        if (kt + 1 < ntile) store_kv(buf ^ 1);                     // stage buf^1 was last read before the previous barrier
        __syncthreads();
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);

    // ---- O^T accumulators: col (lane&31) = query, row = channel.  Transpose through LDS, store 256-byte rows.
    {
        float* st = sO[wave];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = (e & 3) + 8 * (e >> 2) + 4 * h2;
            st[r * OLD + c] = o0[e] / ltot;
            st[r * OLD + 32 + c] = o1[e] / ltot;
        }
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
        for (int qi = 0; qi < 32; ++qi) {
            const int t = t0 + qi;
            if (t < l) out[((int64_t)b * l + t) * C + hd * 64 + lane] = st[qi * OLD + lane];
        }
    }
}

extern "C" int varhip_attn_cached_f32(const float* q, const float* kcache, const float* vcache, float* out,
                                      int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || curL <= 0 || curL > Lmax) return VARHIP_EINVAL;
    if (B2 > 65535 || H > 65535) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_ATTN, (hipStream_t)stream, 4.0 * B2 * H * (double)l * curL * 64,
               4.0 * B2 * H * (2.0 * curL * 64 + 2.0 * l * 64));
    dim3 grid((l + 127) / 128, H, B2);
    hipLaunchKernelGGL(k_attn_cached, grid, dim3(256), 0, (hipStream_t)stream, q, kcache, vcache, out, l, H, curL, Lmax);
    return vh_launch_status();
}
