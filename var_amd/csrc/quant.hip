// quant.hip — the multi-scale quantizer step between two transformer passes, and the encode-side nearest-code lookup.
//   varhip_quant_accum_f32 : codebook gather -> bicubic up to PxP -> Phi (3x3 conv residual mix) -> f_hat +=      (quant.py:187-206)
//   varhip_next_map_f32    : area-pool f_hat to the next scale -> word_embed + level/position embedding, x2 for CFG (var.py:185-187)
//   varhip_nearest_code_f32: argmin_v |z - e_v|^2                                                                     (quant.py:150-157)
// These tensors are tiny (B x 16 x 16 x 32 floats): the kernels are one-thread-per-output with sequential fma chains in
// the oracle's order; what matters is that the whole step is 4 launches instead of the reference's ~15 ATen calls.
#include "common.h"

// FROM_H: `codebook` is the scale's embedding map h [B][pn*pn][Cv] itself and row ids are the identity (more_smooth path)
template <bool FROM_H>
__global__ void k_gather_up(const int64_t* __restrict__ idx, const float* __restrict__ codebook, const int32_t* __restrict__ tap_idx,
                            const float* __restrict__ tap_w, float* __restrict__ up, int B, int pn, int P, int Cv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // [B][P][P][Cv]
    if (i >= (int64_t)B * P * P * Cv) return;
    const int c = (int)(i % Cv); int64_t t = i / Cv; const int x = (int)(t % P); t /= P; const int y = (int)(t % P); const int b = (int)(t / P);
    const int64_t base = (int64_t)b * pn * pn;
    auto rowid = [&](int64_t pos) -> int64_t { return FROM_H ? base + pos : idx[base + pos]; };
    if (pn == P) { up[i] = codebook[rowid(y * pn + x) * Cv + c]; return; }
    const int32_t* iy = tap_idx + y * 4; const float* wy = tap_w + y * 4;
    const int32_t* ix = tap_idx + x * 4; const float* wx = tap_w + x * 4;
    float rr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t ro = (int64_t)iy[a] * pn;
        float acc = codebook[rowid(ro + ix[0]) * Cv + c] * wx[0];
        acc = vm_fma(codebook[rowid(ro + ix[1]) * Cv + c], wx[1], acc);
        acc = vm_fma(codebook[rowid(ro + ix[2]) * Cv + c], wx[2], acc);
        acc = vm_fma(codebook[rowid(ro + ix[3]) * Cv + c], wx[3], acc);
        rr[a] = acc;
    }
    float o = rr[0] * wy[0];
    o = vm_fma(rr[1], wy[1], o); o = vm_fma(rr[2], wy[2], o); o = vm_fma(rr[3], wy[3], o);
    up[i] = o;
}

__global__ void k_phi_accum(const float* __restrict__ up, const float* __restrict__ phi_w, const float* __restrict__ phi_b, float ratio, float keep,
                            float* __restrict__ f_hat, int B, int P, int Cv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // [B][P][P][Cv] over co
    if (i >= (int64_t)B * P * P * Cv) return;
    const int co = (int)(i % Cv); int64_t t = i / Cv; const int x = (int)(t % P); t /= P; const int y = (int)(t % P); const int b = (int)(t / P);
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1; if (yy < 0 || yy >= P) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1; if (xx < 0 || xx >= P) continue;
            const float* u = up + (((int64_t)b * P + yy) * P + xx) * Cv;
            const float* w = phi_w + (((int64_t)co * 3 + ky) * 3 + kx) * Cv;
            for (int ci = 0; ci < Cv; ++ci) acc = vm_fma(u[ci], w[ci], acc);
        }
    }
    const float conv = acc + phi_b[co];
    const float hmix = up[i] * keep + conv * ratio;
    f_hat[i] = f_hat[i] + hmix;
}

// Same step, staged through LDS: one workgroup per (sample, row y) holds the Phi weights ([co][9*Cv] rows padded by one float
// -> conflict-free across co) and the three input rows with a zero halo; thread (x, co) walks the identical (ky, kx, ci) fma chain
// (the zero taps of the halo add exact zeros where the plain kernel skips them).  f_rest (nullable) is the encoder's running residual.
__global__ void __launch_bounds__(256) k_phi_accum_lds(const float* __restrict__ up, const float* __restrict__ phi_w, const float* __restrict__ phi_b, float ratio,
                                                       float keep, float* __restrict__ f_hat, float* __restrict__ f_rest, int P, int Cv) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    const int WS = 9 * Cv + 1, RW = (P + 2) * Cv;
    float* sw = psm;                     // [Cv][WS]
    float* su = psm + Cv * WS;           // [3][P+2][Cv]
    const int tid = threadIdx.x, y = blockIdx.x, b = blockIdx.y;
    for (int i = tid; i < Cv * 9 * Cv; i += 256) { const int co = i / (9 * Cv), k = i - co * 9 * Cv; sw[co * WS + k] = phi_w[i]; }
    for (int i = tid; i < 3 * RW; i += 256) {
        const int ky = i / RW, r = i - ky * RW, xx = r / Cv - 1, ci = r % Cv, yy = y + ky - 1;
        su[i] = (yy >= 0 && yy < P && xx >= 0 && xx < P) ? up[(((int64_t)b * P + yy) * P + xx) * Cv + ci] : 0.f;
    }
    __syncthreads();
    for (int o = tid; o < P * Cv; o += 256) {
        const int x = o / Cv, co = o - x * Cv;
        const float* w = sw + co * WS;
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const float* u = su + ky * RW + (x + kx) * Cv;
                const float* wt = w + (ky * 3 + kx) * Cv;
                for (int ci = 0; ci < Cv; ++ci) acc = vm_fma(u[ci], wt[ci], acc);
            }
        const float conv = acc + phi_b[co];
        const int64_t i = (((int64_t)b * P + y) * P + x) * Cv + co;
        const float hmix = su[RW + (x + 1) * Cv + co] * keep + conv * ratio;
        f_hat[i] = f_hat[i] + hmix;
        if (f_rest) f_rest[i] = f_rest[i] - hmix;
    }
}
static bool launch_phi_accum(const float* up, const float* phi_w, const float* phi_b, float ratio, float* f_hat, float* f_rest, int B, int P, int Cv, hipStream_t s) {
    const size_t lds = ((size_t)Cv * (9 * Cv + 1) + 3 * (size_t)(P + 2) * Cv) * sizeof(float);
    if (lds > 64 * 1024 || B > 65535) return false;             // wide codebooks: the plain kernels
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_phi_accum_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_done = true; }
    hipLaunchKernelGGL(k_phi_accum_lds, dim3(P, B), dim3(256), lds, s, up, phi_w, phi_b, ratio, 1.0f - ratio, f_hat, f_rest, P, Cv);
    return true;
}

extern "C" int varhip_quant_accum_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                                      const float* phi_w, const float* phi_b, float ratio, float* up, float* f_hat,
                                      int B, int pn, int P, int Cv, varhip_stream_t stream) {
    if (B <= 0 || pn <= 0 || P <= 0 || pn > P || Cv <= 0 || Cv > 64) return VARHIP_EINVAL;
    if (pn != P && (!tap_idx || !tap_w)) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * P * P * Cv;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 2.0 * tot * 9 * Cv, 16.0 * tot);
    const unsigned blocks = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_gather_up<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, codebook, tap_idx, tap_w, up, B, pn, P, Cv);
    if (!launch_phi_accum(up, phi_w, phi_b, ratio, f_hat, nullptr, B, P, Cv, (hipStream_t)stream))
        hipLaunchKernelGGL(k_phi_accum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, up, phi_w, phi_b, ratio, 1.0f - ratio, f_hat, B, P, Cv);
    return vh_launch_status();
}

extern "C" int varhip_quant_accum_h_f32(const float* h, const int32_t* tap_idx, const float* tap_w, const float* phi_w, const float* phi_b,
                                        float ratio, float* up, float* f_hat, int B, int pn, int P, int Cv, varhip_stream_t stream) {
    if (B <= 0 || pn <= 0 || P <= 0 || pn > P || Cv <= 0 || Cv > 64 || !h) return VARHIP_EINVAL;
    if (pn != P && (!tap_idx || !tap_w)) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * P * P * Cv;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 2.0 * tot * 9 * Cv, 16.0 * tot);
    const unsigned blocks = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_gather_up<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)nullptr, h, tap_idx, tap_w, up, B, pn, P, Cv);
    if (!launch_phi_accum(up, phi_w, phi_b, ratio, f_hat, nullptr, B, P, Cv, (hipStream_t)stream))
        hipLaunchKernelGGL(k_phi_accum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, up, phi_w, phi_b, ratio, 1.0f - ratio, f_hat, B, P, Cv);
    return vh_launch_status();
}

__global__ void k_area_pool(const float* __restrict__ f_hat, float* __restrict__ pooled, int B, int P, int pq, int Cv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // [B][pq*pq][Cv]
    if (i >= (int64_t)B * pq * pq * Cv) return;
    const int c = (int)(i % Cv); int64_t t = i / Cv; const int tt = (int)(t % (pq * pq)); const int b = (int)(t / (pq * pq));
    const int oy = tt / pq, ox = tt % pq;
    const int y0 = (oy * P) / pq, y1 = ((oy + 1) * P + pq - 1) / pq;
    const int x0 = (ox * P) / pq, x1 = ((ox + 1) * P + pq - 1) / pq;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s = s + f_hat[(((int64_t)b * P + y) * P + x) * Cv + c];
    pooled[i] = (s / (float)(y1 - y0)) / (float)(x1 - x0);
}

__global__ void k_word_embed(const float* __restrict__ pooled, const float* __restrict__ word_w, const float* __restrict__ word_b,
                             const float* __restrict__ lvl_pos, float* __restrict__ x_out, int B, int lq, int C, int Cv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // [B][lq][C]
    if (i >= (int64_t)B * lq * C) return;
    const int n = (int)(i % C); const int64_t bt = i / C; const int t = (int)(bt % lq);
    const float* pl = pooled + bt * Cv;
    const float* w = word_w + (int64_t)n * Cv;
    float acc = 0.f;
    for (int c = 0; c < Cv; ++c) acc = vm_fma(pl[c], w[c], acc);
    const float v = (acc + word_b[n]) + lvl_pos[(int64_t)t * C + n];
    x_out[i] = v;
    x_out[i + (int64_t)B * lq * C] = v;
}

// Cv == 32 (every released VAR): thread n keeps its 32 weights in registers and walks WE_ROWS rows of the pooled map, whose
// values are wave-uniform LDS broadcasts; stores are coalesced over n.  Same c-ascending fma chain as k_word_embed.
#define WE_ROWS 16
__global__ void __launch_bounds__(256) k_word_embed32(const float* __restrict__ pooled, const float* __restrict__ word_w, const float* __restrict__ word_b,
                                                      const float* __restrict__ lvl_pos, float* __restrict__ x_out, int64_t rows, int lq, int C) {
    __shared__ __attribute__((aligned(16))) float sp[WE_ROWS * 32];
    const int tid = threadIdx.x, n = blockIdx.x * 256 + tid;
    const int64_t r0 = (int64_t)blockIdx.y * WE_ROWS;
    for (int i = tid; i < WE_ROWS * 32; i += 256) sp[i] = (r0 + i / 32 < rows) ? pooled[r0 * 32 + i] : 0.f;
    __syncthreads();
    if (n >= C) return;
    f32x4 w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = *(const f32x4*)(word_w + (int64_t)n * 32 + q * 4);
    const float bn = word_b[n];
    for (int r = 0; r < WE_ROWS; ++r) {
        const int64_t bt = r0 + r;
        if (bt >= rows) break;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 pv = *(const f32x4*)(sp + r * 32 + q * 4);
            acc = vm_fma(pv[0], w[q][0], acc); acc = vm_fma(pv[1], w[q][1], acc); acc = vm_fma(pv[2], w[q][2], acc); acc = vm_fma(pv[3], w[q][3], acc);
        }
        const float v = (acc + bn) + lvl_pos[(bt % lq) * C + n];
        x_out[bt * C + n] = v;
        x_out[(bt + rows) * C + n] = v;
    }
}
static void launch_word_embed(const float* pooled, const float* word_w, const float* word_b, const float* lvl_pos, float* x_out, int B, int lq, int C, int Cv, hipStream_t s) {
    const int64_t rows = (int64_t)B * lq, t2 = rows * C;
    if (Cv == 32 && !(((uintptr_t)word_w) & 15) && (rows + WE_ROWS - 1) / WE_ROWS <= 65535)
        hipLaunchKernelGGL(k_word_embed32, dim3((C + 255) / 256, (unsigned)((rows + WE_ROWS - 1) / WE_ROWS)), dim3(256), 0, s, pooled, word_w, word_b, lvl_pos, x_out, rows, lq, C);
    else
        hipLaunchKernelGGL(k_word_embed, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, pooled, word_w, word_b, lvl_pos, x_out, B, lq, C, Cv);
}

extern "C" int varhip_next_map_f32(const float* f_hat, const float* word_w, const float* word_b, const float* lvl_pos,
                                   float* x_out, float* pooled, int B, int P, int pq, int C, int Cv, varhip_stream_t stream) {
    if (B <= 0 || P <= 0 || pq <= 0 || pq > P || C <= 0 || Cv <= 0 || !pooled) return VARHIP_EINVAL;
    const int lq = pq * pq;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 2.0 * B * lq * (double)C * Cv, 8.0 * B * lq * (double)C);
    const int64_t t1 = (int64_t)B * lq * Cv, t2 = (int64_t)B * lq * C;
    hipLaunchKernelGGL(k_area_pool, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f_hat, pooled, B, P, pq, Cv);
    (void)t2;
    launch_word_embed(pooled, word_w, word_b, lvl_pos, x_out, B, lq, C, Cv, (hipStream_t)stream);
    return vh_launch_status();
}

// ---- encode side / teacher forcing (quant.py:135-184) -----------------------------------------------------------------------
extern "C" int varhip_area_pool_f32(const float* f, float* pooled, int B, int P, int pq, int Cv, varhip_stream_t stream) {
    if (B <= 0 || P <= 0 || pq <= 0 || pq > P || Cv <= 0) return VARHIP_EINVAL;
    const int64_t t1 = (int64_t)B * pq * pq * Cv;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * B * P * P * Cv);
    hipLaunchKernelGGL(k_area_pool, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, pooled, B, P, pq, Cv);
    return vh_launch_status();
}

extern "C" int varhip_word_embed_f32(const float* pooled, const float* word_w, const float* word_b, const float* lvl_pos,
                                     float* x_out, int B, int lq, int C, int Cv, varhip_stream_t stream) {
    // x[b][t][:] = word_w . pooled[b][t][:] + word_b + lvl_pos[t][:]; writes rows [0, B*lq) and the CFG copy [B*lq, 2*B*lq)
    if (B <= 0 || lq <= 0 || C <= 0 || Cv <= 0) return VARHIP_EINVAL;
    const int64_t t2 = (int64_t)B * lq * C;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 2.0 * t2 * Cv, 8.0 * t2);
    launch_word_embed(pooled, word_w, word_b, lvl_pos, x_out, B, lq, C, Cv, (hipStream_t)stream);
    return vh_launch_status();
}

// residual quantisation step: like varhip_quant_accum_f32, and the same h is subtracted from the running residual f_rest
__global__ void k_phi_accum_rest(const float* __restrict__ up, const float* __restrict__ phi_w, const float* __restrict__ phi_b, float ratio, float keep,
                                 float* __restrict__ f_hat, float* __restrict__ f_rest, int B, int P, int Cv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * P * P * Cv) return;
    const int co = (int)(i % Cv); int64_t t = i / Cv; const int x = (int)(t % P); t /= P; const int y = (int)(t % P); const int b = (int)(t / P);
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1; if (yy < 0 || yy >= P) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1; if (xx < 0 || xx >= P) continue;
            const float* u = up + (((int64_t)b * P + yy) * P + xx) * Cv;
            const float* w = phi_w + (((int64_t)co * 3 + ky) * 3 + kx) * Cv;
            for (int ci = 0; ci < Cv; ++ci) acc = vm_fma(u[ci], w[ci], acc);
        }
    }
    const float conv = acc + phi_b[co];
    const float hmix = up[i] * keep + conv * ratio;
    f_hat[i] = f_hat[i] + hmix;
    f_rest[i] = f_rest[i] - hmix;
}
extern "C" int varhip_quant_residual_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                                         const float* phi_w, const float* phi_b, float ratio, float* up, float* f_hat, float* f_rest,
                                         int B, int pn, int P, int Cv, varhip_stream_t stream) {
    if (B <= 0 || pn <= 0 || P <= 0 || pn > P || Cv <= 0 || Cv > 64 || !f_rest) return VARHIP_EINVAL;
    if (pn != P && (!tap_idx || !tap_w)) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * P * P * Cv;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 2.0 * tot * 9 * Cv, 24.0 * tot);
    const unsigned blocks = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_gather_up<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, codebook, tap_idx, tap_w, up, B, pn, P, Cv);
    if (!launch_phi_accum(up, phi_w, phi_b, ratio, f_hat, f_rest, B, P, Cv, (hipStream_t)stream))
        hipLaunchKernelGGL(k_phi_accum_rest, dim3(blocks), dim3(256), 0, (hipStream_t)stream, up, phi_w, phi_b, ratio, 1.0f - ratio, f_hat, f_rest, B, P, Cv);
    return vh_launch_status();
}

__global__ void k_token_select(const uint8_t* __restrict__ keep, const int64_t* __restrict__ gt, const int64_t* __restrict__ sampled, int64_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keep[i] ? gt[i] : sampled[i];
}
extern "C" int varhip_token_select_i64(const uint8_t* keep, const int64_t* gt, const int64_t* sampled, int64_t* out, int64_t n, varhip_stream_t stream) {
    if (n < 0) return VARHIP_EINVAL;
    if (n == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 25.0 * n);
    hipLaunchKernelGGL(k_token_select, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keep, gt, sampled, out, n);
    return vh_launch_status();
}

// ---- nearest code (VectorQuantizer2.f_to_idxBl_or_fhat, reference quant.py:151-157) --------------------------------------------------
// k_nearest_mfma: 128 queries per workgroup (32 per wave), the codebook streamed through LDS in shards of 512 codes (64 KB of fp32 rows,
// staged once per workgroup and shard instead of once per QUERY as the first kernel did: 43 520 workgroups x 512 KB of L2 reads at B=64).
// The scores z.e are the reference's own GEMM (addmm_ / matmul) and run on the fp32 MFMA: v_mfma_f32_32x32x2_f32 with lane half kk
// supplying channel 2s + kk in step s is ONE fma chain in ascending channel order starting from 0 — bit for bit the oracle's
// `dot = fma(z[c], e[c], dot)` loop.  Per lane: query = lane & 31, 16 codes of every 32-code block (accumulator register
// 4g + j <-> code 8g + 4h + j, h = lane >> 5), scanned in ascending code order with a strict comparison (first index wins a tie); the two
// lane halves of a query are merged by (value, index).  COS = false: d = (|z|^2 + |e|^2) + (-2 z.e), argmin.  COS = true (using_znorm):
// argmax of the chain over (z[c] / max(|z|, 1e-12)) * (e[c] / max(|e|, 1e-12)), every factor rounded as F.normalize rounds it.
#define NC_SHARD 512
#define NC_ROWB 144            // bytes per staged code row: 16 even channels | 16 odd channels | 16 pad (ds_read_b128 of 16 rows: conflict-free)
template <bool COS>
__global__ void __launch_bounds__(256, 2) k_nearest_mfma(const float* __restrict__ z, const float* __restrict__ codebook, int64_t* __restrict__ idx_out, int N, int V) {
    constexpr int CV = 32;
    extern __shared__ __attribute__((aligned(16))) char nc_smem[];
    float* s_ee = (float*)(nc_smem + NC_SHARD * NC_ROWB);              // |e|^2 of the staged codes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, kk = lane >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t q = (q0 + ql < N) ? q0 + ql : (int64_t)N - 1;
    // this lane's operand of the 16 MFMA steps: channels kk, 2 + kk, 4 + kk, ... of its query
    float zf[CV / 2], zz = 0.f;
    {
        float zr[CV];
#pragma unroll
        for (int c4 = 0; c4 < CV / 4; ++c4) { const f32x4 t = *(const f32x4*)(z + q * CV + c4 * 4); zr[4 * c4] = t[0]; zr[4 * c4 + 1] = t[1]; zr[4 * c4 + 2] = t[2]; zr[4 * c4 + 3] = t[3]; }
#pragma unroll
        for (int c = 0; c < CV; ++c) zz = vm_fma(zr[c], zr[c], zz);
        const float zn = vm_max(vm_sqrt(zz), 1e-12f);
#pragma unroll
        for (int s2 = 0; s2 < CV / 2; ++s2) { const float v = z[q * CV + 2 * s2 + kk]; zf[s2] = COS ? v / zn : v; }      // (re-read: a register array indexed by kk would go to scratch)
    }
    float bd = COS ? -INFINITY : INFINITY; int bi = 0;
    for (int v0 = 0; v0 < V; v0 += NC_SHARD) {
        __syncthreads();                                               // the previous shard has been read by every wave
        for (int r = tid; r < NC_SHARD; r += 256) {                    // stage: one thread per code row, channels de-interleaved
            const int v = v0 + r;
            float er[CV];
            if (v < V) {
#pragma unroll
                for (int c4 = 0; c4 < CV / 4; ++c4) { const f32x4 t = *(const f32x4*)(codebook + (int64_t)v * CV + c4 * 4); er[4 * c4] = t[0]; er[4 * c4 + 1] = t[1]; er[4 * c4 + 2] = t[2]; er[4 * c4 + 3] = t[3]; }
            } else {
#pragma unroll
                for (int c = 0; c < CV; ++c) er[c] = 0.f;
            }
            float ee = 0.f;
#pragma unroll
            for (int c = 0; c < CV; ++c) ee = vm_fma(er[c], er[c], ee);
            if (COS) {
                const float en = vm_max(vm_sqrt(ee), 1e-12f);
#pragma unroll
                for (int c = 0; c < CV; ++c) er[c] = er[c] / en;
            }
            float* row = (float*)(nc_smem + r * NC_ROWB);
#pragma unroll
            for (int c4 = 0; c4 < CV / 8; ++c4) {
                *(f32x4*)(row + 4 * c4) = f32x4{er[8 * c4], er[8 * c4 + 2], er[8 * c4 + 4], er[8 * c4 + 6]};
                *(f32x4*)(row + CV / 2 + 4 * c4) = f32x4{er[8 * c4 + 1], er[8 * c4 + 3], er[8 * c4 + 5], er[8 * c4 + 7]};
            }
            s_ee[r] = ee;
        }
        __syncthreads();
        const int nblk = ((V - v0 < NC_SHARD ? V - v0 : NC_SHARD) + 31) / 32;
        for (int blk = 0; blk < nblk; ++blk) {
            const float* er = (const float*)(nc_smem + (blk * 32 + ql) * NC_ROWB) + kk * (CV / 2);      // A operand: code row ql of the block, this half's channels
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < CV / 8; ++c4) {
                const f32x4 ef = *(const f32x4*)(er + 4 * c4);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[u], zf[4 * c4 + u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = blk * 32 + 8 * g + 4 * kk;              // first of this lane's four codes of the group (shard-local)
                const f32x4 e4 = *(const f32x4*)(s_ee + cb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int code = v0 + cb + j;
                    if (COS) { const float d = acc[4 * g + j]; if (d > bd && code < V) { bd = d; bi = code; } }
                    else { const float d = (zz + e4[j]) + (-2.0f * acc[4 * g + j]); if (d < bd && code < V) { bd = d; bi = code; } }
                }
            }
        }
    }
    {   // the other half of the query's codes: smaller distance (larger cosine) wins, the smaller index on a tie
        const float od = __shfl_xor(bd, 32, 64); const int oi = __shfl_xor(bi, 32, 64);
        const bool better = COS ? (od > bd) : (od < bd);
        if (better || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (kk == 0 && q0 + ql < N) idx_out[q0 + ql] = bi;
}
template <bool COS>
static int launch_nearest_mfma(const float* z, const float* codebook, int64_t* idx_out, int N, int V, hipStream_t stream) {
    constexpr size_t lds = (size_t)NC_SHARD * NC_ROWB + NC_SHARD * 4;
    auto kfn = k_nearest_mfma<COS>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
    hipLaunchKernelGGL(kfn, dim3((unsigned)((N + 127) / 128)), dim3(256), lds, stream, z, codebook, idx_out, N, V);
    return vh_launch_status();
}

// fallback for codebooks the MFMA kernel does not take (Cvae != 32): one workgroup per z row; thread t scores codes t, t+256, ...;
// (distance, index) min with first-index ties
__global__ void __launch_bounds__(256) k_nearest_code(const float* __restrict__ z, const float* __restrict__ codebook, int64_t* __restrict__ idx_out, int V, int Cv) {
    __shared__ float sz[64];
    __shared__ float s_d[4]; __shared__ int s_i[4];
    const int tid = threadIdx.x; const int64_t n = blockIdx.x;
    if (tid < Cv) sz[tid] = z[n * Cv + tid];
    __syncthreads();
    float zz = 0.f;
    for (int c = 0; c < Cv; ++c) zz = vm_fma(sz[c], sz[c], zz);
    float bd = INFINITY; int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        const float* e = codebook + (int64_t)v * Cv;
        float ee = 0.f, dot = 0.f;
        for (int c = 0; c < Cv; ++c) { ee = vm_fma(e[c], e[c], ee); dot = vm_fma(sz[c], e[c], dot); }
        const float d = (zz + ee) + (-2.0f * dot);
        if (d < bd) { bd = d; bi = v; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if ((tid & 63) == 0) { s_d[tid >> 6] = bd; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (s_d[w] < bd || (s_d[w] == bd && s_i[w] < bi)) { bd = s_d[w]; bi = s_i[w]; }
        idx_out[n] = bi == 0x7fffffff ? 0 : bi;
    }
}
extern "C" int varhip_nearest_code_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv, varhip_stream_t stream) {
    if (N < 0 || V <= 0 || Cv <= 0 || Cv > 64) return VARHIP_EINVAL;
    if (N == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 4.0 * N * (double)V * Cv, 4.0 * ((double)N * Cv + (double)V * Cv) + 8.0 * N);
    if (Cv == 32 && (((uintptr_t)z | (uintptr_t)codebook) & 15) == 0) return launch_nearest_mfma<false>(z, codebook, idx_out, N, V, (hipStream_t)stream);
    hipLaunchKernelGGL(k_nearest_code, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, z, codebook, idx_out, V, Cv);
    return vh_launch_status();
}

// cosine variant (using_znorm=True, quant.py:151-153): argmax_v  (z / max(|z|, 1e-12)) . (e_v / max(|e_v|, 1e-12)), first index on ties.
// Every factor is rounded as F.normalize would round it (element / norm), the dot product is one c-ascending fma chain.
__global__ void __launch_bounds__(256) k_nearest_code_cos(const float* __restrict__ z, const float* __restrict__ codebook, int64_t* __restrict__ idx_out, int V, int Cv) {
    __shared__ float sz[64];
    __shared__ float s_d[4]; __shared__ int s_i[4];
    const int tid = threadIdx.x; const int64_t n = blockIdx.x;
    if (tid < Cv) sz[tid] = z[n * Cv + tid];
    __syncthreads();
    float zz = 0.f;
    for (int c = 0; c < Cv; ++c) zz = vm_fma(sz[c], sz[c], zz);
    const float zn = vm_max(vm_sqrt(zz), 1e-12f);
    float bd = -INFINITY; int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        const float* e = codebook + (int64_t)v * Cv;
        float ee = 0.f;
        for (int c = 0; c < Cv; ++c) ee = vm_fma(e[c], e[c], ee);
        const float en = vm_max(vm_sqrt(ee), 1e-12f);
        float dot = 0.f;
        for (int c = 0; c < Cv; ++c) dot = vm_fma(sz[c] / zn, e[c] / en, dot);
        if (dot > bd) { bd = dot; bi = v; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if ((tid & 63) == 0) { s_d[tid >> 6] = bd; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (s_d[w] > bd || (s_d[w] == bd && s_i[w] < bi)) { bd = s_d[w]; bi = s_i[w]; }
        idx_out[n] = bi == 0x7fffffff ? 0 : bi;
    }
}
extern "C" int varhip_nearest_code_cos_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv, varhip_stream_t stream) {
    if (N < 0 || V <= 0 || Cv <= 0 || Cv > 64) return VARHIP_EINVAL;
    if (N == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 4.0 * N * (double)V * Cv, 4.0 * ((double)N * Cv + (double)V * Cv) + 8.0 * N);
    if (Cv == 32 && (((uintptr_t)z | (uintptr_t)codebook) & 15) == 0) return launch_nearest_mfma<true>(z, codebook, idx_out, N, V, (hipStream_t)stream);
    hipLaunchKernelGGL(k_nearest_code_cos, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, z, codebook, idx_out, V, Cv);
    return vh_launch_status();
}
