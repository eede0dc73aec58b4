// rowops.hip — bandwidth-bound row/elementwise kernels: SiLU, AdaLN LayerNorm+modulate, q/k/v post-processing with
// in-place KV-cache append, GroupNorm (stats + apply), row softmax, layout copies, prologue embeddings.
// Reductions follow the canonical orders of common.h so results are bit-identical to oracle/var_oracle.c.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------
__global__ void k_silu(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = vm_silu(x[i]);
}
extern "C" int varhip_silu_f32(const float* x, float* y, int64_t n, varhip_stream_t stream) {
    if (n < 0) return VARHIP_EINVAL;
    if (n == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * n);
    int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_silu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return vh_launch_status();
}

__global__ void k_add_bcast(const float* __restrict__ base, const float* __restrict__ cond, float* __restrict__ out, int rows, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = (int64_t)rows * n;
    if (i < tot) out[i] = base[i % n] + cond[i];
}
extern "C" int varhip_add_bcast_f32(const float* base, const float* cond, float* out, int rows, int n, varhip_stream_t stream) {
    if (rows <= 0 || n <= 0) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * rows * n);
    int64_t tot = (int64_t)rows * n;
    hipLaunchKernelGGL(k_add_bcast, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, base, cond, out, rows, n);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// AdaLN: one wave per row.  Lane j holds elements 256*t + 4*j + {0..3}: canonical W64(vw=4) partials.
#define LN_MAXV 10      // C <= 2560
// OUT: 0 = fp32 result; 1 / 2 = rounded to fp16 / bf16 (the A operand of the 16-bit GEMMs); the statistics and the modulation stay fp32
// EARLY: the modulation vectors are requested together with x instead of after the reductions — the small scales (M <= 4096 rows), where a launch is
// two dependent memory round trips long and registers / occupancy do not matter; at large M it was measured 40 % slower (+40 VGPRs)
template <int OUT, bool EARLY = false>
__global__ void __launch_bounds__(256) k_ln_modulate(const float* __restrict__ x, const float* __restrict__ scale, int64_t lds_,
                                                     const float* __restrict__ shift, int64_t ldh, void* __restrict__ out_,
                                                     int M, int C, int rows_per_group, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = (C + 255) / 256;
    const float* xr = x + (int64_t)m * C;
    f32x4 v[LN_MAXV];
    constexpr int NE = EARLY ? 4 : 1;                 // (EARLY serves C <= 1024)
    f32x4 se[NE], he[NE];
    if constexpr (EARLY) {
        const float* sc0 = scale + (int64_t)(m / rows_per_group) * lds_;
        const float* sh0 = shift + (int64_t)(m / rows_per_group) * ldh;
#pragma unroll
        for (int t = 0; t < NE; ++t) { const int i = 256 * t + 4 * lane; if (i < C) { se[t] = *(const f32x4*)(sc0 + i); he[t] = *(const f32x4*)(sh0 + i); } }
    }
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        if (t < nv) {
            const int i = 256 * t + 4 * lane;
            if (i < C) { v[t] = *(const f32x4*)(xr + i); part = part + v[t][0]; part = part + v[t][1]; part = part + v[t][2]; part = part + v[t][3]; }
            else { v[t][0] = v[t][1] = v[t][2] = v[t][3] = 0.f; }
        }
    }
    const float mean = vh_wave_sum(part) / (float)C;
    part = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        if (t < nv) {
            const int i = 256 * t + 4 * lane;
            if (i < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[t][e] = v[t][e] - mean; part = part + v[t][e] * v[t][e]; }
            }
        }
    }
    const float var = vh_wave_sum(part) / (float)C;
    const float rstd = 1.0f / vm_sqrt(var + eps);
    // (fetching the modulation vectors before the reductions was measured: +40 VGPRs, lower occupancy, 40 % slower)
    const float* sc = scale + (int64_t)(m / rows_per_group) * lds_;
    const float* sh = shift + (int64_t)(m / rows_per_group) * ldh;
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        if (t < nv) {
            const int i = 256 * t + 4 * lane;
            if (i < C) {
                f32x4 s4, h4;
                if constexpr (EARLY) { s4 = se[t < NE ? t : 0]; h4 = he[t < NE ? t : 0]; }
                else { s4 = *(const f32x4*)(sc + i); h4 = *(const f32x4*)(sh + i); }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[t][e] * rstd) * (s4[e] + 1.0f) + h4[e];
                if constexpr (OUT == 1) { typedef _Float16 h4_ __attribute__((ext_vector_type(4))); h4_ o16; o16[0] = (_Float16)o[0]; o16[1] = (_Float16)o[1]; o16[2] = (_Float16)o[2]; o16[3] = (_Float16)o[3];
                                          *(h4_*)((_Float16*)out_ + (int64_t)m * C + i) = o16; }
                else if constexpr (OUT == 2) { typedef __bf16 b4_ __attribute__((ext_vector_type(4))); b4_ o16; o16[0] = (__bf16)o[0]; o16[1] = (__bf16)o[1]; o16[2] = (__bf16)o[2]; o16[3] = (__bf16)o[3];
                                               *(b4_*)((__bf16*)out_ + (int64_t)m * C + i) = o16; }
                else *(f32x4*)((float*)out_ + (int64_t)m * C + i) = o;
            }
        }
    }
}
extern "C" int varhip_ln_modulate_f32(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                                      float* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream) {
    if (M < 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV || rows_per_group <= 0 || (ld_scale & 3) || (ld_shift & 3)) return VARHIP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out) & 15) return VARHIP_EINVAL;
    if (M == 0) return 0;
    VhScope sc(VH_FAM_LN, (hipStream_t)stream, 8.0 * M * C, 8.0 * M * C);
    if (M <= 4096 && C <= 1024) hipLaunchKernelGGL((k_ln_modulate<0, true>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, (void*)out, M, C, rows_per_group, eps);
    else hipLaunchKernelGGL((k_ln_modulate<0>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, (void*)out, M, C, rows_per_group, eps);
    return vh_launch_status();
}
extern "C" int varhip_ln_modulate_f16out(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                                         void* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream) {
    if (M < 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV || rows_per_group <= 0 || (ld_scale & 3) || (ld_shift & 3)) return VARHIP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out) & 15) return VARHIP_EINVAL;
    if (M == 0) return 0;
    VhScope sc(VH_FAM_LN, (hipStream_t)stream, 8.0 * M * C, 6.0 * M * C);
    if (M <= 4096 && C <= 1024) hipLaunchKernelGGL((k_ln_modulate<1, true>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, out, M, C, rows_per_group, eps);
    else hipLaunchKernelGGL((k_ln_modulate<1>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, out, M, C, rows_per_group, eps);
    return vh_launch_status();
}
extern "C" int varhip_ln_modulate_bf16out(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                                         void* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream) {
    if (M < 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV || rows_per_group <= 0 || (ld_scale & 3) || (ld_shift & 3)) return VARHIP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out) & 15) return VARHIP_EINVAL;
    if (M == 0) return 0;
    VhScope sc(VH_FAM_LN, (hipStream_t)stream, 8.0 * M * C, 6.0 * M * C);
    if (M <= 4096 && C <= 1024) hipLaunchKernelGGL((k_ln_modulate<2, true>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, out, M, C, rows_per_group, eps);
    else hipLaunchKernelGGL((k_ln_modulate<2>), dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, scale, ld_scale, shift, ld_shift, out, M, C, rows_per_group, eps);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// q/k/v post-processing: one wave per (row, head), lane = channel (head_dim 64).
__global__ void __launch_bounds__(256) k_qkv_prep(const float* __restrict__ qkv, const float* __restrict__ scale_mul, float plain_scale, int l2norm,
                                                  float* __restrict__ q_out, float* __restrict__ kcache, float* __restrict__ vcache,
                                                  int B2, int l, int H, int pos0, int Lmax) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (row, head)
    const int64_t rows = (int64_t)B2 * l;
    if (item >= rows * H) return;
    const int64_t row = item / H; const int h = (int)(item - row * H);
    const int b = (int)(row / l), t = (int)(row - (int64_t)b * l);
    const int C = H * 64;
    const float* src = qkv + row * 3 * C + h * 64 + lane;
    float q = src[0], k = src[C], v = src[2 * C];
    if (l2norm) {
        const float dq = vm_max(vm_sqrt(vh_wave_sum(q * q)), 1e-12f);
        const float dk = vm_max(vm_sqrt(vh_wave_sum(k * k)), 1e-12f);
        const float sm = vm_exp(vm_min(scale_mul[h], 4.605170249938965f));
        q = (q / dq) * sm;
        k = k / dk;
    } else {
        q = q * plain_scale;
    }
    q_out[row * C + h * 64 + lane] = q;
    const int64_t co = (((int64_t)b * H + h) * Lmax + pos0 + t) * 64 + lane;
    kcache[co] = k; vcache[co] = v;
}
extern "C" int varhip_qkv_prep_f32(const float* qkv, const float* scale_mul, float plain_scale, int l2norm,
                                   float* q_out, float* kcache, float* vcache, int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || H <= 0 || pos0 < 0 || pos0 + l > Lmax || (l2norm && !scale_mul)) return VARHIP_EINVAL;
    const int64_t items = (int64_t)B2 * l * H;
    VhScope sc(VH_FAM_QKV, (hipStream_t)stream, 0, 4.0 * items * 64 * 6);
    hipLaunchKernelGGL(k_qkv_prep, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, qkv, scale_mul, plain_scale, l2norm,
                       q_out, kcache, vcache, B2, l, H, pos0, Lmax);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// row softmax (VAE attention, n = HW): one wave per row, lane j holds i = j, j+64, ...: canonical W64(vw=1).
#define SM_MAXE 16      // n <= 1024
__global__ void __launch_bounds__(256) k_softmax_rows(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float e[SM_MAXE];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < SM_MAXE; ++t) {
        const int i = lane + 64 * t;
        if (i < n) { e[t] = x[r * n + i] * scale; m = fmaxf(m, e[t]); }
    }
    m = vh_wave_max(m);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < SM_MAXE; ++t) {
        const int i = lane + 64 * t;
        if (i < n) { e[t] = vm_exp(e[t] - m); part = part + e[t]; }
    }
    const float S = vh_wave_sum(part);
#pragma unroll
    for (int t = 0; t < SM_MAXE; ++t) {
        const int i = lane + 64 * t;
        if (i < n) out[r * n + i] = e[t] / S;
    }
}
extern "C" int varhip_softmax_rows_f32(const float* x, float* out, int64_t rows, int n, float scale, varhip_stream_t stream) {
    if (rows < 0 || n <= 0 || n > 64 * SM_MAXE) return VARHIP_EINVAL;
    if (rows == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * rows * n);
    hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, n, scale);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm (32 groups, channels-last).  Both kernels give every thread one float4 of channels (column q = t % (C/4)) and walk
// pixels with it, so loads/stores are 16-byte and coalesced and all per-channel constants live in registers.
// Statistics: pass 1, block = (chunk of GN_PIX pixels, sample): fp64 sum / sum of squares per thread, folded through LDS to
// the 32 groups -> scratch[b][chunk][g][2]; pass 2 sums the chunks in order.  Fixed order: run-to-run deterministic.
#define GN_PIX 256
__global__ void __launch_bounds__(256) k_gn_partial(const float* __restrict__ x, double* __restrict__ scratch, int HW, int C, int G, int nchunk) {
    extern __shared__ double gsm[];                 // [rows_per_pass][C][2]
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C4 = C >> 2, rpp = 256 / C4, q = tid % C4, prow = tid / C4;
    const int p0 = chunk * GN_PIX, p1 = (p0 + GN_PIX < HW) ? p0 + GN_PIX : HW;
    if (prow < rpp) {
        double s[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
        const float* px = x + ((int64_t)b * HW + p0 + prow) * C + 4 * q;
        const int64_t step = (int64_t)rpp * C;
        int p = p0 + prow;
        for (; p + 3 * rpp < p1; p += 4 * rpp, px += 4 * step) {    // four loads in flight; accumulation order unchanged (ascending p)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(px + u * step);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double d = (double)v[u][e]; s[e] += d; s2[e] += d * d; }
        }
        for (; p < p1; p += rpp, px += step) {
            const f32x4 v = *(const f32x4*)px;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s[e] += d; s2[e] += d * d; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { gsm[((int64_t)prow * C + 4 * q + e) * 2] = s[e]; gsm[((int64_t)prow * C + 4 * q + e) * 2 + 1] = s2[e]; }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = tid; g < G; g += 256) {
        double s = 0.0, s2 = 0.0;
        for (int rr = 0; rr < rpp; ++rr)
            for (int c = 0; c < cpg; ++c) { s += gsm[((int64_t)rr * C + g * cpg + c) * 2]; s2 += gsm[((int64_t)rr * C + g * cpg + c) * 2 + 1]; }
        double* o = scratch + (((int64_t)b * nchunk + chunk) * G + g) * 2;
        o[0] = s; o[1] = s2;
    }
}
__global__ void k_gn_final(const double* __restrict__ scratch, float* __restrict__ stats, int B, int G, int nchunk, double count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    double s = 0.0, s2 = 0.0;
    for (int c = 0; c < nchunk; ++c) { const double* o = scratch + (((int64_t)b * nchunk + c) * G + g) * 2; s += o[0]; s2 += o[1]; }
    const double mean = s / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
extern "C" int64_t varhip_gn_scratch_elems(int B, int HW, int C, int G) {
    (void)C;
    return (int64_t)B * ((HW + GN_PIX - 1) / GN_PIX) * G * 2;
}
extern "C" int varhip_gn_stats_f32(const float* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps, varhip_stream_t stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || !scratch || (C & 3) || C > 1024 || ((uintptr_t)x & 15)) return VARHIP_EINVAL;
    const int nchunk = (HW + GN_PIX - 1) / GN_PIX;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 4.0 * B * (double)HW * C);
    const size_t lds = (size_t)(256 / (C / 4)) * C * 2 * sizeof(double);
    hipLaunchKernelGGL(k_gn_partial, dim3(nchunk, B), dim3(256), lds, (hipStream_t)stream, x, scratch, HW, C, G, nchunk);
    hipLaunchKernelGGL(k_gn_final, dim3((B * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, stats, B, G, nchunk, (double)HW * (C / G), eps);
    return vh_launch_status();
}

// statistics from the per-block per-channel partials a convolution left behind (varhip_conv3x3_gn_nhwc_f32): blocks in order,
// channels of the group in order, fp64
__global__ void __launch_bounds__(64) k_gn_final_part(const double* __restrict__ part, float* __restrict__ stats, int G, int nblk, int C, double count, float eps) {
    // one wave per (sample, group): lane j sums blocks j, j+64, ... (channels of the group in order), then a fixed butterfly
    const int i = blockIdx.x, lane = threadIdx.x;
    const int b = i / G, g = i - b * G, cpg = C / G;
    double s = 0.0, s2 = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        const double* o = part + (((int64_t)b * nblk + k) * C + g * cpg) * 2;
        for (int c = 0; c < cpg; ++c) { s += o[2 * c]; s2 += o[2 * c + 1]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (lane == 0) {
        const double mean = s / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
extern "C" int varhip_gn_stats_part_f32(const double* part, float* stats, int B, int nblk, int HW, int C, int G, float eps, varhip_stream_t stream) {
    if (B <= 0 || nblk <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || !part) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 16.0 * B * (double)nblk * C);
    hipLaunchKernelGGL(k_gn_final_part, dim3(B * G), dim3(64), 0, (hipStream_t)stream, part, stats, G, nblk, C, (double)HW * (C / G), eps);
    return vh_launch_status();
}

// GroupNorm as one multiply-add per element: table[b][0][c] = rstd * gamma[c], table[b][1][c] = beta[c] - mean * (rstd * gamma[c]) — the two numbers
// k_gn_apply / k_gn16_apply form per channel (same operations, same order), for the convolution that applies the norm to its own input patch
// (varhip_gnconv3x3_nhwc_*, conv16.hip)
__global__ void __launch_bounds__(256) k_gn_scale_shift(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ table, int B, int C, int G) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* st = stats + ((int64_t)b * G + c / (C / G)) * 2;
    const float sc = st[1] * gamma[c];
    table[((int64_t)b * 2) * C + c] = sc;
    table[((int64_t)b * 2 + 1) * C + c] = beta[c] - st[0] * sc;
}
extern "C" int varhip_gn_scale_shift_f32(const float* stats, const float* gamma, const float* beta, float* table, int B, int C, int G, varhip_stream_t stream) {
    if (B <= 0 || C <= 0 || G <= 0 || (C % G) || !stats || !gamma || !beta || !table) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 8.0 * B * C);
    hipLaunchKernelGGL(k_gn_scale_shift, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, table, B, C, G);
    return vh_launch_status();
}

// SiLU of the decoder's GroupNorm: hardware exp2 / rcp (about 1 ulp each) instead of include/var_math.h's reproducible forms.
// The decoder is off the token path (pixels within 1e-3 of the reference, measured ~1e-6); this halves the kernel's VALU work.
__device__ __forceinline__ float gn_fast_silu(float y) {
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y));
}
__global__ void __launch_bounds__(256) k_gn_apply(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ out, int HW, int C, int G, int silu) {
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C4 = C >> 2, rpp = 256 / C4, q = tid % C4, prow = tid / C4, cpg = C / G;
    if (prow >= rpp) return;
    float mean[4], rstd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float* st = stats + ((int64_t)b * G + (4 * q + e) / cpg) * 2; mean[e] = st[0]; rstd[e] = st[1]; }
    const f32x4 g4 = *(const f32x4*)(gamma + 4 * q), b4 = *(const f32x4*)(beta + 4 * q);
    const int p0 = chunk * GN_PIX, p1 = (p0 + GN_PIX < HW) ? p0 + GN_PIX : HW;
    int64_t off = ((int64_t)b * HW + p0 + prow) * C + 4 * q;
    const int64_t step = (int64_t)rpp * C;
    int p = p0 + prow;
    for (; p + 3 * rpp < p1; p += 4 * rpp, off += 4 * step) {      // four independent 16-byte loads in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(x + off + u * step);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = ((v[u][e] - mean[e]) * rstd[e]) * g4[e] + b4[e];
                o[e] = silu ? gn_fast_silu(y) : y;
            }
            *(f32x4*)(out + off + u * step) = o;
        }
    }
    for (; p < p1; p += rpp, off += step) {
        const f32x4 v = *(const f32x4*)(x + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = ((v[e] - mean[e]) * rstd[e]) * g4[e] + b4[e];
            o[e] = silu ? gn_fast_silu(y) : y;
        }
        *(f32x4*)(out + off) = o;
    }
}
extern "C" int varhip_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, float* out,
                                   int B, int HW, int C, int G, int silu, varhip_stream_t stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || (C & 3) || C > 1024) return VARHIP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 8.0 * B * (double)HW * C);
    hipLaunchKernelGGL(k_gn_apply, dim3((HW + GN_PIX - 1) / GN_PIX, B), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, out, HW, C, G, silu);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t tot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // index into out [B][HW][C]
    if (i >= tot) return;
    const int c = (int)(i % C); const int64_t bp = i / C; const int p = (int)(bp % HW); const int64_t b = bp / HW;
    out[i] = in[(b * C + c) * HW + p];
}
__global__ void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t tot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // index into out [B][C][HW]
    if (i >= tot) return;
    const int p = (int)(i % HW); const int64_t bc = i / HW; const int c = (int)(bc % C); const int64_t b = bc / C;
    out[i] = in[(b * HW + p) * C + c];
}
extern "C" int varhip_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int HW, varhip_stream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * C * HW;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * tot);
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, HW, tot);
    return vh_launch_status();
}
// image (B,3,H,W) -> channels-last with the channel count padded by zeros (the conv kernels want Cin % 32 == 0)
__global__ void k_nchw_to_nhwc_pad(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int Cpad, int64_t tot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // index into out [B][HW][Cpad]
    if (i >= tot) return;
    const int c = (int)(i % Cpad); const int64_t bp = i / Cpad; const int p = (int)(bp % HW); const int64_t b = bp / HW;
    out[i] = c < C ? in[(b * C + c) * HW + p] : 0.f;
}
extern "C" int varhip_nchw_to_nhwc_pad_f32(const float* in, float* out, int B, int C, int HW, int Cpad, varhip_stream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || Cpad < C) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * Cpad * HW;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * tot);
    hipLaunchKernelGGL(k_nchw_to_nhwc_pad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, HW, Cpad, tot);
    return vh_launch_status();
}

extern "C" int varhip_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int HW, varhip_stream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return VARHIP_EINVAL;
    const int64_t tot = (int64_t)B * C * HW;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 8.0 * tot);
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, HW, tot);
    return vh_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void k_lvl_pos(const float* __restrict__ lvl_embed, const int64_t* __restrict__ lvl, const float* __restrict__ pos,
                          float* __restrict__ out, int L, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)L * C) return;
    const int t = (int)(i / C), n = (int)(i - (int64_t)t * C);
    out[i] = lvl_embed[lvl[t] * C + n] + pos[i];
}
extern "C" int varhip_lvl_pos_f32(const float* lvl_embed, const int64_t* lvl, const float* pos, float* out, int L, int C, varhip_stream_t stream) {
    if (L <= 0 || C <= 0) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 12.0 * L * C);
    hipLaunchKernelGGL(k_lvl_pos, dim3((unsigned)(((int64_t)L * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lvl_embed, lvl, pos, out, L, C);
    return vh_launch_status();
}

__global__ void k_first_map(const float* __restrict__ class_emb, const int64_t* __restrict__ labels, int num_classes,
                            const float* __restrict__ pos_start, const float* __restrict__ lvl_pos, float* __restrict__ cond,
                            float* __restrict__ x_out, int B, int C, int first_l) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over [2B][C]
    if (i >= (int64_t)2 * B * C) return;
    const int b2 = (int)(i / C), n = (int)(i - (int64_t)b2 * C);
    int64_t cls = b2 < B ? labels[b2] : num_classes;
    if (cls < 0 || cls > num_classes) cls = num_classes;                    // host validates; stay in bounds regardless
    const float cv = class_emb[cls * C + n];
    cond[i] = cv;
    for (int t = 0; t < first_l; ++t)
        x_out[((int64_t)b2 * first_l + t) * C + n] = (cv + pos_start[(int64_t)t * C + n]) + lvl_pos[(int64_t)t * C + n];
}
extern "C" int varhip_first_map_f32(const float* class_emb, const int64_t* labels, int num_classes, const float* pos_start,
                                    const float* lvl_pos, float* cond, float* x_out, int B, int C, int first_l, varhip_stream_t stream) {
    if (B <= 0 || C <= 0 || first_l <= 0 || num_classes < 0) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 16.0 * B * C);
    hipLaunchKernelGGL(k_first_map, dim3((unsigned)(((int64_t)2 * B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       class_emb, labels, num_classes, pos_start, lvl_pos, cond, x_out, B, C, first_l);
    return vh_launch_status();
}
