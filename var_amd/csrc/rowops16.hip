// rowops16.hip — element-wise / row kernels of the 16-bit throughput mode's decoder: GroupNorm on fp16 channels-last activations
// (statistics and the affine + SiLU in fp32 / fp64, one rounding to fp16 at the store) and the fp32 <-> fp16 casts at the mode's edges.
// (reference basic_vae.py:18-19,57-60 under the harness' fp16 autocast: GroupNorm computes in fp32 on fp16 tensors.)
#include "common.h"
#include "elem16.h"

namespace VH16_NS {

typedef vh_e16 h8 __attribute__((ext_vector_type(8)));
typedef vh_e16 h4 __attribute__((ext_vector_type(4)));

__global__ void k_cast_f32_f16(const float* __restrict__ in, vh_e16* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = *(const f32x4*)(in + 4 * i);
    h4 o; o[0] = (vh_e16)v[0]; o[1] = (vh_e16)v[1]; o[2] = (vh_e16)v[2]; o[3] = (vh_e16)v[3];
    *(h4*)(out + 4 * i) = o;
}
__global__ void k_cast_f16_f32(const vh_e16* __restrict__ in, float* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const h4 v = *(const h4*)(in + 4 * i);
    *(f32x4*)(out + 4 * i) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
extern "C" int VH16_FN_CAST_TO(const float* in, void* out, int64_t n, varhip_stream_t stream) {
    if (n < 0 || (n & 3) || (((uintptr_t)in | (uintptr_t)out) & 15)) return VARHIP_EINVAL;
    if (n == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 6.0 * n);
    hipLaunchKernelGGL(k_cast_f32_f16, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (vh_e16*)out, n / 4);
    return vh_launch_status();
}
extern "C" int VH16_FN_CAST_FROM(const void* in, float* out, int64_t n, varhip_stream_t stream) {
    if (n < 0 || (n & 3) || (((uintptr_t)in | (uintptr_t)out) & 15)) return VARHIP_EINVAL;
    if (n == 0) return 0;
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 0, 6.0 * n);
    hipLaunchKernelGGL(k_cast_f16_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const vh_e16*)in, out, n / 4);
    return vh_launch_status();
}

// ---- GroupNorm on fp16 [B][HW][C]: every thread owns 8 consecutive channels (one 16-byte access) and walks pixels -----------------
#define GN16_PIX 256
__global__ void __launch_bounds__(256) k_gn16_partial(const vh_e16* __restrict__ x, double* __restrict__ scratch, int HW, int C, int G, int nchunk) {
    extern __shared__ double gsm16[];               // [rows per pass][C][2]
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C8 = C >> 3, rpp = 256 / C8, q = tid % C8, prow = tid / C8;
    const int p0 = chunk * GN16_PIX, p1 = (p0 + GN16_PIX < HW) ? p0 + GN16_PIX : HW;
    if (prow < rpp) {
        double s[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.0; s2[e] = 0.0; }
        for (int p = p0 + prow; p < p1; p += rpp) {
            const h8 v = *(const h8*)(x + ((int64_t)b * HW + p) * C + 8 * q);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const double d = (double)(float)v[e]; s[e] += d; s2[e] += d * d; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { gsm16[((int64_t)prow * C + 8 * q + e) * 2] = s[e]; gsm16[((int64_t)prow * C + 8 * q + e) * 2 + 1] = s2[e]; }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = tid; g < G; g += 256) {
        double s = 0.0, s2 = 0.0;
        for (int rr = 0; rr < rpp; ++rr)
            for (int c = 0; c < cpg; ++c) { s += gsm16[((int64_t)rr * C + g * cpg + c) * 2]; s2 += gsm16[((int64_t)rr * C + g * cpg + c) * 2 + 1]; }
        double* o = scratch + (((int64_t)b * nchunk + chunk) * G + g) * 2;
        o[0] = s; o[1] = s2;
    }
}
__global__ void k_gn16_final(const double* __restrict__ scratch, float* __restrict__ stats, int B, int G, int nchunk, double count, float eps) {
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
// stats[b][g] = (mean, 1/sqrt(var + eps)); scratch: varhip_gn_scratch_elems(B, HW, C, G) doubles
extern "C" int VH16_FN(gn_stats)(const void* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps, varhip_stream_t stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || !scratch || (C & 7) || C > 2048 || ((uintptr_t)x & 15)) return VARHIP_EINVAL;
    const int nchunk = (HW + GN16_PIX - 1) / GN16_PIX;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 2.0 * B * (double)HW * C);
    const size_t lds = (size_t)(256 / (C / 8)) * C * 2 * sizeof(double);
    if (lds > 64 * 1024) return VARHIP_EINVAL;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_gn16_partial, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_done = true; }
    hipLaunchKernelGGL(k_gn16_partial, dim3(nchunk, B), dim3(256), lds, (hipStream_t)stream, (const vh_e16*)x, scratch, HW, C, G, nchunk);
    hipLaunchKernelGGL(k_gn16_final, dim3((B * G + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, stats, B, G, nchunk, (double)HW * (C / G), eps);
    return vh_launch_status();
}

__device__ __forceinline__ float gn16_silu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y)); }
__global__ void __launch_bounds__(256) k_gn16_apply(const vh_e16* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, vh_e16* __restrict__ out, int HW, int C, int G, int silu) {
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int C8 = C >> 3, rpp = 256 / C8, q = tid % C8, prow = tid / C8, cpg = C / G;
    if (prow >= rpp) return;
    float sc[8], sh[8];                              // y = x * sc + sh with sc = rstd * gamma, sh = beta - mean * rstd * gamma
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float* st = stats + ((int64_t)b * G + (8 * q + e) / cpg) * 2;
        sc[e] = st[1] * gamma[8 * q + e]; sh[e] = beta[8 * q + e] - st[0] * sc[e];
    }
    const int p0 = chunk * GN16_PIX, p1 = (p0 + GN16_PIX < HW) ? p0 + GN16_PIX : HW;
    for (int p = p0 + prow; p < p1; p += rpp) {
        const int64_t off = ((int64_t)b * HW + p) * C + 8 * q;
        const h8 v = *(const h8*)(x + off);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float y = __builtin_fmaf((float)v[e], sc[e], sh[e]); o[e] = (vh_e16)(silu ? gn16_silu(y) : y); }
        *(h8*)(out + off) = o;
    }
}
extern "C" int VH16_FN(gn_apply)(const void* x, const float* stats, const float* gamma, const float* beta, void* out,
                                   int B, int HW, int C, int G, int silu, varhip_stream_t stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || (C & 7) || C > 2048) return VARHIP_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out) & 15) return VARHIP_EINVAL;
    VhScope sc(VH_FAM_GN, (hipStream_t)stream, 0, 4.0 * B * (double)HW * C);
    hipLaunchKernelGGL(k_gn16_apply, dim3((HW + GN16_PIX - 1) / GN16_PIX, B), dim3(256), 0, (hipStream_t)stream,
                       (const vh_e16*)x, stats, gamma, beta, (vh_e16*)out, HW, C, G, silu);
    return vh_launch_status();
}

}  // namespace VH16_NS
