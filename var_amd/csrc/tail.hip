// tail.hip — the fp32 decoder's tail in one pass: norm_out -> swish -> conv_out -> clamp (-> (x + 1) / 2)
// (reference basic_vae.py:224-226 `conv_out(F.silu(norm_out(h)))`, the callers' clamp vqvae.py:63 and de-normalisation var.py:190).
//
// As two launches this pair moved the largest map of the decoder three times (GroupNorm apply: read + write 2 x 2.7 GB at B = 64; the
// convolution: read it again, nine times through L2) and ran a 3-channel convolution on a matrix tile 32 channels wide (10x the MACs).
// Here a workgroup of 256 threads owns an 8 x 32 patch of one image, ONE OUTPUT PIXEL PER THREAD.  Per 32-channel chunk the (8+2) x (32+2)
// halo patch goes global -> registers -> (((x - mean) * rstd) * gamma + beta, SiLU: k_gn_apply's operations in its order) -> LDS, pixels
// outside the image as zeros (the convolution pads the NORMALISED map), and every thread runs its own fma chains over the nine taps:
// with 3 output channels the vector ALU does the convolution at the algorithmic MAC count (the weights are wave-uniform: scalar loads).
// Summation order = k_dma_gemm's for convolutions (and the oracle's conv3x3_core): 32-channel chunk outermost, then tap, then channel, one
// k-ascending fp32 fma chain per output from 0 — the result is bit-identical to varhip_gn_apply_f32 + varhip_conv3x3_nhwc_f32 (tests).
// A chunk of a pixel is one 128-byte line (640-byte records): every input line is requested once per patch that touches it.
#include "common.h"

struct TailP {
    const float* in; const float* stats; const float* gamma; const float* beta; const float* w; const float* bias; float* out;
    int H, Wd, Cin, G, out_mode;
};

__device__ __forceinline__ float tail_silu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y)); }   // = gn_fast_silu (rowops.hip)

template <int NOUT>
__global__ void __launch_bounds__(256) k_gnconv32o(TailP p) {
    constexpr int PH = 8, PW = 32, P = PW + 2, PROWS = (PH + 2) * P, ROWF = 36;      // LDS row: 32 floats + 4 (16 consecutive rows on 16 different bank quads)
    constexpr int NPC = (PROWS * 8 + 255) / 256;                                        // 16-byte pieces of a patch chunk per thread
    extern __shared__ __attribute__((aligned(16))) float smt[];
    float* const sPatch = smt;                                      // [PROWS][ROWF]
    float* const sTab = smt + PROWS * ROWF;                         // [4][Cin]: mean, rstd, gamma, beta per channel
    const int tid = threadIdx.x;
    const int tX = p.Wd / PW, tY = p.H / PH, tps = tX * tY;
    const int b = blockIdx.x / tps, trem = blockIdx.x - b * tps, tyi = trem / tX, ty0 = tyi * PH, tx0 = (trem - tyi * tX) * PW;
    const int hw = p.H * p.Wd, K = 9 * p.Cin, nch = p.Cin / 32, cpg = p.Cin / p.G;
    for (int c = tid; c < p.Cin; c += 256) {
        const float* st = p.stats + ((int64_t)b * p.G + c / cpg) * 2;
        sTab[c] = st[0]; sTab[p.Cin + c] = st[1]; sTab[2 * p.Cin + c] = p.gamma[c]; sTab[3 * p.Cin + c] = p.beta[c];
    }
    // this thread's pieces: piece e = tid + 256 k -> patch row e >> 3, 4-channel slot e & 7 (= tid & 7 for every k)
    const int slot = tid & 7;
    uint32_t goff[NPC]; uint32_t okmask = 0, inmask = 0;
    const float* const src = p.in + (int64_t)b * hw * p.Cin + slot * 4;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int pr = (tid + 256 * k) >> 3, py = pr / P, px = pr - py * P, y = ty0 - 1 + py, x = tx0 - 1 + px;
        const bool in_patch = pr < PROWS, ok = in_patch && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
        goff[k] = ok ? (uint32_t)((y * p.Wd + x) * p.Cin) : 0u;
        okmask |= (ok ? 1u : 0u) << k; inmask |= (in_patch ? 1u : 0u) << k;
    }
    f32x4 raw[NPC];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) raw[k] = *(const f32x4*)(src + goff[k] + c * 32);      // (pieces past the patch / outside the image: offset 0, never parked / parked as zeros)
    };
    auto park = [&](int c) {
        const int ch = c * 32 + slot * 4;
        const f32x4 mean = *(const f32x4*)(sTab + ch), rstd = *(const f32x4*)(sTab + p.Cin + ch);
        const f32x4 g4 = *(const f32x4*)(sTab + 2 * p.Cin + ch), b4 = *(const f32x4*)(sTab + 3 * p.Cin + ch);
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            if (!((inmask >> k) & 1u)) continue;
            const float okf = ((okmask >> k) & 1u) ? 1.0f : 0.0f;        // (a multiply, not a select: hipcc turns the select into a branch around every element)
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = ((raw[k][e] - mean[e]) * rstd[e]) * g4[e] + b4[e];
                o[e] = tail_silu(y) * okf;
            }
            *(f32x4*)(sPatch + ((tid + 256 * k) >> 3) * ROWF + slot * 4) = o;
        }
    };
    const int ty = tid >> 5, tx = tid & 31;                         // this thread's output pixel inside the patch
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.0f;
    fetch(0);
    __syncthreads();                                                // the table is in LDS
    park(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) fetch(c + 1);                              // in flight while the taps of chunk c run
        // (the tap and half-chunk loops stay rolled: unrolled, the 864 wave-uniform weights of a chunk are hoisted into more scalar
        // registers than there are and come back through v_readlane; 48 per iteration fit, fetched as whole 64-byte vectors)
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const float* xr = sPatch + ((ty + ky) * P + tx + kx) * ROWF;
            const float* wt = p.w + t * p.Cin + c * 32;             // wave-uniform: scalar loads
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
                f32x16 wv[NOUT];                                    // 16 channels of every output channel's weights: one s_load_dwordx16 each
#pragma unroll
                for (int n = 0; n < NOUT; ++n) wv[n] = *(const f32x16*)(wt + (int64_t)n * K + hh * 16);
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f32x4 xv = *(const f32x4*)(xr + hh * 16 + j4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int n = 0; n < NOUT; ++n) acc[n] = __builtin_fmaf(xv[e], wv[n][j4 * 4 + e], acc[n]);
                }
            }
        }
        __syncthreads();                                            // every thread is done reading chunk c
        if (c + 1 < nch) { park(c + 1); __syncthreads(); }
    }
    const int y = ty0 + ty, x = tx0 + tx;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const float v = vm_min(vm_max(acc[n] + p.bias[n], -1.0f), 1.0f);
        p.out[((int64_t)b * NOUT + n) * hw + y * p.Wd + x] = p.out_mode == 1 ? (v + 1.0f) * 0.5f : v;
    }
}

template <int NOUT> static int launch_tail(const TailP& p, int B, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_gnconv32o<NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_done = true; }
    hipLaunchKernelGGL(k_gnconv32o<NOUT>, dim3(B * (p.H / 8) * (p.Wd / 32)), dim3(256), lds, s, p);
    return vh_launch_status();
}

// out = clamp(conv3x3(SiLU(GroupNorm(x))) + bias, -1, 1) as fp32 NCHW (out_mode 2), de-normalised to [0, 1] (out_mode 1): x [B][H][W][Cin]
// channels-last, stats [B][G][2] = (mean, rstd) as varhip_gn_stats_f32 / varhip_gn_stats_part_f32 leave them, w [Cout][3][3][Cin].
// Takes maps that tile into 8 x 32 patches with Cin % 32 == 0 and Cout <= 4; anything else: VARHIP_EINVAL (the caller then runs
// varhip_gn_apply_f32 + varhip_conv3x3_nhwc_f32, which this call equals bit for bit).
extern "C" int varhip_gn_silu_conv_out_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                                           float* out, int B, int H, int W, int Cin, int Cout, int G, int out_mode, varhip_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || !x || !stats || !gamma || !beta || !w || !bias || !out) return VARHIP_EINVAL;
    if ((H % 8) || (W % 32) || (Cin % 32) || (Cin % G) || Cout > 4 || (out_mode != 1 && out_mode != 2)) return VARHIP_EINVAL;
    if ((((uintptr_t)x) & 15) || (int64_t)H * W * Cin >= (1ll << 31) || (int64_t)B * (H / 8) * (W / 32) >= (1ll << 31)) return VARHIP_EINVAL;
    const size_t lds = ((size_t)10 * 34 * 36 + (size_t)4 * Cin) * sizeof(float);
    if (lds > 64 * 1024) return VARHIP_EINVAL;
    TailP p{x, stats, gamma, beta, w, bias, out, H, W, Cin, G, out_mode};
    const double npix = (double)B * H * W;
    VhScope scope(VH_FAM_CONV_SMALL, (hipStream_t)stream, 2.0 * npix * Cout * 9.0 * Cin, 4.0 * npix * Cin + 4.0 * npix * Cout + 36.0 * Cin * Cout);
    switch (Cout) {
        case 1: return launch_tail<1>(p, B, lds, (hipStream_t)stream);
        case 2: return launch_tail<2>(p, B, lds, (hipStream_t)stream);
        case 3: return launch_tail<3>(p, B, lds, (hipStream_t)stream);
        default: return launch_tail<4>(p, B, lds, (hipStream_t)stream);
    }
}
