// block.hip — one AdaLNSelfAttn block (reference basic_var.py:152-159) as ONE library call: the seven launches
//   LN+modulate -> q/k/v GEMM (+ norm, scale, cache append) -> attention -> proj (+ gamma1, residual)
//   -> LN+modulate -> fc1 (+ GELU) -> fc2 (+ gamma2, residual)
// are issued back to back from C++.  No new arithmetic: every step is the entry point of the same name in include/var_hip.h.
// Why it exists: at the small scales (l <= 16) a launch lasts 5-20 us on the GPU while one ctypes call from Python costs about
// 10 us on the host, so a per-kernel Python loop leaves the GPU idle; 16 composite calls per scale instead of 112 keep it fed.
#include "common.h"

extern "C" int varhip_adaln_block_f32(float* x, float* x2, float* xn, float* q, float* att, float* hid,
                                      const float* ada, int64_t ld_ada,
                                      const float* qkv_w, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                                      const float* proj_w, const float* proj_b, const float* fc1_w, const float* fc1_b,
                                      const float* fc2_w, const float* fc2_b, float* kcache, float* vcache,
                                      int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || C <= 0 || H <= 0 || C != H * 64 || hidden <= 0 || !ada) return VARHIP_EINVAL;
    const int M = B2 * l;
    // ada row b: [gamma1 | gamma2 | scale1 | scale2 | shift1 | shift2], each C wide (basic_var.py:156: unbind(2) of (B,1,6,C))
    const float *g1 = ada, *g2 = ada + C, *s1 = ada + 2 * C, *s2 = ada + 3 * C, *h1 = ada + 4 * C, *h2 = ada + 5 * C;
    int rc;
    if ((rc = varhip_ln_modulate_f32(x, s1, ld_ada, h1, ld_ada, xn, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_qkv_f32(xn, C, qkv_w, C, qkv_b, M, C, C, scale_mul, plain_scale, l2norm, q, kcache, vcache, B2, l, H, pos0, Lmax, stream))) return rc;
    if ((rc = varhip_attn_cached_f32(q, kcache, vcache, att, B2, l, H, pos0 + l, Lmax, stream))) return rc;
    if ((rc = varhip_gemm_nt_f32(att, C, proj_w, C, proj_b, x2, C, M, C, C, VARHIP_EPI_RESID, x, C, g1, ld_ada, l, 0, 1, 0, 0, 0, stream))) return rc;
    if ((rc = varhip_ln_modulate_f32(x2, s2, ld_ada, h2, ld_ada, xn, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_nt_f32(xn, C, fc1_w, C, fc1_b, hid, hidden, M, hidden, C, VARHIP_EPI_GELU, nullptr, 0, nullptr, 0, 1, 0, 1, 0, 0, 0, stream))) return rc;
    return varhip_gemm_nt_f32(hid, hidden, fc2_w, hidden, fc2_b, x, C, M, C, hidden, VARHIP_EPI_RESID, x2, C, g2, ld_ada, l, 0, 1, 0, 0, 0, stream);
}

// The same block in the 16-bit throughput mode (gemm16.hip, attn16.hip): the residual stream x / x2 and the AdaLN parameters stay fp32,
// the GEMM operands (LayerNorm output, q, attention output, MLP hidden), the weights and the KV cache are fp16.
extern "C" int varhip_adaln_block_f16(float* x, float* x2, void* xn16, void* q16, void* att16, void* hid16,
                                      const float* ada, int64_t ld_ada,
                                      const void* qkv_w16, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                                      const void* proj_w16, const float* proj_b, const void* fc1_w16, const float* fc1_b,
                                      const void* fc2_w16, const float* fc2_b, void* kcache16, void* vcache16,
                                      int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || C <= 0 || H <= 0 || C != H * 64 || hidden <= 0 || !ada) return VARHIP_EINVAL;
    const int M = B2 * l;
    const float *g1 = ada, *g2 = ada + C, *s1 = ada + 2 * C, *s2 = ada + 3 * C, *h1 = ada + 4 * C, *h2 = ada + 5 * C;
    int rc;
    if ((rc = varhip_ln_modulate_f16out(x, s1, ld_ada, h1, ld_ada, xn16, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_qkv_f16(xn16, C, qkv_w16, C, qkv_b, M, C, C, scale_mul, plain_scale, l2norm, q16, kcache16, vcache16, B2, l, H, pos0, Lmax, stream))) return rc;
    if ((rc = varhip_attn_cached_f16(q16, kcache16, vcache16, att16, B2, l, H, pos0 + l, Lmax, stream))) return rc;
    if ((rc = varhip_gemm_nt_f16(att16, C, proj_w16, C, proj_b, x2, C, 0, M, C, C, VARHIP_EPI_RESID, x, C, 0, g1, ld_ada, l, 1, 0, 0, 0, stream))) return rc;
    if ((rc = varhip_ln_modulate_f16out(x2, s2, ld_ada, h2, ld_ada, xn16, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_nt_f16(xn16, C, fc1_w16, C, fc1_b, hid16, hidden, 1, M, hidden, C, VARHIP_EPI_GELU, nullptr, 0, 0, nullptr, 0, 1, 1, 0, 0, 0, stream))) return rc;
    return varhip_gemm_nt_f16(hid16, hidden, fc2_w16, hidden, fc2_b, x, C, 0, M, C, hidden, VARHIP_EPI_RESID, x2, C, 0, g2, ld_ada, l, 1, 0, 0, 0, stream);
}

// ... and with bfloat16 operands (the -DVH_BF16 builds of the same kernels)
extern "C" int varhip_adaln_block_bf16(float* x, float* x2, void* xn16, void* q16, void* att16, void* hid16,
                                      const float* ada, int64_t ld_ada,
                                      const void* qkv_w16, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                                      const void* proj_w16, const float* proj_b, const void* fc1_w16, const float* fc1_b,
                                      const void* fc2_w16, const float* fc2_b, void* kcache16, void* vcache16,
                                      int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream) {
    if (B2 <= 0 || l <= 0 || C <= 0 || H <= 0 || C != H * 64 || hidden <= 0 || !ada) return VARHIP_EINVAL;
    const int M = B2 * l;
    const float *g1 = ada, *g2 = ada + C, *s1 = ada + 2 * C, *s2 = ada + 3 * C, *h1 = ada + 4 * C, *h2 = ada + 5 * C;
    int rc;
    if ((rc = varhip_ln_modulate_bf16out(x, s1, ld_ada, h1, ld_ada, xn16, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_qkv_bf16(xn16, C, qkv_w16, C, qkv_b, M, C, C, scale_mul, plain_scale, l2norm, q16, kcache16, vcache16, B2, l, H, pos0, Lmax, stream))) return rc;
    if ((rc = varhip_attn_cached_bf16(q16, kcache16, vcache16, att16, B2, l, H, pos0 + l, Lmax, stream))) return rc;
    if ((rc = varhip_gemm_nt_bf16(att16, C, proj_w16, C, proj_b, x2, C, 0, M, C, C, VARHIP_EPI_RESID, x, C, 0, g1, ld_ada, l, 1, 0, 0, 0, stream))) return rc;
    if ((rc = varhip_ln_modulate_bf16out(x2, s2, ld_ada, h2, ld_ada, xn16, M, C, l, eps, stream))) return rc;
    if ((rc = varhip_gemm_nt_bf16(xn16, C, fc1_w16, C, fc1_b, hid16, hidden, 1, M, hidden, C, VARHIP_EPI_GELU, nullptr, 0, 0, nullptr, 0, 1, 1, 0, 0, 0, stream))) return rc;
    return varhip_gemm_nt_bf16(hid16, hidden, fc2_w16, hidden, fc2_b, x, C, 0, M, C, hidden, VARHIP_EPI_RESID, x2, C, 0, g2, ld_ada, l, 1, 0, 0, 0, stream);
}
