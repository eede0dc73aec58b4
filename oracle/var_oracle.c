/* var_oracle.c — CPU oracle for the VAR next-scale sampling path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (var_amd/, models/) never does.  It restates, in plain scalar C, the algorithm of the reference
 * (culiver/VAR, all Python/PyTorch) for the path VAR.autoregressive_infer_cfg -> VectorQuantizer2 -> VQVAE decode;
 * every function cites the reference lines it follows.  The arithmetic PyTorch delegates to ATen (oneDNN sgemm,
 * vectorised reductions, Sleef exp) has no specified rounding order; this oracle fixes one:
 *   - dot products: fp32 fma chain, k ascending, from 0;   - exp & friends: include/var_math.h;
 *   - row reductions: the "canonical" lane-strided + butterfly orders defined below.
 * Parity pinning: tests/test_oracle_vs_golden.py checks this oracle against fixtures produced by running the
 * reference itself (tools/gen_golden.py): token ids equal, logits/pixels to fp32 rounding noise.
 *
 * Signatures mirror include/var_hip.h (prefix varref_, host pointers, no stream).
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -mavx2 -mfma -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/var_math.h"

#define VARHIP_EINVAL (-1)
#define EPI_NONE 0
#define EPI_GELU 1
#define EPI_RESID 2

/* ---- canonical reductions ---------------------------------------------------------------------------------------
 * W64(vw): element i goes to partial (i / vw) % 64, partials accumulate in ascending i from 0.0f, then an xor
 * butterfly (offsets 32,16,8,4,2,1) — exactly what one 64-lane wavefront does with vw-wide loads and __shfl_xor. */
static float canon_sum64(const float* x, int n, int vw) {
    float p[64], q[64];
    for (int j = 0; j < 64; ++j) p[j] = 0.0f;
    for (int i = 0; i < n; ++i) { int j = (i / vw) & 63; p[j] = p[j] + x[i]; }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int j = 0; j < 64; ++j) q[j] = p[j] + p[j ^ off];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}
/* W256: 256 threads, thread t sums i = t, t+256, ...; butterfly inside each 64-lane wave; waves added in order. */
static float canon_sum256(const float* x, int n) {
    float p[256], q[64];
    for (int j = 0; j < 256; ++j) p[j] = 0.0f;
    for (int i = 0; i < n; ++i) p[i & 255] = p[i & 255] + x[i];
    float w[4];
    for (int wv = 0; wv < 4; ++wv) {
        float* pw = p + 64 * wv;
        for (int off = 32; off >= 1; off >>= 1) {
            for (int j = 0; j < 64; ++j) q[j] = pw[j] + pw[j ^ off];
            memcpy(pw, q, sizeof(q));
        }
        w[wv] = pw[0];
    }
    return ((w[0] + w[1]) + w[2]) + w[3];
}

const char* varref_version(void) { return "var_oracle 0.1 (cpu restatement)"; }

/* OpenMP team size used by every function below (bench.py reports it as cpu_baseline.cores) */
#ifdef _OPENMP
#include <omp.h>
int varref_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int varref_set_threads(int n) { (void)n; return 1; }
#endif

/* ================================================================================================================
 * GEMM  out = epi(A . W^T + bias)        reference: F.linear at basic_var.py:93,119,52,147,170; var.py:124;
 * 1x1 convs and bmm's of basic_vae.py:53,69,71,83,89.   Wt is W transposed to [K][N] (so the n loop vectorises;
 * each out[m][n] is still its own k-ascending fma chain). */
#include <immintrin.h>
/* register-blocked inner kernel: MR rows x 16 columns of out held in ymm registers over the whole k loop.  Every element is still its own
 * k-ascending chain acc = fma(a[m][k], w[k][n], acc) from 0 (one _mm256_fmadd_ps lane each) — the blocking only changes how often a
 * weight row is re-read (once per MR rows instead of once per row), not one bit of any result. */
#define GEMM_MR 6
static inline void gemm_micro_6x16(const float* const a[GEMM_MR], const float* wt, int64_t ldwt, int K, float acc[GEMM_MR][16]) {
    __m256 c[GEMM_MR][2];
    for (int r = 0; r < GEMM_MR; ++r) { c[r][0] = _mm256_setzero_ps(); c[r][1] = _mm256_setzero_ps(); }
    for (int k = 0; k < K; ++k) {
        const __m256 b0 = _mm256_loadu_ps(wt + (int64_t)k * ldwt), b1 = _mm256_loadu_ps(wt + (int64_t)k * ldwt + 8);
        for (int r = 0; r < GEMM_MR; ++r) {
            const __m256 av = _mm256_broadcast_ss(a[r] + k);
            c[r][0] = _mm256_fmadd_ps(av, b0, c[r][0]);
            c[r][1] = _mm256_fmadd_ps(av, b1, c[r][1]);
        }
    }
    for (int r = 0; r < GEMM_MR; ++r) { _mm256_storeu_ps(acc[r], c[r][0]); _mm256_storeu_ps(acc[r] + 8, c[r][1]); }
}

static void gemm_kn_core(const float* A, int64_t lda, const float* Wt, int64_t ldwt, const float* bias, float* out, int64_t ldo,
                         int M, int N, int K, int epi, const float* resid, int64_t ldr, const float* gamma, int64_t ldg,
                         int rows_per_group, int bias_per_row) {
    const int NB = 16;
    const int nblk = (N + NB - 1) / NB, mblk = (M + GEMM_MR - 1) / GEMM_MR;
    /* one task = one 16-column panel of Wt, copied once into a contiguous K x 16 buffer (its rows lie N floats apart: at N = 3072 every
     * row of the panel lands in the same cache sets) and used for every row block; with few panels (N < 512) the (panel, row block) pairs
     * are the tasks and the panel is read in place */
    const int pack = nblk >= 32;
#pragma omp parallel
    {
        float* panel = pack ? (float*)aligned_alloc(64, sizeof(float) * (size_t)K * NB) : NULL;
        const int64_t ntask = pack ? nblk : (int64_t)nblk * mblk;
#pragma omp for schedule(dynamic, 1)
        for (int64_t task = 0; task < ntask; ++task) {
            const int nb = pack ? (int)task : (int)(task / mblk), mb_lo = pack ? 0 : (int)(task % mblk), mb_hi = pack ? mblk : mb_lo + 1;
            const int n0 = nb * NB, w = (n0 + NB < N ? n0 + NB : N) - n0;
            const float* wp = Wt + n0; int64_t ldp = ldwt;
            if (pack && w == NB) {
                for (int k = 0; k < K; ++k) { _mm256_store_ps(panel + (int64_t)k * NB, _mm256_loadu_ps(Wt + (int64_t)k * ldwt + n0));
                                              _mm256_store_ps(panel + (int64_t)k * NB + 8, _mm256_loadu_ps(Wt + (int64_t)k * ldwt + n0 + 8)); }
                wp = panel; ldp = NB;
            }
            for (int mb = mb_lo; mb < mb_hi; ++mb) {
                const int m0 = mb * GEMM_MR, mr = (m0 + GEMM_MR < M ? m0 + GEMM_MR : M) - m0;
                float acc[GEMM_MR][16];
                if (w == NB) {
                    const float* a[GEMM_MR];
                    for (int r = 0; r < GEMM_MR; ++r) a[r] = A + (int64_t)(m0 + (r < mr ? r : 0)) * lda;      /* (rows past M: a repeat, discarded) */
                    gemm_micro_6x16(a, wp, ldp, K, acc);
                } else {                                             /* ragged last column block: the plain loops, same chains */
                    for (int r = 0; r < mr; ++r) {
                        for (int j = 0; j < w; ++j) acc[r][j] = 0.0f;
                        const float* a = A + (int64_t)(m0 + r) * lda;
                        for (int k = 0; k < K; ++k) {
                            const float av = a[k];
                            const float* wr = Wt + (int64_t)k * ldwt + n0;
                            for (int j = 0; j < w; ++j) acc[r][j] = vm_fma(av, wr[j], acc[r][j]);
                        }
                    }
                }
                for (int r = 0; r < mr; ++r) {
                    const int m = m0 + r;
                    float* o = out + (int64_t)m * ldo + n0;
                    for (int j = 0; j < w; ++j) {
                        float v = acc[r][j];
                        if (bias) v = v + (bias_per_row ? bias[m] : bias[n0 + j]);
                        if (epi == EPI_GELU) v = vm_gelu_tanh(v);
                        else if (epi == EPI_RESID) {
                            if (gamma) v = v * gamma[(int64_t)(m / rows_per_group) * ldg + n0 + j];
                            v = resid[(int64_t)m * ldr + n0 + j] + v;
                        }
                        o[j] = v;
                    }
                }
            }
        }
        free(panel);
    }
}

static float* transpose_nk(const float* W, int64_t ldw, int N, int K) {
    float* t = (float*)malloc(sizeof(float) * (size_t)N * K);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) t[(int64_t)k * N + n] = W[(int64_t)n * ldw + k];
    return t;
}

int varref_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                       float* out, int64_t ldo, int M, int N, int K, int epi,
                       const float* resid, int64_t ldr, const float* gamma, int64_t ldg, int rows_per_group,
                       int bias_per_row, int batch, int64_t sA, int64_t sW, int64_t sO) {
    if (M < 0 || N <= 0 || K <= 0 || batch < 1) return VARHIP_EINVAL;
    if (batch > 1 && (resid || gamma)) return VARHIP_EINVAL;
    if (epi == EPI_RESID && !resid) return VARHIP_EINVAL;
    if (rows_per_group <= 0) rows_per_group = 1;
    float* wt = NULL;
    for (int b = 0; b < batch; ++b) {
        if (b == 0 || sW != 0) { free(wt); wt = transpose_nk(W + b * sW, ldw, N, K); }
        gemm_kn_core(A + b * sA, lda, wt, N, bias, out + b * sO, ldo, M, N, K, epi, resid, ldr, gamma, ldg, rows_per_group, bias_per_row);
    }
    free(wt);
    return 0;
}

/* same, with the weight already transposed to [K][N] (the oracle driver caches transposes; not part of the HIP ABI) */
int varref_gemm_kn_f32(const float* A, int64_t lda, const float* Wt, int64_t ldwt, const float* bias,
                       float* out, int64_t ldo, int M, int N, int K, int epi,
                       const float* resid, int64_t ldr, const float* gamma, int64_t ldg, int rows_per_group) {
    if (rows_per_group <= 0) rows_per_group = 1;
    gemm_kn_core(A, lda, Wt, ldwt, bias, out, ldo, M, N, K, epi, resid, ldr, gamma, ldg, rows_per_group, 0);
    return 0;
}

/* nn.SiLU in front of ada_lin (basic_var.py:147,170; var.py:80) */
int varref_silu_f32(const float* x, float* y, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = vm_silu(x[i]);
    return 0;
}

/* shared AdaLN: ada_gss + cond_BD (basic_var.py:153-154) */
int varref_add_bcast_f32(const float* base, const float* cond, float* out, int rows, int n) {
    for (int r = 0; r < rows; ++r)
        for (int j = 0; j < n; ++j) out[(int64_t)r * n + j] = base[j] + cond[(int64_t)r * n + j];
    return 0;
}

/* ln_wo_grad(x).mul(scale.add(1)).add_(shift)   (basic_var.py:157,158,174; LayerNorm eps 1e-6, no affine: var.py:82, basic_var.py:141) */
int varref_ln_modulate_f32(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                           float* out, int M, int C, int rows_per_group, float eps) {
    if (C <= 0 || rows_per_group <= 0) return VARHIP_EINVAL;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* xr = x + (int64_t)m * C;
        float* d = (float*)malloc(sizeof(float) * 2 * C);
        float* d2 = d + C;
        float mean = canon_sum64(xr, C, 4) / (float)C;
        for (int i = 0; i < C; ++i) { d[i] = xr[i] - mean; d2[i] = d[i] * d[i]; }
        float var = canon_sum64(d2, C, 4) / (float)C;
        float rstd = 1.0f / vm_sqrt(var + eps);
        const float* sc = scale + (int64_t)(m / rows_per_group) * ld_scale;
        const float* sh = shift + (int64_t)(m / rows_per_group) * ld_shift;
        for (int i = 0; i < C; ++i) out[(int64_t)m * C + i] = (d[i] * rstd) * (sc[i] + 1.0f) + sh[i];
        free(d);
    }
    return 0;
}

/* SelfAttention.forward up to the cache append (basic_var.py:98-109), head_dim 64 */
int varref_qkv_prep_f32(const float* qkv, const float* scale_mul, float plain_scale, int l2norm,
                        float* q_out, float* kcache, float* vcache, int B2, int l, int H, int pos0, int Lmax) {
    if (pos0 < 0 || pos0 + l > Lmax) return VARHIP_EINVAL;
    const int C = H * 64;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B2; ++b) {
        for (int t = 0; t < l; ++t) {
            const float* row = qkv + ((int64_t)b * l + t) * 3 * C;
            for (int h = 0; h < H; ++h) {
                const float* q = row + h * 64; const float* k = row + C + h * 64; const float* v = row + 2 * C + h * 64;
                float* qo = q_out + ((int64_t)b * l + t) * C + h * 64;
                float* ko = kcache + (((int64_t)b * H + h) * Lmax + pos0 + t) * 64;
                float* vo = vcache + (((int64_t)b * H + h) * Lmax + pos0 + t) * 64;
                if (l2norm) {
                    float sq[64];
                    for (int c = 0; c < 64; ++c) sq[c] = q[c] * q[c];
                    float dq = vm_max(vm_sqrt(canon_sum64(sq, 64, 1)), 1e-12f);       /* F.normalize eps */
                    for (int c = 0; c < 64; ++c) sq[c] = k[c] * k[c];
                    float dk = vm_max(vm_sqrt(canon_sum64(sq, 64, 1)), 1e-12f);
                    float sm = vm_exp(vm_min(scale_mul[h], 4.605170249938965f));     /* clamp_max(log 100).exp(): basic_var.py:70,102 */
                    for (int c = 0; c < 64; ++c) { qo[c] = (q[c] / dq) * sm; ko[c] = k[c] / dk; vo[c] = v[c]; }
                } else {
                    for (int c = 0; c < 64; ++c) { qo[c] = q[c] * plain_scale; ko[c] = k[c]; vo[c] = v[c]; }
                }
            }
        }
    }
    return 0;
}

/* mat_qkv + the post-processing above as one call (basic_var.py:93-109): twin of the fused-epilogue HIP entry point,
 * stated as the two steps it is defined to equal. */
int varref_gemm_qkv_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int M, int C, int K,
                        const float* scale_mul, float plain_scale, int l2norm,
                        float* q_out, float* kcache, float* vcache, int B2, int l, int H, int pos0, int Lmax) {
    if (C != H * 64 || M != B2 * l || K <= 0) return VARHIP_EINVAL;
    float* qkv = (float*)malloc(sizeof(float) * (size_t)M * 3 * C);
    int rc = varref_gemm_nt_f32(A, lda, W, ldw, bias, qkv, 3 * C, M, 3 * C, K, EPI_NONE, NULL, 0, NULL, 0, 1, 0, 1, 0, 0, 0);
    if (!rc) rc = varref_qkv_prep_f32(qkv, scale_mul, plain_scale, l2norm, q_out, kcache, vcache, B2, l, H, pos0, Lmax);
    free(qkv);
    return rc;
}

/* forward declarations for the composite below */
int varref_attn_cached_f32(const float* q, const float* kcache, const float* vcache, float* out, int B2, int l, int H, int curL, int Lmax);

/* AdaLNSelfAttn.forward (basic_var.py:152-159) as the seven steps above, in order: twin of varhip_adaln_block_f32 */
int varref_adaln_block_f32(float* x, float* x2, float* xn, float* q, float* att, float* hid, const float* ada, int64_t ld_ada,
                           const float* qkv_w, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                           const float* proj_w, const float* proj_b, const float* fc1_w, const float* fc1_b,
                           const float* fc2_w, const float* fc2_b, float* kcache, float* vcache,
                           int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps) {
    if (B2 <= 0 || l <= 0 || C != H * 64 || hidden <= 0 || !ada) return VARHIP_EINVAL;
    const int M = B2 * l;
    const float *g1 = ada, *g2 = ada + C, *s1 = ada + 2 * C, *s2 = ada + 3 * C, *h1 = ada + 4 * C, *h2 = ada + 5 * C;
    int rc;
    if ((rc = varref_ln_modulate_f32(x, s1, ld_ada, h1, ld_ada, xn, M, C, l, eps))) return rc;
    if ((rc = varref_gemm_qkv_f32(xn, C, qkv_w, C, qkv_b, M, C, C, scale_mul, plain_scale, l2norm, q, kcache, vcache, B2, l, H, pos0, Lmax))) return rc;
    if ((rc = varref_attn_cached_f32(q, kcache, vcache, att, B2, l, H, pos0 + l, Lmax))) return rc;
    if ((rc = varref_gemm_nt_f32(att, C, proj_w, C, proj_b, x2, C, M, C, C, EPI_RESID, x, C, g1, ld_ada, l, 0, 1, 0, 0, 0))) return rc;
    if ((rc = varref_ln_modulate_f32(x2, s2, ld_ada, h2, ld_ada, xn, M, C, l, eps))) return rc;
    if ((rc = varref_gemm_nt_f32(xn, C, fc1_w, C, fc1_b, hid, hidden, M, hidden, C, EPI_GELU, NULL, 0, NULL, 0, 1, 0, 1, 0, 0, 0))) return rc;
    return varref_gemm_nt_f32(hid, hidden, fc2_w, hidden, fc2_b, x, C, M, C, hidden, EPI_RESID, x2, C, g2, ld_ada, l, 0, 1, 0, 0, 0);
}

/* slow_attn / SDPA without mask over the cached keys (basic_var.py:111-117).
 * Summation orders (include/var_hip.h, varhip_attn_cached_f32): a dot product over k = 0..63 (q.k) or over the 32 keys of a tile
 * (p.v) is ONE fma chain in the "4-interleaved" order 0,4,1,5,2,6,3,7, 8,12,9,13, ... (inside every group of eight: j, j+4 for
 * j = 0..3).  Row sum of the softmax numerators: four partial sums S[h][x] over the keys with ((key >> 2) & 1) == h and
 * (key & 1) == x, each ascending; l = (S[0][0] + S[0][1]) + (S[1][0] + S[1][1]).  The exponential is vm_exp_le0
 * (include/var_math.h); the output is acc * (1 / l).  See DESIGN.md §Numerics. */
static inline int il4(int i) { return (i & ~7) + ((i & 1) << 2) + ((i & 7) >> 1); }      /* position i of the chain -> index */
int varref_attn_cached_f32(const float* q, const float* kcache, const float* vcache, float* out,
                           int B2, int l, int H, int curL, int Lmax) {
    if (curL <= 0 || curL > Lmax) return VARHIP_EINVAL;
    const int C = H * 64;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int b = 0; b < B2; ++b) {
        for (int h = 0; h < H; ++h) {
            const float* K = kcache + ((int64_t)b * H + h) * Lmax * 64;
            const float* Vv = vcache + ((int64_t)b * H + h) * Lmax * 64;
            /* The reference runs flash_attn_func / SDPA here; both are the running-max ("online softmax") recurrence with an
             * implementation-defined tile.  This restatement fixes the tile at 32 keys:
             *   m' = max(m, max_tile s); a = exp(m - m'); l = l*a + sum p; O = O*a + sum p v,  p = exp(s - m')
             * (a == 1 exactly when the maximum stands, so rescaling every tile and rescaling only when it moved are the same bits). */
            for (int t = 0; t < l; ++t) {
                const float* qr = q + ((int64_t)b * l + t) * C + h * 64;
                float m = -INFINITY, ls[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
                float acc[64], s[32], pj[32];
                for (int c = 0; c < 64; ++c) acc[c] = 0.0f;
                for (int j0 = 0; j0 < curL; j0 += 32) {
                    const int nj = curL - j0 < 32 ? curL - j0 : 32;
                    float tmax = -INFINITY;
                    for (int jj = 0; jj < nj; ++jj) {
                        float a = 0.0f;
                        for (int i = 0; i < 64; ++i) { const int d = il4(i); a = vm_fma(K[(int64_t)(j0 + jj) * 64 + d], qr[d], a); }
                        s[jj] = a; tmax = vm_max(tmax, a);
                    }
                    const float mnew = vm_max(m, tmax);
                    const float alpha = vm_exp_le0(m - mnew);
                    m = mnew;
                    ls[0][0] = ls[0][0] * alpha; ls[0][1] = ls[0][1] * alpha; ls[1][0] = ls[1][0] * alpha; ls[1][1] = ls[1][1] * alpha;
                    for (int c = 0; c < 64; ++c) acc[c] = acc[c] * alpha;
                    for (int jj = 0; jj < nj; ++jj) {                              /* row sums: ascending keys inside each class */
                        const int key = j0 + jj;
                        pj[jj] = vm_exp_le0(s[jj] - m);
                        ls[(key >> 2) & 1][key & 1] = ls[(key >> 2) & 1][key & 1] + pj[jj];
                    }
                    for (int i = 0; i < 32; ++i) {                                 /* p.v: 4-interleaved key order (tile starts are multiples of 32) */
                        const int jj = il4(i);
                        if (jj >= nj) continue;                                    /* keys past curL do not exist (the kernel adds p = 0 times v = 0) */
                        const float pv = pj[jj];
                        const float* vr = Vv + (int64_t)(j0 + jj) * 64;
#pragma omp simd
                        for (int c = 0; c < 64; ++c) acc[c] = vm_fma(pv, vr[c], acc[c]);
                    }
                }
                const float inv = 1.0f / ((ls[0][0] + ls[0][1]) + (ls[1][0] + ls[1][1]));
                float* o = out + ((int64_t)b * l + t) * C + h * 64;
                for (int c = 0; c < 64; ++c) o[c] = acc[c] * inv;
            }
        }
    }
    return 0;
}

/* Twin of varhip_attn_cached_f16 (the 16-bit throughput mode, include/var_hip.h): q, k, v hold fp16-representable values; scores, the
 * running maximum, p = exp(s - m) and the row sum are fp32; p is rounded to fp16 (round-to-nearest-even) for the p.v product only.
 * Plain ascending orders: this mode is compared with a tolerance, not bit for bit. */
#include <immintrin.h>
static inline float round_f16(float x) { return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT)); }
/* bfloat16 flavour of the same mode (include/var_hip.h "bf16"): round-to-nearest-even on the upper 16 bits of the fp32 pattern (finite inputs) */
static inline float round_bf16(float x) { uint32_t u; memcpy(&u, &x, 4); u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; memcpy(&x, &u, 4); return x; }
static int attn_cached_p16(const float* q, const float* kcache, const float* vcache, float* out,
                           int B2, int l, int H, int curL, int Lmax, int bf16) {
    if (curL <= 0 || curL > Lmax) return VARHIP_EINVAL;
    const int C = H * 64;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int b = 0; b < B2; ++b) {
        for (int h = 0; h < H; ++h) {
            const float* K = kcache + ((int64_t)b * H + h) * Lmax * 64;
            const float* Vv = vcache + ((int64_t)b * H + h) * Lmax * 64;
            for (int t = 0; t < l; ++t) {
                const float* qr = q + ((int64_t)b * l + t) * C + h * 64;
                float m = -INFINITY, ls = 0.0f;
                float acc[64], s[32];
                for (int c = 0; c < 64; ++c) acc[c] = 0.0f;
                for (int j0 = 0; j0 < curL; j0 += 32) {
                    const int nj = curL - j0 < 32 ? curL - j0 : 32;
                    float tmax = -INFINITY;
                    for (int jj = 0; jj < nj; ++jj) {
                        float a = 0.0f;
                        for (int d = 0; d < 64; ++d) a = vm_fma(K[(int64_t)(j0 + jj) * 64 + d], qr[d], a);
                        s[jj] = a; tmax = vm_max(tmax, a);
                    }
                    const float mnew = vm_max(m, tmax);
                    const float alpha = vm_exp(m - mnew);
                    m = mnew;
                    ls = ls * alpha;
                    for (int c = 0; c < 64; ++c) acc[c] = acc[c] * alpha;
                    for (int jj = 0; jj < nj; ++jj) {
                        const float pj = vm_exp(s[jj] - m);
                        ls = ls + pj;
                        const float p16 = bf16 ? round_bf16(pj) : round_f16(pj);
                        const float* vr = Vv + (int64_t)(j0 + jj) * 64;
#pragma omp simd
                        for (int c = 0; c < 64; ++c) acc[c] = vm_fma(p16, vr[c], acc[c]);
                    }
                }
                float* o = out + ((int64_t)b * l + t) * C + h * 64;
                for (int c = 0; c < 64; ++c) o[c] = bf16 ? round_bf16(acc[c] / ls) : round_f16(acc[c] / ls);
            }
        }
    }
    return 0;
}

int varref_attn_cached_p16_f32(const float* q, const float* kcache, const float* vcache, float* out, int B2, int l, int H, int curL, int Lmax) {
    return attn_cached_p16(q, kcache, vcache, out, B2, l, H, curL, Lmax, 0);
}
int varref_attn_cached_pbf16_f32(const float* q, const float* kcache, const float* vcache, float* out, int B2, int l, int H, int curL, int Lmax) {
    return attn_cached_p16(q, kcache, vcache, out, B2, l, H, curL, Lmax, 1);
}

/* CFG (var.py:172-173) + sample_with_top_k_top_p_ (helpers.py:6-19) + torch.multinomial(n=1) == argmax(p / Exp(1) noise) */
typedef struct { uint32_t key; int32_t idx; } sortent_t;
static int cmp_sortent(const void* a, const void* b) {
    const sortent_t* x = (const sortent_t*)a; const sortent_t* y = (const sortent_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
int varref_cfg_sample_f32(const float* logits, const float* noise, int64_t* idx_out, float* masked_out,
                          int B, int l, int V, double t_cfg, int top_k, double top_p) {
    if (V <= 0 || (V & 255) || V > 8192 || top_k < 0 || top_k > V) return VARHIP_EINVAL;
    const float ca = (float)(1.0 + t_cfg), cb = (float)t_cfg, thr = (float)(1.0 - top_p);
    int64_t rows = (int64_t)B * l;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t r = 0; r < rows; ++r) {
        const float* lc = logits + r * V; const float* lu = logits + (rows + r) * V;
        float* x = (float*)malloc(sizeof(float) * 2 * V); float* e = x + V;
        for (int i = 0; i < V; ++i) { float a = ca * lc[i]; float b = cb * lu[i]; x[i] = a - b; }
        if (top_k > 0) {                                   /* helpers.py:8-10: strict '<' keeps ties with the k-th value */
            uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * V);
            for (int i = 0; i < V; ++i) keys[i] = vm_float_key(x[i]);
            /* k-th largest by counting from the top: exact selection via full sort of a copy */
            sortent_t* tmp = (sortent_t*)malloc(sizeof(sortent_t) * V);
            for (int i = 0; i < V; ++i) { tmp[i].key = keys[i]; tmp[i].idx = i; }
            qsort(tmp, V, sizeof(sortent_t), cmp_sortent);
            float kth = x[tmp[V - top_k].idx];
            for (int i = 0; i < V; ++i) if (x[i] < kth) x[i] = -INFINITY;
            free(tmp); free(keys);
        }
        float m = -INFINITY;
        for (int i = 0; i < V; ++i) m = vm_max(m, x[i]);
        if (top_p > 0.0) {                                 /* helpers.py:11-15 */
            for (int i = 0; i < V; ++i) e[i] = vm_exp(x[i] - m);
            float S = canon_sum256(e, V);
            sortent_t* srt = (sortent_t*)malloc(sizeof(sortent_t) * V);
            for (int i = 0; i < V; ++i) { srt[i].key = vm_float_key(x[i]); srt[i].idx = i; }
            qsort(srt, V, sizeof(sortent_t), cmp_sortent);          /* ascending, stable by index */
            double c = 0.0;
            for (int s = 0; s < V - 1; ++s) {                         /* the last (largest) is never removed */
                int i = srt[s].idx;
                c += (double)(e[i] / S);
                if ((float)c <= thr) x[i] = -INFINITY; else break;    /* cum is non-decreasing: the removed set is a prefix */
            }
            free(srt);
        }
        for (int i = 0; i < V; ++i) e[i] = vm_exp(x[i] - m);
        float S = canon_sum256(e, V);
        const float* qn = noise + r * V;
        int best = 0; float bestv = 0.0f; int have = 0, best_nan = 0;
        for (int i = 0; i < V; ++i) {
            float rv = (e[i] / S) / qn[i];
            int isn = (rv != rv);
            if (!have) { best = i; bestv = rv; best_nan = isn; have = 1; }
            else if (!best_nan && (isn || rv > bestv)) { best = i; bestv = rv; best_nan = isn; }   /* torch.argmax: NaN wins, first max */
        }
        idx_out[r] = best;
        if (masked_out) memcpy(masked_out + r * V, x, sizeof(float) * V);
        free(x);
    }
    return 0;
}

/* codebook gather + bicubic up + Phi + f_hat accumulate  (var.py:177,182; quant.py:187-196, 199-206) */
static int quant_step_core(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                           const float* phi_w, const float* phi_b, float ratio,
                           float* up, float* f_hat, float* f_rest, int B, int pn, int P, int Cv) {
    if ((pn != P) && (!tap_idx || !tap_w)) return VARHIP_EINVAL;
    const float keep = 1.0f - ratio;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < P; ++y)
            for (int x = 0; x < P; ++x) {
                float* u = up + (((int64_t)b * P + y) * P + x) * Cv;
                if (pn == P) {
                    const float* e = codebook + idx[(int64_t)b * pn * pn + y * pn + x] * Cv;
                    for (int c = 0; c < Cv; ++c) u[c] = e[c];
                } else {
                    const int32_t* iy = tap_idx + y * 4; const float* wy = tap_w + y * 4;
                    const int32_t* ix = tap_idx + x * 4; const float* wx = tap_w + x * 4;
                    for (int c = 0; c < Cv; ++c) {
                        float rr[4];
                        for (int a = 0; a < 4; ++a) {
                            const int64_t* ir = idx + (int64_t)b * pn * pn + iy[a] * pn;
                            float acc = codebook[ir[ix[0]] * Cv + c] * wx[0];
                            acc = vm_fma(codebook[ir[ix[1]] * Cv + c], wx[1], acc);
                            acc = vm_fma(codebook[ir[ix[2]] * Cv + c], wx[2], acc);
                            acc = vm_fma(codebook[ir[ix[3]] * Cv + c], wx[3], acc);
                            rr[a] = acc;
                        }
                        float o = rr[0] * wy[0];
                        o = vm_fma(rr[1], wy[1], o); o = vm_fma(rr[2], wy[2], o); o = vm_fma(rr[3], wy[3], o);
                        u[c] = o;
                    }
                }
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < P; ++y)
            for (int x = 0; x < P; ++x) {
                const float* u0 = up + (((int64_t)b * P + y) * P + x) * Cv;
                float* f = f_hat + (((int64_t)b * P + y) * P + x) * Cv;
                for (int co = 0; co < Cv; ++co) {
                    float acc = 0.0f;
                    for (int ky = 0; ky < 3; ++ky) {
                        int yy = y + ky - 1; if (yy < 0 || yy >= P) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            int xx = x + kx - 1; if (xx < 0 || xx >= P) continue;
                            const float* u = up + (((int64_t)b * P + yy) * P + xx) * Cv;
                            const float* w = phi_w + (((int64_t)co * 3 + ky) * 3 + kx) * Cv;
                            for (int ci = 0; ci < Cv; ++ci) acc = vm_fma(u[ci], w[ci], acc);
                        }
                    }
                    float conv = acc + phi_b[co];
                    float hmix = u0[co] * keep + conv * ratio;       /* Phi.forward: h*(1-r) + conv(h)*r  (quant.py:205-206) */
                    f[co] = f[co] + hmix;                            /* f_hat.add_(h)  (quant.py:191,195) */
                    if (f_rest) { float* fr = f_rest + (((int64_t)b * P + y) * P + x) * Cv; fr[co] = fr[co] - hmix; }   /* f_rest.sub_(h) (quant.py:163) */
                }
            }
    return 0;
}
int varref_quant_accum_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                           const float* phi_w, const float* phi_b, float ratio,
                           float* up, float* f_hat, int B, int pn, int P, int Cv) {
    return quant_step_core(idx, codebook, tap_idx, tap_w, phi_w, phi_b, ratio, up, f_hat, NULL, B, pn, P, Cv);
}
/* one scale of the residual quantisation of an encoder feature map (quant.py:159-163): f_hat += h, f_rest -= h */
int varref_quant_residual_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                              const float* phi_w, const float* phi_b, float ratio, float* up, float* f_hat, float* f_rest,
                              int B, int pn, int P, int Cv) {
    if (!f_rest) return VARHIP_EINVAL;
    return quant_step_core(idx, codebook, tap_idx, tap_w, phi_w, phi_b, ratio, up, f_hat, f_rest, B, pn, P, Cv);
}
/* F.interpolate(f, (pq,pq), mode='area') == adaptive_avg_pool2d (quant.py:150,183), channels-last -> [B][pq*pq][Cv] */
int varref_area_pool_f32(const float* f, float* pooled, int B, int P, int pq, int Cv) {
    if (pq <= 0 || pq > P) return VARHIP_EINVAL;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < pq * pq; ++t) {
            int oy = t / pq, ox = t % pq;
            int y0 = (oy * P) / pq, y1 = ((oy + 1) * P + pq - 1) / pq, x0 = (ox * P) / pq, x1 = ((ox + 1) * P + pq - 1) / pq;
            for (int c = 0; c < Cv; ++c) {
                float s = 0.0f;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) s = s + f[(((int64_t)b * P + y) * P + x) * Cv + c];
                pooled[((int64_t)b * pq * pq + t) * Cv + c] = (s / (float)(y1 - y0)) / (float)(x1 - x0);
            }
        }
    return 0;
}
/* word_embed(pooled) + lvl_pos, rows duplicated for CFG (var.py:186-187; var.py:206-207 for teacher forcing) */
int varref_word_embed_f32(const float* pooled, const float* word_w, const float* word_b, const float* lvl_pos,
                          float* x_out, int B, int lq, int C, int Cv) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lq; ++t)
            for (int n = 0; n < C; ++n) {
                float acc = 0.0f;
                for (int c = 0; c < Cv; ++c) acc = vm_fma(pooled[((int64_t)b * lq + t) * Cv + c], word_w[(int64_t)n * Cv + c], acc);
                float v = (acc + word_b[n]) + lvl_pos[(int64_t)t * C + n];
                x_out[((int64_t)b * lq + t) * C + n] = v;
                x_out[((int64_t)(b + B) * lq + t) * C + n] = v;
            }
    return 0;
}

/* the same step fed with embeddings instead of token ids (more_smooth path, var.py:178-182): h [B][pn*pn][Cv] acts as a
 * per-call codebook addressed by the identity */
int varref_quant_accum_h_f32(const float* h, const int32_t* tap_idx, const float* tap_w, const float* phi_w, const float* phi_b, float ratio,
                             float* up, float* f_hat, int B, int pn, int P, int Cv) {
    int64_t n = (int64_t)B * pn * pn;
    int64_t* ident = (int64_t*)malloc(sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; ++i) ident[i] = i;
    int rc = varref_quant_accum_f32(ident, h, tap_idx, tap_w, phi_w, phi_b, ratio, up, f_hat, B, pn, P, Cv);
    free(ident);
    return rc;
}

/* gumbel_softmax_with_rng(logits * mul, tau, hard=False, rng) (helpers.py:22-36): gumbels = -log(Exp(1) noise);
 * y = softmax((x*mul + g) / tau) with the canonical W256 row sum */
int varref_gumbel_softmax_f32(const float* x, const float* noise, float* y, int64_t rows, int V, float mul, float tau) {
    if (V <= 0 || (V & 255)) return VARHIP_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float* z = (float*)malloc(sizeof(float) * V);
        float m = -INFINITY;
        for (int i = 0; i < V; ++i) {
            float g = -vm_log(noise[r * V + i]);
            z[i] = (x[r * V + i] * mul + g) / tau;
            m = vm_max(m, z[i]);
        }
        for (int i = 0; i < V; ++i) z[i] = vm_exp(z[i] - m);
        float S = canon_sum256(z, V);
        for (int i = 0; i < V; ++i) y[r * V + i] = z[i] / S;
        free(z);
    }
    return 0;
}

/* area-downsample to the next scale + word_embed + level/position embedding, duplicated for CFG (quant.py:192; var.py:185-187) */
int varref_next_map_f32(const float* f_hat, const float* word_w, const float* word_b, const float* lvl_pos,
                        float* x_out, float* pooled, int B, int P, int pq, int C, int Cv) {
    if (pq <= 0 || pq > P) return VARHIP_EINVAL;
    int lq = pq * pq;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lq; ++t) {
            int oy = t / pq, ox = t % pq;
            int y0 = (oy * P) / pq, y1 = ((oy + 1) * P + pq - 1) / pq;      /* adaptive_avg_pool2d windows == interpolate(mode='area') */
            int x0 = (ox * P) / pq, x1 = ((ox + 1) * P + pq - 1) / pq;
            float pl[64];
            for (int c = 0; c < Cv; ++c) {
                float s = 0.0f;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) s = s + f_hat[(((int64_t)b * P + y) * P + x) * Cv + c];
                pl[c] = (s / (float)(y1 - y0)) / (float)(x1 - x0);
                if (pooled) pooled[((int64_t)b * lq + t) * Cv + c] = pl[c];
            }
            for (int n = 0; n < C; ++n) {
                float acc = 0.0f;
                const float* w = word_w + (int64_t)n * Cv;
                for (int c = 0; c < Cv; ++c) acc = vm_fma(pl[c], w[c], acc);
                float v = (acc + word_b[n]) + lvl_pos[(int64_t)t * C + n];
                x_out[((int64_t)b * lq + t) * C + n] = v;
                x_out[((int64_t)(b + B) * lq + t) * C + n] = v;        /* .repeat(2,1,1) */
            }
        }
    return 0;
}

/* lvl_embed(lvl_1L) + pos_1LC  (var.py:153) */
int varref_lvl_pos_f32(const float* lvl_embed, const int64_t* lvl, const float* pos, float* out, int L, int C) {
    for (int t = 0; t < L; ++t)
        for (int n = 0; n < C; ++n) out[(int64_t)t * C + n] = lvl_embed[lvl[t] * C + n] + pos[(int64_t)t * C + n];
    return 0;
}

/* sos / cond_BD and the first token map (var.py:151,154) */
int varref_first_map_f32(const float* class_emb, const int64_t* labels, int num_classes, const float* pos_start,
                         const float* lvl_pos, float* cond, float* x_out, int B, int C, int first_l) {
    for (int b2 = 0; b2 < 2 * B; ++b2) {
        int64_t cls = b2 < B ? labels[b2] : num_classes;
        if (cls < 0 || cls > num_classes) return VARHIP_EINVAL;
        for (int n = 0; n < C; ++n) cond[(int64_t)b2 * C + n] = class_emb[cls * C + n];
        for (int t = 0; t < first_l; ++t)
            for (int n = 0; n < C; ++n)
                x_out[((int64_t)b2 * first_l + t) * C + n] = (cond[(int64_t)b2 * C + n] + pos_start[(int64_t)t * C + n]) + lvl_pos[(int64_t)t * C + n];
    }
    return 0;
}

/* Conv2d k=3 s=1 p=1 in channels-last form (basic_vae.py:25,48,51,180,208; vqvae.py:49).  wt is the weight
 * re-laid as [3][3][Cin][Cout] so the co loop vectorises; each out element is one (ky,kx,ci)-ascending fma chain. */
/* taps and source pixel of output pixel (y, x): returns 0 when tap (ky, kx) falls into the zero padding */
static inline int conv_src(int up2, int H, int W, int Hi, int Wi, int y, int x, int ky, int kx, int* sy, int* sx) {
    const int yy = up2 == 3 ? 2 * y + ky : y + ky - 1, xx = up2 == 3 ? 2 * x + kx : x + kx - 1;
    if (yy < 0 || yy >= (up2 == 3 ? Hi : H) || xx < 0 || xx >= (up2 == 3 ? Wi : W)) return 0;
    *sy = up2 == 1 ? yy >> 1 : yy; *sx = up2 == 1 ? xx >> 1 : xx;                    /* nearest 2x: src = dst // 2 */
    return 1;
}
static inline void conv_store(const float* acc, int co0, int co1, const float* bias, const float* resid, float* out,
                              int b, int y, int x, int H, int W, int Cout, int out_mode) {
    for (int co = co0; co < co1; ++co) {
        float v = acc[co - co0] + bias[co];
        if (resid) v = v + resid[(((int64_t)b * H + y) * W + x) * Cout + co];
        if (out_mode != 0) {
            v = vm_min(vm_max(v, -1.0f), 1.0f);                                   /* vqvae.py:63 clamp_(-1,1) */
            out[(((int64_t)b * Cout + co) * H + y) * W + x] = out_mode == 1 ? (v + 1.0f) * 0.5f : v;   /* var.py:190 add_(1).mul_(0.5) */
        } else out[(((int64_t)b * H + y) * W + x) * Cout + co] = v;
    }
}
/* one output pixel, output channels [co0, co1): the plain loops */
static void conv_pixel(const float* in, const float* wt, float* acc, int b, int y, int x, int H, int W, int Hi, int Wi, int Cin, int Cout,
                       int up2, int co0, int co1) {
    for (int co = co0; co < co1; ++co) acc[co - co0] = 0.0f;
    /* summation order of the contract (include/var_hip.h): channel chunks of 32 outermost, then the 9 taps, then the
     * channels of the chunk — all taps of a chunk touch the same few cache lines, which is what keeps the GPU kernel's
     * operand re-reads inside the L2 */
    for (int c0 = 0; c0 < Cin; c0 += 32) {
        const int c1 = c0 + 32 < Cin ? c0 + 32 : Cin;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                int sy, sx;
                if (!conv_src(up2, H, W, Hi, Wi, y, x, ky, kx, &sy, &sx)) continue;
                const float* ip = in + (((int64_t)b * Hi + sy) * Wi + sx) * Cin;
                const float* wp = wt + ((int64_t)(ky * 3 + kx) * Cin) * Cout;
                for (int ci = c0; ci < c1; ++ci) {
                    const float a = ip[ci];
                    const float* wr = wp + (int64_t)ci * Cout;
#pragma omp simd
                    for (int co = co0; co < co1; ++co) acc[co - co0] = vm_fma(a, wr[co], acc[co - co0]);
                }
            }
    }
}

/* Conv2d k=3 s=1 p=1 in channels-last form (basic_vae.py:25,48,51,180,208; vqvae.py:49).  wt is the weight
 * re-laid as [3][3][Cin][Cout] so the co loop vectorises; each out element is one (chunk, ky, kx, ci)-ascending fma chain.
 * Interior pixels (every tap's column inside the image) are computed six at a time x sixteen output channels in registers — the same
 * chains, each weight row read once per six pixels instead of once per pixel; the first and last pixel of a row take the plain loops. */
static void conv3x3_core(const float* in, const float* wt, const float* bias, const float* resid, float* out,
                         int B, int H, int W, int Cin, int Cout, int up2, int out_mode) {
    /* up2: 0 plain, 1 input read through a nearest 2x upsampling, 3 stride-2 over an input padded (0,1,0,1) (Downsample2x) */
    int Hi = up2 == 1 ? H / 2 : (up2 == 3 ? 2 * H : H), Wi = up2 == 1 ? W / 2 : (up2 == 3 ? 2 * W : W);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            float* acc = (float*)malloc(sizeof(float) * (Cout > 16 ? Cout : 16));
            for (int xe = 0; xe < 2; ++xe) {                               /* the two border pixels of the row */
                const int x = xe ? W - 1 : 0;
                if (xe && W == 1) break;
                conv_pixel(in, wt, acc, b, y, x, H, W, Hi, Wi, Cin, Cout, up2, 0, Cout);
                conv_store(acc, 0, Cout, bias, resid, out, b, y, x, H, W, Cout, out_mode);
            }
            for (int x0 = 1; x0 < W - 1; x0 += GEMM_MR) {
                const int npx = (x0 + GEMM_MR < W - 1 ? x0 + GEMM_MR : W - 1) - x0;
                for (int co0 = 0; co0 < Cout; co0 += 16) {
                    if (co0 + 16 > Cout) {                                 /* ragged last channel block: pixel by pixel */
                        for (int r = 0; r < npx; ++r) {
                            conv_pixel(in, wt, acc, b, y, x0 + r, H, W, Hi, Wi, Cin, Cout, up2, co0, Cout);
                            conv_store(acc, co0, Cout, bias, resid, out, b, y, x0 + r, H, W, Cout, out_mode);
                        }
                        continue;
                    }
                    __m256 c[GEMM_MR][2];
                    for (int r = 0; r < GEMM_MR; ++r) { c[r][0] = _mm256_setzero_ps(); c[r][1] = _mm256_setzero_ps(); }
                    for (int c0 = 0; c0 < Cin; c0 += 32) {
                        const int c1 = c0 + 32 < Cin ? c0 + 32 : Cin;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const float* ip[GEMM_MR];
                                int sy, sx, ok = 1;
                                for (int r = 0; r < GEMM_MR; ++r) {        /* (pixels past the block: a repeat of its last one, discarded) */
                                    if (!conv_src(up2, H, W, Hi, Wi, y, x0 + (r < npx ? r : npx - 1), ky, kx, &sy, &sx)) { ok = 0; break; }   /* only the row can be outside: all six or none */
                                    ip[r] = in + (((int64_t)b * Hi + sy) * Wi + sx) * Cin;
                                }
                                if (!ok) continue;
                                const float* wp = wt + ((int64_t)(ky * 3 + kx) * Cin) * Cout + co0;
                                for (int ci = c0; ci < c1; ++ci) {
                                    const __m256 b0 = _mm256_loadu_ps(wp + (int64_t)ci * Cout), b1 = _mm256_loadu_ps(wp + (int64_t)ci * Cout + 8);
                                    for (int r = 0; r < GEMM_MR; ++r) {
                                        const __m256 av = _mm256_broadcast_ss(ip[r] + ci);
                                        c[r][0] = _mm256_fmadd_ps(av, b0, c[r][0]);
                                        c[r][1] = _mm256_fmadd_ps(av, b1, c[r][1]);
                                    }
                                }
                            }
                    }
                    for (int r = 0; r < npx; ++r) {
                        _mm256_storeu_ps(acc, c[r][0]); _mm256_storeu_ps(acc + 8, c[r][1]);
                        conv_store(acc, co0, co0 + 16, bias, resid, out, b, y, x0 + r, H, W, Cout, out_mode);
                    }
                }
            }
            free(acc);
        }
}

int varref_conv3x3_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out,
                            int B, int H, int W, int Cin, int Cout, int up2, int out_mode) {
    if (up2 && ((H & 1) || (W & 1))) return VARHIP_EINVAL;
    if (out_mode < 0 || out_mode > 2 || (out_mode != 0 && resid)) return VARHIP_EINVAL;
    float* wt = (float*)malloc(sizeof(float) * 9 * (size_t)Cin * Cout);     /* [Cout][3][3][Cin] -> [3][3][Cin][Cout] */
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < Cin; ++ci) wt[((int64_t)t * Cin + ci) * Cout + co] = w[((int64_t)co * 9 + t) * Cin + ci];
    conv3x3_core(in, wt, bias, resid, out, B, H, W, Cin, Cout, up2, out_mode);
    free(wt);
    return 0;
}

/* Downsample2x (basic_vae.py:31-37): F.pad(x, (0,1,0,1)) + Conv2d(k=3, stride=2); in [B][2H][2W][Cin] -> out [B][H][W][Cout] */
int varref_conv3x3_s2_nhwc_f32(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout) {
    float* wt = (float*)malloc(sizeof(float) * 9 * (size_t)Cin * Cout);
    for (int co = 0; co < Cout; ++co)
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < Cin; ++ci) wt[((int64_t)t * Cin + ci) * Cout + co] = w[((int64_t)co * 9 + t) * Cin + ci];
    conv3x3_core(in, wt, bias, NULL, out, B, H, W, Cin, Cout, 3, 0);
    free(wt);
    return 0;
}

/* image NCHW -> channels-last, channel count zero-padded to Cpad */
int varref_nchw_to_nhwc_pad_f32(const float* in, float* out, int B, int C, int HW, int Cpad) {
    if (Cpad < C) return VARHIP_EINVAL;
    for (int b = 0; b < B; ++b) for (int p = 0; p < HW; ++p) for (int c = 0; c < Cpad; ++c)
        out[((int64_t)b * HW + p) * Cpad + c] = c < C ? in[((int64_t)b * C + c) * HW + p] : 0.0f;
    return 0;
}

/* Upsample2x (basic_vae.py:27-28) as four 2x2 convolutions on the low-resolution map: taps of the 3x3 kernel that read the same
 * source pixel through the nearest-neighbour upsampling are pre-summed.  w_phase: [4][Cout][2][2][Cin], phase = 2*py + px. */
int varref_upconv_pack_f32(const float* w, float* wp, int Cin, int Cout) {
    for (int ph = 0; ph < 4; ++ph)
        for (int co = 0; co < Cout; ++co)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int ci = 0; ci < Cin; ++ci) {
                        int py = ph >> 1, px = ph & 1;
                        int ky0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
                        int kx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
                        float s = 0.0f;
                        for (int ky = ky0; ky <= ky1; ++ky)
                            for (int kx = kx0; kx <= kx1; ++kx) s = s + w[(((int64_t)co * 3 + ky) * 3 + kx) * Cin + ci];
                        wp[((((int64_t)ph * Cout + co) * 2 + a) * 2 + b) * Cin + ci] = s;
                    }
    return 0;
}
int varref_upconv_phase_f32(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int Cin, int Cout) {
    if ((H & 1) || (W & 1)) return VARHIP_EINVAL;
    const int Hl = H / 2, Wl = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) {
                const int y = Y >> 1, py = Y & 1, x = X >> 1, px = X & 1, ph = 2 * py + px;
                for (int co = 0; co < Cout; ++co) {
                    float acc = 0.0f;
                    for (int c0 = 0; c0 < Cin; c0 += 32) {                /* same order as conv3x3_core: chunk, tap, channel */
                        int c1 = c0 + 32 < Cin ? c0 + 32 : Cin;
                        for (int a = 0; a < 2; ++a) {
                            int yy = y + a - 1 + py; if (yy < 0 || yy >= Hl) continue;
                            for (int bb = 0; bb < 2; ++bb) {
                                int xx = x + bb - 1 + px; if (xx < 0 || xx >= Wl) continue;
                                const float* ip = in + (((int64_t)b * Hl + yy) * Wl + xx) * Cin;
                                const float* w = wp + ((((int64_t)ph * Cout + co) * 2 + a) * 2 + bb) * Cin;
                                for (int ci = c0; ci < c1; ++ci) acc = vm_fma(ip[ci], w[ci], acc);
                            }
                        }
                    }
                    out[(((int64_t)b * H + Y) * W + X) * Cout + co] = acc + bias[co];
                }
            }
    return 0;
}

/* ---- convolution that also leaves the GroupNorm partial sums of its result (a GPU-side fusion; restated here as conv followed by
 * plain per-block sums in fp64).  gn_part[b][blk][co][2]; blocks of 128 consecutive pixels; the phase form numbers its blocks
 * phase-major over the low-resolution pixel index (include/var_hip.h). */
int varref_conv_gn_blocks(int H, int W, int Cout, int phase) {
    const int hw = phase ? (H / 2) * (W / 2) : H * W;
    if (H <= 0 || W <= 0 || Cout <= 0 || (Cout & 31) || (hw & 127) || (phase && ((H & 1) || (W & 1)))) return 0;
    return (phase ? 4 : 1) * (hw / 128);
}
int varref_conv3x3_gn_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out, double* gn_part,
                               int B, int H, int W, int Cin, int Cout, int up2) {
    const int nblk = varref_conv_gn_blocks(H, W, Cout, 0);
    if (!nblk || !gn_part) return VARHIP_EINVAL;
    int rc = varref_conv3x3_nhwc_f32(in, w, bias, resid, out, B, H, W, Cin, Cout, up2, 0);
    if (rc) return rc;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < nblk; ++k)
            for (int co = 0; co < Cout; ++co) {
                double s = 0.0, q = 0.0;
                for (int p = 0; p < 128; ++p) { double d = out[(((int64_t)b * H * W) + k * 128 + p) * Cout + co]; s += d; q += d * d; }
                double* o = gn_part + (((int64_t)b * nblk + k) * Cout + co) * 2; o[0] = s; o[1] = q;
            }
    return 0;
}
int varref_upconv_phase_f32(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int Cin, int Cout);
int varref_upconv_phase_gn_f32(const float* in, const float* wp, const float* bias, float* out, double* gn_part,
                               int B, int H, int W, int Cin, int Cout) {
    const int nblk = varref_conv_gn_blocks(H, W, Cout, 1);
    if (!nblk || !gn_part) return VARHIP_EINVAL;
    int rc = varref_upconv_phase_f32(in, wp, bias, out, B, H, W, Cin, Cout);
    if (rc) return rc;
    const int Hl = H / 2, Wl = W / 2, per = Hl * Wl / 128;
    for (int b = 0; b < B; ++b)
        for (int ph = 0; ph < 4; ++ph)
            for (int t = 0; t < per; ++t)
                for (int co = 0; co < Cout; ++co) {
                    double s = 0.0, q = 0.0;
                    for (int p = 0; p < 128; ++p) {
                        int lin = t * 128 + p, y = lin / Wl, x = lin - y * Wl;
                        double d = out[(((int64_t)b * H + 2 * y + (ph >> 1)) * W + 2 * x + (ph & 1)) * Cout + co]; s += d; q += d * d;
                    }
                    double* o = gn_part + (((int64_t)b * nblk + ph * per + t) * Cout + co) * 2; o[0] = s; o[1] = q;
                }
    return 0;
}
int varref_gn_stats_part_f32(const double* part, float* stats, int B, int nblk, int HW, int C, int G, float eps) {
    if (B <= 0 || nblk <= 0 || C <= 0 || G <= 0 || (C % G)) return VARHIP_EINVAL;
    const int cpg = C / G;
    const double count = (double)HW * cpg;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            double s = 0.0, s2 = 0.0;
            for (int k = 0; k < nblk; ++k)
                for (int c = 0; c < cpg; ++c) { const double* o = part + (((int64_t)b * nblk + k) * C + g * cpg + c) * 2; s += o[0]; s2 += o[1]; }
            double mean = s / count, var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            stats[((int64_t)b * G + g) * 2] = (float)mean;
            stats[((int64_t)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    return 0;
}

int64_t varref_gn_scratch_elems(int B, int HW, int C, int G) { (void)HW; (void)C; return (int64_t)B * G * 2; }

/* GroupNorm(32, C, eps=1e-6) statistics (basic_vae.py:18-19): biased variance over (HW, C/G); two-pass in double */
int varref_gn_stats_f32(const float* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps) {
    (void)scratch;
    if (G <= 0 || C % G) return VARHIP_EINVAL;
    int cpg = C / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            double s = 0.0;
            for (int p = 0; p < HW; ++p)
                for (int c = 0; c < cpg; ++c) s += (double)x[((int64_t)b * HW + p) * C + g * cpg + c];
            double n = (double)HW * cpg, mean = s / n, v = 0.0;
            for (int p = 0; p < HW; ++p)
                for (int c = 0; c < cpg; ++c) { double d = (double)x[((int64_t)b * HW + p) * C + g * cpg + c] - mean; v += d * d; }
            stats[((int64_t)b * G + g) * 2 + 0] = (float)mean;
            stats[((int64_t)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(v / n + (double)eps));
        }
    return 0;
}

int varref_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, float* out,
                        int B, int HW, int C, int G, int silu) {
    int cpg = C / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p)
            for (int c = 0; c < C; ++c) {
                const float* st = stats + ((int64_t)b * G + c / cpg) * 2;
                float v = ((x[((int64_t)b * HW + p) * C + c] - st[0]) * st[1]) * gamma[c] + beta[c];
                out[((int64_t)b * HW + p) * C + c] = silu ? vm_silu(v) : v;
            }
    return 0;
}

/* softmax over the last dim after scaling (basic_vae.py:83-84) */
int varref_softmax_rows_f32(const float* x, float* out, int64_t rows, int n, float scale) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        float* e = (float*)malloc(sizeof(float) * n);
        float m = -INFINITY;
        for (int i = 0; i < n; ++i) { e[i] = x[r * n + i] * scale; m = vm_max(m, e[i]); }
        for (int i = 0; i < n; ++i) e[i] = vm_exp(e[i] - m);
        float S = canon_sum64(e, n, 1);
        for (int i = 0; i < n; ++i) out[r * n + i] = e[i] / S;
        free(e);
    }
    return 0;
}

int varref_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int HW) {
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p)
        out[((int64_t)b * HW + p) * C + c] = in[((int64_t)b * C + c) * HW + p];
    return 0;
}
int varref_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int HW) {
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p)
        out[((int64_t)b * C + c) * HW + p] = in[((int64_t)b * HW + p) * C + c];
    return 0;
}

/* torch.where(mask, gt_tokens, sampled_tokens) of VAR.inpainting (var.py:312-328, fork) */
int varref_token_select_i64(const uint8_t* keep, const int64_t* gt, const int64_t* sampled, int64_t* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = keep[i] ? gt[i] : sampled[i];
    return 0;
}

/* nearest codebook entry (quant.py:155-157): d = |z|^2 + |e|^2 - 2 z.e, argmin, first index on ties */
int varref_nearest_code_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv) {
    float* ee = (float*)malloc(sizeof(float) * V);
    for (int v = 0; v < V; ++v) { float a = 0.0f; for (int c = 0; c < Cv; ++c) a = vm_fma(codebook[(int64_t)v * Cv + c], codebook[(int64_t)v * Cv + c], a); ee[v] = a; }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const float* zr = z + (int64_t)n * Cv;
        float zz = 0.0f; for (int c = 0; c < Cv; ++c) zz = vm_fma(zr[c], zr[c], zz);
        int best = 0; float bd = INFINITY;
        for (int v = 0; v < V; ++v) {
            float dot = 0.0f; for (int c = 0; c < Cv; ++c) dot = vm_fma(zr[c], codebook[(int64_t)v * Cv + c], dot);
            float d = (zz + ee[v]) + (-2.0f * dot);
            if (d < bd) { bd = d; best = v; }
        }
        idx_out[n] = best;
    }
    free(ee);
    return 0;
}

/* using_znorm=True (reference models/quant.py:151-153): z_NC = F.normalize(z_NC, dim=-1); argmax(z_NC @ F.normalize(E.T, dim=0)).
 * F.normalize divides every element by max(|v|_2, 1e-12); the dot product is restated as one c-ascending fma chain. */
int varref_nearest_code_cos_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv) {
    float* en = (float*)malloc(sizeof(float) * V);
    for (int v = 0; v < V; ++v) { float a = 0.0f; for (int c = 0; c < Cv; ++c) a = vm_fma(codebook[(int64_t)v * Cv + c], codebook[(int64_t)v * Cv + c], a); en[v] = vm_max(vm_sqrt(a), 1e-12f); }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const float* zr = z + (int64_t)n * Cv;
        float zz = 0.0f; for (int c = 0; c < Cv; ++c) zz = vm_fma(zr[c], zr[c], zz);
        const float zn = vm_max(vm_sqrt(zz), 1e-12f);
        int best = 0; float bd = -INFINITY;
        for (int v = 0; v < V; ++v) {
            float dot = 0.0f; for (int c = 0; c < Cv; ++c) dot = vm_fma(zr[c] / zn, codebook[(int64_t)v * Cv + c] / en[v], dot);
            if (dot > bd) { bd = dot; best = v; }
        }
        idx_out[n] = best;
    }
    free(en);
    return 0;
}

/* ================================================================================================================
 * VAR.smooth_sampling (fork, reference models/var.py:367-572).
 * Neighbour table (var.py:459-462: torch.cdist -> argsort -> [:, :n]).  Restated with the direct-form distance
 * sqrt(sum (a-b)^2) as one fma chain and a total order (distance, then index): the reference's own table depends on its
 * BLAS (cdist's |a|^2+|b|^2-2ab form) and on an unstable argsort, so only well-separated neighbours are comparable with it. */
int varref_neighbor_table_f32(const float* codebook, int V, int D, int n, int32_t* nbr_idx, float* nbr_dist) {
    if (V <= 0 || V > 8192 || D <= 0 || n <= 0 || n > V) return VARHIP_EINVAL;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; ++v) {
        sortent_t* s = (sortent_t*)malloc(sizeof(sortent_t) * V);
        float* dd = (float*)malloc(sizeof(float) * V);
        const float* a = codebook + (int64_t)v * D;
        for (int u = 0; u < V; ++u) {
            const float* b = codebook + (int64_t)u * D;
            float acc = 0.0f;
            for (int c = 0; c < D; ++c) { float d = a[c] - b[c]; acc = vm_fma(d, d, acc); }
            dd[u] = vm_sqrt(acc);
            s[u].key = vm_float_key(dd[u]); s[u].idx = u;
        }
        qsort(s, V, sizeof(sortent_t), cmp_sortent);
        for (int c = 0; c < n; ++c) { nbr_idx[(int64_t)v * n + c] = s[c].idx; nbr_dist[(int64_t)v * n + c] = dd[s[c].idx]; }
        free(s); free(dd);
    }
    return 0;
}

/* one scale of the selection (var.py:482-537): CFG combine, log_softmax, candidates = nearest neighbours of the ground-truth
 * token, count or threshold validity, arg-max of the log-probability (first index on ties; nothing valid -> candidate 0),
 * and the log-softmax of the negated candidate distances at the winner. */
int varref_smooth_select_f32(const float* logits, const int64_t* gt, const int32_t* nbr_idx, const float* nbr_dist, int n,
                             int cand_count, int use_thr, float thr, float ratio, int B, int l, int V, double t_cfg,
                             int64_t* idx_out, float* maxval_out, float* distlp_out, float* cfg_out) {
    if (B <= 0 || l <= 0 || V <= 0 || n <= 0 || n > V || (!use_thr && (cand_count < 1 || cand_count > n))) return VARHIP_EINVAL;
    const float ca = (float)(1.0 + t_cfg), cb = (float)t_cfg;
    const int64_t rows = (int64_t)B * l;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float* lc = logits + r * V; const float* lu = logits + (rows + r) * V;
        float* x = (float*)malloc(sizeof(float) * 2 * V); float* e = x + V;
        float m = -INFINITY;
        for (int i = 0; i < V; ++i) { float a = ca * lc[i]; float b = cb * lu[i]; x[i] = a - b; m = vm_max(m, x[i]); }
        for (int i = 0; i < V; ++i) e[i] = vm_exp(x[i] - m);
        const float ls = vm_log(canon_sum256(e, V));
        if (cfg_out) memcpy(cfg_out + r * V, x, sizeof(float) * V);
        const int32_t* ni = nbr_idx + gt[r] * n; const float* nd = nbr_dist + gt[r] * n;
        const float d0 = nd[0];
        const float eff = d0 + (thr - d0) * ratio;
        int best = 0; float bl = -INFINITY; float dmax = -INFINITY;
        for (int c = 0; c < n; ++c) {
            float lp = (x[ni[c]] - m) - ls;
            int valid = use_thr ? (nd[c] <= eff) : (c < cand_count);
            if (!valid) lp = -INFINITY;
            if (lp > bl) { bl = lp; best = c; }
            dmax = vm_max(dmax, -nd[c]);
        }
        float* ed = (float*)malloc(sizeof(float) * n);
        for (int c = 0; c < n; ++c) ed[c] = vm_exp(-nd[c] - dmax);
        const float dls = vm_log(canon_sum256(ed, n));
        idx_out[r] = ni[best]; maxval_out[r] = bl; distlp_out[r] = (-nd[best] - dmax) - dls;
        free(ed); free(x);
    }
    return 0;
}
