/* var_hip.h — C ABI of libvar_hip.so: the MI355X (gfx950) kernels of VAR's next-scale sampling path.
 *
 * The reference (culiver/VAR) is pure Python on PyTorch: it has no FFI layer.  The drop-in boundary is the
 * `models` nn.Module API; *under* it the build calls this library through ctypes (var_amd/hip.py).  Each entry
 * point below names the reference code it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors); the library never allocates or
 *     frees caller-visible memory and keeps no state besides the optional timing table;
 *   - all matrices are row-major fp32; "ld*" are leading dimensions in elements;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); launches are asynchronous;
 *   - return value: 0 on success, VARHIP_EINVAL (-1) for a shape/argument the kernels do not support,
 *     -(1000+hipError_t) when the launch itself failed.  No exceptions cross the ABI.
 *   - the CPU oracle (oracle/var_oracle.c) exports the same functions with prefix `varref_` and no stream
 *     argument, on HOST pointers: tests drive both with the same arguments.
 *   - arithmetic contract: dot products are k-ascending fp32 fma chains starting
 *     from 0 (what v_mfma_f32_32x32x2_f32 computes), then bias, then epilogue, each separately rounded.
 */
#ifndef VAR_HIP_H
#define VAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VARHIP_EINVAL (-1)

typedef void* varhip_stream_t;

/* library / build info: returns a static string "var_hip <version> gfx950" */
const char* varhip_version(void);

/* ---- GEMM ------------------------------------------------------------------------------------------------
 * out[b][m][n] = epi( sum_k A[b][m][k] * W[b][n][k] + bias )          (torch F.linear: x @ W^T + b)
 * replaces: F.linear at basic_var.py:93 (mat_qkv), :119 (proj), :52 (fc1/fc2), :147,:170 (ada_lin), var.py:124
 * (head), and the 1x1 convs of basic_vae.py:53,69,71,83,89 (nin_shortcut, qkv, proj_out, the two bmm's).
 *   epi = VARHIP_EPI_NONE : acc + bias
 *         VARHIP_EPI_GELU : gelu_tanh(acc + bias)                              (basic_var.py:40,52)
 *         VARHIP_EPI_RESID: resid[m][n] + (acc + bias) * gamma[m / rows_per_group][n]   (basic_var.py:157-158);
 *                           gamma == NULL means no scaling: resid + (acc + bias)       (basic_vae.py:60,92)
 *   bias may be NULL.  bias_per_row != 0: bias is indexed by m instead of n.
 *   batch >= 1 with element strides sA/sW/sO (sW or sA may be 0 to share an operand); resid/gamma only with batch==1.
 * Fast path: K % 32 == 0, lda/ldw/sA/sW % 4 == 0, 16-byte aligned A/W and each operand of one batch element spanning < 4 GiB
 * (rows are addressed as 32-bit byte offsets from a scalar base); anything else takes an element-wise-load variant. */
#define VARHIP_EPI_NONE 0
#define VARHIP_EPI_GELU 1
#define VARHIP_EPI_RESID 2
int varhip_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                       float* out, int64_t ldo, int M, int N, int K, int epi,
                       const float* resid, int64_t ldr, const float* gamma, int64_t ldg, int rows_per_group,
                       int bias_per_row, int batch, int64_t sA, int64_t sW, int64_t sO, varhip_stream_t stream);

/* y[i] = x[i] * sigmoid(x[i])   — the SiLU in front of every ada_lin (basic_var.py:147,170; var.py:80) */
int varhip_silu_f32(const float* x, float* y, int64_t n, varhip_stream_t stream);

/* out[b][j] = base[j] + cond[b][j], j < n   — shared AdaLN: ada_gss + cond_BD (basic_var.py:153-154) */
int varhip_add_bcast_f32(const float* base, const float* cond, float* out, int rows, int n, varhip_stream_t stream);

/* ---- AdaLN -----------------------------------------------------------------------------------------------
 * out[m][:] = LN(x[m][:]) * (scale[g][:] + 1) + shift[g][:],  g = m / rows_per_group, LN without affine,
 * biased variance, eps inside the sqrt.   replaces basic_var.py:157,158,174 (ln_wo_grad(...).mul(scale.add(1)).add_(shift)) */
int varhip_ln_modulate_f32(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                           float* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream);

/* ---- q/k/v post-processing + KV-cache append ------------------------------------------------------------
 * qkv: [B2*l][3*C] (q | k | v, each H heads x 64).  For every row and head:
 *   l2norm != 0: q = q/max(|q|,1e-12) * exp(min(scale_mul[h], ln 100)),  k = k/max(|k|,1e-12)   (basic_var.py:101-105)
 *   l2norm == 0: q = q * plain_scale (attention scale folded into q; basic_var.py:72,117), k unchanged
 * q -> q_out[B2*l][C]; k,v -> caches [B2][H][Lmax][64] at positions pos0 .. pos0+l-1 (replaces the torch.cat
 * cache growth of basic_var.py:107-109 by an in-place append). */
int varhip_qkv_prep_f32(const float* qkv, const float* scale_mul, float plain_scale, int l2norm,
                        float* q_out, float* kcache, float* vcache,
                        int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream);

/* ---- mat_qkv GEMM with that post-processing fused into its epilogue -------------------------------------
 * Equivalent, bit for bit, to varhip_gemm_nt_f32(A, W[3C][K], bias[3C]) -> qkv[M][3C] followed by varhip_qkv_prep_f32,
 * without the [M][3C] round trip through HBM.   replaces basic_var.py:93 (F.linear with the q_bias|zero_k_bias|v_bias
 * concatenation) through :109.  Requires M == B2*l, C == H*64, K % 32 == 0, lda/ldw % 4 == 0, 16-byte aligned pointers and
 * A and W each spanning < 4 GiB (VARHIP_EINVAL otherwise). */
int varhip_gemm_qkv_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, int M, int C, int K,
                        const float* scale_mul, float plain_scale, int l2norm,
                        float* q_out, float* kcache, float* vcache,
                        int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream);

/* ---- attention of l new queries over curL cached keys (no mask: block-causal by construction) -----------
 * out[b][t][h*64+c] = sum_j softmax_j(q[b][t][h] . k[b][h][j]) v[b][h][j][c],  j < curL
 * replaces slow_attn / flash_attn_func / memory_efficient_attention at basic_var.py:111-117 (head_dim 64, scale 1:
 * the scale is already folded into q by varhip_qkv_prep_f32).
 * Arithmetic contract (what makes the GPU result reproducible bit for bit by oracle/var_oracle.c):
 *   - the running-maximum recurrence over tiles of 32 keys: m' = max(m, max_tile s), a = e(m - m'), l = l a + sum p, O = O a + P V,
 *     p = e(s - m'), with e = vm_exp_le0 of include/var_math.h;
 *   - every dot product (64 channels of q.k; the 32 keys of a tile in p.v) is one fp32 fma chain in the 4-interleaved order
 *     0,4,1,5,2,6,3,7, 8,12,9,13, ... (inside each group of eight: j, j+4) — the order in which an MFMA 32x32x2 consumes operands
 *     that both lane halves read as 16 contiguous bytes;
 *   - l is kept as four partial sums over the keys with equal ((key >> 2) & 1, key & 1), ascending, added as (S00 + S01) + (S10 + S11);
 *   - out = O * (1 / l). */
int varhip_attn_cached_f32(const float* q, const float* kcache, const float* vcache, float* out,
                           int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream);

/* ---- one AdaLNSelfAttn block (basic_var.py:152-159) in one call ---------------------------------------------
 * x <- x + gamma1 * proj(attn(LN(x)(1+scale1)+shift1));  x <- x + gamma2 * fc2(gelu(fc1(LN(x)(1+scale2)+shift2)))
 * issued as the seven launches ln_modulate, gemm_qkv, attn_cached, gemm_nt(RESID), ln_modulate, gemm_nt(GELU),
 * gemm_nt(RESID) with exactly their arithmetic.  ada: this block's [B2][6C] AdaLN rows (gamma1|gamma2|scale1|scale2|shift1|
 * shift2, row stride ld_ada); x2, xn, q, att, hid are workspaces ([M][C], hid [M][hidden]); the result is left in x.
 * A host that pays ~10 us per FFI call would otherwise starve the GPU at the small scales (launches of 5-20 us). */
int varhip_adaln_block_f32(float* x, float* x2, float* xn, float* q, float* att, float* hid,
                           const float* ada, int64_t ld_ada,
                           const float* qkv_w, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                           const float* proj_w, const float* proj_b, const float* fc1_w, const float* fc1_b,
                           const float* fc2_w, const float* fc2_b, float* kcache, float* vcache,
                           int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream);

/* ---- classifier-free guidance + top-k/top-p + multinomial(1) -------------------------------------------
 * logits: [2B][l][V] (rows 0..B-1 conditional, B..2B-1 unconditional).  For row r=(b,t):
 *   x = (float)(1+t_cfg) * cond - (float)t_cfg * uncond                             (var.py:172-173)
 *   top_k > 0: x[x < kth_largest(x, top_k)] = -inf                                   (helpers.py:8-10)
 *   top_p > 0: ascending stable sort, softmax, cumsum (fp64 accumulate, fp32 per element), remove where
 *              cum <= (float)(1-top_p) except the largest                            (helpers.py:11-15)
 *   idx = argmax_v softmax(x)[v] / noise[r][v]    (first max)  == torch.multinomial(p, 1, generator) with
 *              noise = empty_like(p).exponential_(1, generator)                      (helpers.py:19)
 * idx_out: int64 [B*l].  masked_out (optional, may be NULL): [B*l][V] the filtered logits (what the reference leaves in place).
 * Constraint: V % 256 == 0, V <= 8192, 0 <= top_k <= V; `logits` 16-byte aligned (rows are read 16 bytes per lane; V % 256 == 0 keeps
 * every row aligned once the base is): VARHIP_EINVAL otherwise.  Any tensor start handed out by hipMalloc / torch's allocator qualifies;
 * a view at an odd element offset does not — copy it first. */
int varhip_cfg_sample_f32(const float* logits, const float* noise, int64_t* idx_out, float* masked_out,
                          int B, int l, int V, double t_cfg, int top_k, double top_p, varhip_stream_t stream);
/* test hook: 1 = every top-p cut is decided by the sequential fp64 walk (the definition); 0 (default) = by a parallel prefix sum wherever
 * that provably gives the same cut, by the walk otherwise.  Same results either way. */
int varhip_sampler_force_walk(int on);

/* ---- multi-scale quantizer step ---------------------------------------------------------------------------
 * Feature maps are kept channels-last: f_hat[B][P][P][Cv].
 * (1) h = codebook[idx] as [B][pn][pn][Cv]                                          (var.py:177,182; quant.py:39)
 * (2) up = bicubic(h -> PxP) via 4-tap tables (tap_idx/tap_w: [P][4], same table for rows and columns; NULL when pn==P)
 *                                                                                    (quant.py:190, F.interpolate 'bicubic')
 * (3) f_hat += (1-ratio)*up + ratio*(conv3x3(up; phi_w[Cv][3][3][Cv]) + phi_b)      (quant.py:199-206, :191)
 * `up` is caller-provided scratch [B][P][P][Cv]. */
int varhip_quant_accum_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                           const float* phi_w, const float* phi_b, float ratio,
                           float* up, float* f_hat, int B, int pn, int P, int Cv, varhip_stream_t stream);

/* Same step with the scale's embeddings given directly, h: [B][pn*pn][Cv] (more_smooth: gumbel-softmax @ codebook, var.py:178-182) */
int varhip_quant_accum_h_f32(const float* h, const int32_t* tap_idx, const float* tap_w,
                             const float* phi_w, const float* phi_b, float ratio,
                             float* up, float* f_hat, int B, int pn, int P, int Cv, varhip_stream_t stream);

/* y[r][:] = softmax((x[r][:] * mul + (-ln noise[r][:])) / tau)   — gumbel_softmax_with_rng(logits.mul(1+ratio), tau, hard=False, rng)
 * (helpers.py:22-36, var.py:179-180); x are the top-k/top-p filtered CFG logits (-inf entries give probability 0). V % 256 == 0. */
int varhip_gumbel_softmax_f32(const float* x, const float* noise, float* y, int64_t rows, int V, float mul, float tau, varhip_stream_t stream);

/* (4) next-scale input: pooled = adaptive_avg_pool(f_hat -> pq x pq)  (quant.py:192, F.interpolate 'area');
 *     x[b][t][:] = x[b+B][t][:] = word_w[C][Cv] . pooled[b][t][:] + word_b + lvl_pos[t][:]     (var.py:185-187)
 * lvl_pos must already point at row cur_L.  pooled: caller-provided [B][pq*pq][Cv] buffer (intermediate; kept for inspection). */
int varhip_next_map_f32(const float* f_hat, const float* word_w, const float* word_b, const float* lvl_pos,
                        float* x_out, float* pooled, int B, int P, int pq, int C, int Cv, varhip_stream_t stream);

/* lvl_pos[t][:] = lvl_embed[lvl[t]][:] + pos[t][:]     (var.py:153) */
int varhip_lvl_pos_f32(const float* lvl_embed, const int64_t* lvl, const float* pos, float* out, int L, int C, varhip_stream_t stream);

/* prologue (var.py:151,154): cond[b2][:] = class_emb[b2 < B ? label[b2] : num_classes][:];
 * x[b2][t][:] = (cond[b2][:] + pos_start[t][:]) + lvl_pos[t][:],  t < first_l */
int varhip_first_map_f32(const float* class_emb, const int64_t* labels, int num_classes, const float* pos_start,
                         const float* lvl_pos, float* cond, float* x_out, int B, int C, int first_l, varhip_stream_t stream);

/* ---- VQVAE decoder (channels-last) ----------------------------------------------------------------------
 * 3x3 convolution, stride 1, zero padding 1 (cross-correlation), w packed [Cout][3][3][Cin]:
 *   out[b][y][x][co] = sum_{ky,kx,ci} in[b][y+ky-1][x+kx-1][ci] * w[co][ky][kx][ci] + bias[co]  (+ resid[b][y][x][co])
 *   up2 != 0: `in` is [B][H/2][W/2][Cin] and is read through a nearest-neighbour 2x upsampling
 *             (basic_vae.py:27-28 Upsample2x: F.interpolate(scale 2,'nearest') then conv)
 *   out_mode 0: out is [B][H][W][Cout];  out_mode 1: out is [B][Cout][H][W] and holds (clamp(v,-1,1)+1)*0.5
 *             (vqvae.py:63 clamp_ and var.py:190 add_(1).mul_(0.5) fused into the last conv);
 *   out_mode 2: [B][Cout][H][W] holding clamp(v,-1,1) only (VQVAE.fhat_to_img's own contract)
 * replaces every Conv2d(k=3) of basic_vae.py (ResnetBlock :48,:51; conv_in :180; conv_out :208; Upsample2x :25)
 * and vqvae.py:49 post_quant_conv.   Constraints: Cin % 32 == 0 (a K tile of the implicit GEMM lies inside one tap); the input
 * samples one 128-pixel tile can touch (one sample when H*W >= 128) plus one row must span < 2 GiB and the packed weights < 4 GiB
 * (buffer-descriptor window / 32-bit offsets of the DMA requests; VARHIP_EINVAL otherwise).  The batch itself may exceed 4 GiB.
 * Summation order (arithmetic contract of every 3x3 convolution here): one fma chain per output over 32-channel chunks
 * (outermost), then the taps (ky, kx), then the channels of the chunk. */
int varhip_conv3x3_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out,
                            int B, int H, int W, int Cin, int Cout, int up2, int out_mode, varhip_stream_t stream);

/* Upsample2x (basic_vae.py:27-28: nearest 2x, then conv3x3) as four 2x2 convolutions on the LOW-resolution map, one per output
 * parity (py,px): the 3x3 taps that read the same source pixel through the upsampling are pre-summed, 2.25x fewer MACs.
 *   varhip_upconv_pack_f32 : w [Cout][3][3][Cin] -> w_phase [4][Cout][2][2][Cin]   (one-time, per weight)
 *   varhip_upconv_phase_f32: in [B][H/2][W/2][Cin] -> out [B][H][W][Cout], bias added
 * Mathematically equal to varhip_conv3x3_nhwc_f32(up2=1); rounding differs at the 1e-7 level (decoder only: pixels, not tokens). */
int varhip_upconv_pack_f32(const float* w, float* w_phase, int Cin, int Cout, varhip_stream_t stream);
int varhip_upconv_phase_f32(const float* in, const float* w_phase, const float* bias, float* out,
                            int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
/* ... and with the GroupNorm partials of its result (see varhip_conv3x3_gn_nhwc_f32): block blk = phase * ((H/2)*(W/2)/128) + t
 * covers low-resolution pixels [128 t, 128 t + 128) of that phase */
int varhip_upconv_phase_gn_f32(const float* in, const float* w_phase, const float* bias, float* out, double* gn_part,
                               int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);

/* GroupNorm statistics: stats[b][g] = {mean, rstd} over (HW, C/G) with biased variance (basic_vae.py:18-19, eps 1e-6).
 * scratch: caller-provided, at least varhip_gn_scratch_elems(B,HW,C,G) doubles. */
int64_t varhip_gn_scratch_elems(int B, int HW, int C, int G);
int varhip_gn_stats_f32(const float* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps, varhip_stream_t stream);
/* out = ((x - mean) * rstd) * gamma[c] + beta[c], then SiLU if silu != 0   (basic_vae.py:58,59,74,225) */
int varhip_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, float* out,
                        int B, int HW, int C, int G, int silu, varhip_stream_t stream);

/* the decoder's tail in one pass (basic_vae.py:224-226 norm_out -> swish -> conv_out, the callers' clamp vqvae.py:63 and (x + 1) / 2 var.py:190):
 * out = clamp(conv3x3(SiLU(GroupNorm(x))) + bias, -1, 1) as fp32 NCHW (out_mode 2) or de-normalised to [0, 1] (out_mode 1); stats [B][G][2] = (mean, rstd).
 * Bit-identical to varhip_gn_apply_f32 (silu = 1) followed by varhip_conv3x3_nhwc_f32 (same out_mode): same operations per element, same
 * chunk / tap / channel summation order.  Shapes it does not take (H % 8, W % 32, Cin % 32, Cout > 4): VARHIP_EINVAL. */
int varhip_gn_silu_conv_out_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                                float* out, int B, int H, int W, int Cin, int Cout, int G, int out_mode, varhip_stream_t stream);

/* out[r][:] = softmax(x[r][:] * scale), rows of length n   (basic_vae.py:83-84: bmm(...).mul_(w_ratio); softmax(dim=2)) */
int varhip_softmax_rows_f32(const float* x, float* out, int64_t rows, int n, float scale, varhip_stream_t stream);

/* [B][C][H][W] <-> [B][H][W][C] copies for the API edge (f_hat is NCHW in the reference's API) */
int varhip_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int HW, varhip_stream_t stream);
int varhip_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int HW, varhip_stream_t stream);

/* ---- encode side and teacher forcing ("next" rows of SURVEY.md §8f: image -> tokens -> teacher-forced logits) -----------------
 * Downsample2x of the encoder (basic_vae.py:31-37): F.pad(x,(0,1,0,1)) + Conv2d(k=3, stride=2): in [B][2H][2W][Cin] -> out [B][H][W][Cout] */
/* The same convolution (out_mode 0) that also leaves GroupNorm partial sums of its result: gn_part[b][blk][co][2] (doubles) =
 * (sum, sum of squares) over the blk-th block of 128 consecutive pixels of sample b, so the GroupNorm that follows
 * (basic_vae.py:18-19 inside ResnetBlock / AttnBlock / norm_out) needs no statistics pass over the tensor.
 * varhip_conv_gn_blocks(H, W, Cout, phase) = blocks per sample, or 0 when unsupported (needs H*W % 128 == 0, Cout % 32 == 0;
 * phase != 0 describes varhip_upconv_phase_gn_f32 below: 4 * ((H/2)*(W/2) / 128) blocks, phase-major).
 * varhip_gn_stats_part_f32 turns the partials into the (mean, rstd) pairs varhip_gn_apply_f32 takes (blocks in order, channels of
 * a group in order, fp64). */
int varhip_conv_gn_blocks(int H, int W, int Cout, int phase);
int varhip_conv3x3_gn_nhwc_f32(const float* in, const float* w, const float* bias, const float* resid, float* out, double* gn_part,
                               int B, int H, int W, int Cin, int Cout, int up2, varhip_stream_t stream);
int varhip_gn_stats_part_f32(const double* gn_part, float* stats, int B, int nblk, int HW, int C, int G, float eps, varhip_stream_t stream);
int varhip_conv3x3_s2_nhwc_f32(const float* in, const float* w, const float* bias, float* out,
                               int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
/* image [B][C][HW] -> [B][HW][Cpad] with zero channels C..Cpad-1 (conv kernels need Cin % 32 == 0; zero channels add exact zeros) */
int varhip_nchw_to_nhwc_pad_f32(const float* in, float* out, int B, int C, int HW, int Cpad, varhip_stream_t stream);
/* pooled[b][t][:] = mean of f[b] over the adaptive window of output cell t   (F.interpolate(mode='area'), quant.py:150,183) */
int varhip_area_pool_f32(const float* f, float* pooled, int B, int P, int pq, int Cv, varhip_stream_t stream);
/* x[b][t][:] = x[b+B][t][:] = word_w . pooled[b][t][:] + word_b + lvl_pos[t][:]   (var.py:186-187, 206-207); x_out holds 2*B*lq rows */
int varhip_word_embed_f32(const float* pooled, const float* word_w, const float* word_b, const float* lvl_pos,
                          float* x_out, int B, int lq, int C, int Cv, varhip_stream_t stream);
/* one scale of VectorQuantizer2.f_to_idxBl_or_fhat (quant.py:159-163): as varhip_quant_accum_f32, and f_rest -= the same h */
int varhip_quant_residual_f32(const int64_t* idx, const float* codebook, const int32_t* tap_idx, const float* tap_w,
                              const float* phi_w, const float* phi_b, float ratio, float* up, float* f_hat, float* f_rest,
                              int B, int pn, int P, int Cv, varhip_stream_t stream);

/* out[i] = keep[i] ? gt[i] : sampled[i]   — VAR.inpainting's torch.where(mask, gt_tokens, sampled_tokens) (var.py:312-328, fork) */
int varhip_token_select_i64(const uint8_t* keep, const int64_t* gt, const int64_t* sampled, int64_t* out, int64_t n, varhip_stream_t stream);

/* ---- VAR.smooth_sampling (fork, var.py:367-572) ----------------------------------------------------------
 * Neighbour table of the codebook (var.py:459-462: cdist + argsort + [:, :n]): nbr_idx[v][c] = the c-th nearest code of v
 * (c = 0 is v itself), nbr_dist[v][c] its L2 distance; ascending distance, ties by index.  Distances are the direct form
 * sqrt(sum (a-b)^2), one fma chain over the channels (the reference's BLAS-based cdist differs from it by rounding only).
 * codebook: [V][D], V <= 8192, 1 <= n <= V. */
int varhip_neighbor_table_f32(const float* codebook, int V, int D, int n, int32_t* nbr_idx, float* nbr_dist, varhip_stream_t stream);
/* One scale's selection (var.py:482-537).  logits: [2B][l][V] (CFG pair), gt: [B*l] ground-truth tokens of the scale.
 *   lp = log_softmax((1+t)*cond - t*uncond);  candidates = nbr_idx[gt][0..n);
 *   use_thr == 0: the first cand_count candidates are valid;  use_thr != 0: those with dist <= d0 + (thr - d0)*ratio;
 *   idx_out = the valid candidate with the largest lp (first on ties; none valid -> candidate 0, maxval -inf),
 *   maxval_out = its lp, distlp_out = log_softmax(-nbr_dist[gt][0..n))[winner];  cfg_out (nullable): the combined logits [B*l][V]. */
int varhip_smooth_select_f32(const float* logits, const int64_t* gt, const int32_t* nbr_idx, const float* nbr_dist, int n,
                             int cand_count, int use_thr, float thr, float ratio, int B, int l, int V, double t_cfg,
                             int64_t* idx_out, float* maxval_out, float* distlp_out, float* cfg_out, varhip_stream_t stream);

/* ---- nearest-codebook lookup (encode side; quant.py:150-157) --------------------------------------------
 * idx[n] = argmin_v ( |z_n|^2 + |e_v|^2 - 2 z_n.e_v ), first index on ties; z: [N][Cv], codebook: [V][Cv] */
int varhip_nearest_code_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv, varhip_stream_t stream);

/* the same lookup for VectorQuantizer2(using_znorm=True) (quant.py:151-153): idx[n] = argmax_v (z_n / max(|z_n|,1e-12)) . (e_v / max(|e_v|,1e-12)),
 * first index on ties; every element is divided by its vector's norm before the (c-ascending fma) dot product, as F.normalize does */
int varhip_nearest_code_cos_f32(const float* z, const float* codebook, int64_t* idx_out, int N, int V, int Cv, varhip_stream_t stream);

/* ==== 16-bit-input throughput mode ("f16") ===================================================================
 * The reference's harness runs the path under torch.autocast('cuda', dtype=torch.float16) (demo_sample.py:66-68): every F.linear of
 * basic_var.py then computes in fp16 and attention takes the flash path with fp16 q/k/v (basic_var.py:97,113).  These entry points are
 * that mode on MI355X: fp16 operands (`void*` = _Float16 data), fp32 accumulation on v_mfma_f32_*_f16, fp32 bias / GELU / AdaLN gate /
 * residual, one rounding where an output is fp16.  They are NOT under the fp32 bit-exactness contract (the MFMA-internal reduction over
 * k is not a k-ascending chain): tests compare them with the CPU twin (oracle/var_oracle.py, f16=True: the fp32 restatement with the same
 * rounding points) within a stated tolerance.  LayerNorm, GroupNorm, softmax statistics, the sampler and the quantizer stay fp32. */

/* out[m][n] = epi(sum_k A[m][k] W[n][k] + bias[n]); A: [M][lda] fp16, W: [N][ldw] fp16, bias / gamma fp32, out and resid fp16 or fp32
 * (out_f16 / resid_f16).  epi as varhip_gemm_nt_f32.  Requires K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0, 16-byte aligned pointers.
 * replaces F.linear at basic_var.py:93,119,52 and var.py:124 under fp16 autocast. */
int varhip_gemm_nt_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* out, int64_t ldo, int out_f16, int M, int N, int K, int epi,
                       const void* resid, int64_t ldr, int resid_f16, const float* gamma, int64_t ldg, int rows_per_group,
                       int batch, int64_t sA, int64_t sW, int64_t sO, varhip_stream_t stream);
/* testing / experiments: force the tile of the following f16 GEMM calls (0: 128x128, 1: 64x64, 2: 256x256 with 8 waves; -1: automatic choice) */
int varhip_gemm16_force_tile(int tile);
/* experiments / tests: 1 (default) = whole 256x256 tiles run on the persistent kernel k_gemm16p (one workgroup per CU walking a tile list),
 * 0 = on k_gemm16<8,4,2,4> (one workgroup per tile).  Identical results. */
int varhip_gemm16_persistent(int on);
/* mat_qkv + q/k L2-norm + scale + KV-cache append (basic_var.py:93-109): fp16 x fp16 -> fp32 -> fp16 q [M][C] and fp16 caches [B2][H][Lmax][64] */
int varhip_gemm_qkv_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int M, int C, int K,
                        const float* scale_mul, float plain_scale, int l2norm,
                        void* q_out, void* kcache, void* vcache, int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream);
/* attention over the fp16 KV cache (basic_var.py:107-117 on the flash path): fp32 scores and softmax, p rounded to fp16 for p.v, fp16 out */
int varhip_attn_cached_f16(const void* q, const void* kcache, const void* vcache, void* out,
                           int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream);
/* varhip_ln_modulate_f32 with the result rounded to fp16 (the A operand of the next GEMM) */
int varhip_ln_modulate_f16out(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                              void* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream);
/* one AdaLNSelfAttn block (basic_var.py:152-159) in this mode: fp32 residual stream and AdaLN parameters, fp16 GEMM operands and KV cache */
int varhip_adaln_block_f16(float* x, float* x2, void* xn16, void* q16, void* att16, void* hid16, const float* ada, int64_t ld_ada,
                           const void* qkv_w16, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                           const void* proj_w16, const float* proj_b, const void* fc1_w16, const float* fc1_b,
                           const void* fc2_w16, const float* fc2_b, void* kcache16, void* vcache16,
                           int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream);

/* the decoder's convolutions in this mode (basic_vae.py:22-28,40-60,163-226 under fp16 autocast): fp16 channels-last activations and
 * weights ([Cout][3][3][Cin], or the phase form of varhip_upconv_pack_f32 rounded to fp16), fp32 bias, fp16 residual and output;
 * out_mode 1 / 2: the last conv writes fp32 NCHW, de-normalised to [0,1] / clamped to [-1,1].  gn_part (nullable): per-block
 * per-channel (sum, sum of squares) of the ROUNDED result, as varhip_conv3x3_gn_nhwc_f32.  Cin % 32 == 0. */
int varhip_conv3x3_nhwc_f16(const void* in, const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                            int B, int H, int W, int Cin, int Cout, int out_mode, varhip_stream_t stream);
int varhip_upconv_phase_f16(const void* in, const void* w_phase, const float* bias, void* out, double* gn_part,
                            int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
/* ResnetBlock's `conv(swish(norm(x)))` (basic_vae.py:57-60) in ONE launch: `in` is the RAW 16-bit map, table [B][2][Cin] fp32 its GroupNorm as one
 * multiply-add per element (varhip_gn_scale_shift_f32 below: scale = rstd * gamma, shift = beta - mean * scale), silu != 0: swish.  The normalisation
 * runs on the convolution's input patch in LDS, so the apply pass over the map (varhip_gn_apply_f16: one read + one write of every activation)
 * disappears.  Bit-identical to varhip_gn_apply_f16 followed by varhip_conv3x3_nhwc_f16 (out_mode 0; bias, resid, gn_part as there).  Shapes:
 * varhip_conv16_gn_fusable(B, H, W, Cin, Cout) != 0 (maps that tile into 8 x 32 or 16 x 16 patches with a workgroup for every CU, Cout % 128 == 0 or
 * % 160 == 0, the table within the LDS budget: Cin <= 320 at 8 x 32 patches); VARHIP_EINVAL otherwise — the caller then runs the two launches. */
int varhip_conv16_gn_fusable(int B, int H, int W, int Cin, int Cout);
int varhip_gnconv3x3_nhwc_f16(const void* in, const float* table, int silu,
                              const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                              int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
/* table[b][0][c] = stats[b][g(c)].rstd * gamma[c], table[b][1][c] = beta[c] - stats[b][g(c)].mean * table[b][0][c]   (what varhip_gn_apply_* forms per channel) */
int varhip_gn_scale_shift_f32(const float* stats, const float* gamma, const float* beta, float* table, int B, int C, int G, varhip_stream_t stream);
/* testing / experiments: force the pixel tile of the following f16 convolutions (2: 128 pixels, 4 waves, two workgroups per CU;
 * 4: 256 pixels, 8 waves, two workgroups per CU; 8: the halo-patch kernel where the shape allows it; anything else: by size) */
int varhip_conv16_force_tile(int wm);
/* GroupNorm on fp16 [B][HW][C] (basic_vae.py:18-19): statistics in fp64, affine + optional SiLU in fp32, fp16 result */
int varhip_gn_stats_f16(const void* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps, varhip_stream_t stream);
int varhip_gn_apply_f16(const void* x, const float* stats, const float* gamma, const float* beta, void* out,
                        int B, int HW, int C, int G, int silu, varhip_stream_t stream);
/* casts at the edges of the mode (n % 4 == 0, 16-byte aligned) */
/* the decoder's tail in one pass (basic_vae.py:224-226 norm_out -> swish -> conv_out, the caller's clamp vqvae.py:63 and (x + 1) / 2 var.py:190):
 * out = clamp(conv3x3(SiLU(GroupNorm(x))) + bias) as fp32 NCHW (out_mode 2) or de-normalised to [0, 1] (out_mode 1); stats [B][G][2] = (mean, rstd).
 * Bit-identical to varhip_gn_apply_f16 followed by varhip_conv3x3_nhwc_f16.  Shapes it does not take (H % 8, W % 32, Cin % 32, Cout > 16): VARHIP_EINVAL. */
int varhip_gn_silu_conv_out_f16(const void* x, const float* stats, const float* gamma, const float* beta, const void* w, const float* bias,
                                float* out, int B, int H, int W, int Cin, int Cout, int G, int out_mode, varhip_stream_t stream);
int varhip_cast_f32_to_f16(const float* in, void* out, int64_t n, varhip_stream_t stream);
int varhip_cast_f16_to_f32(const void* in, float* out, int64_t n, varhip_stream_t stream);

/* ---- the same mode with bfloat16 storage ("bf16"; the reference's other 16-bit option: utils/arg_util.py `fp16: int  # 1: using fp16, 2: bf16`,
 * trainer / AmpOptimizer autocast dtype, utils/amp_sc.py).  One entry point per _f16 entry point above, same arguments and meaning with
 * `void*` = bfloat16 data (the out_f16 / resid_f16 flags then mean bfloat16): the same kernels compiled with the bf16 MFMA opcodes and conversions
 * (var_amd/csrc/elem16.h).  Rounding points as in the fp16 flavour; 8 significant bits instead of 11, fp32's exponent range. */
int varhip_gemm_nt_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* out, int64_t ldo, int out_f16, int M, int N, int K, int epi,
                       const void* resid, int64_t ldr, int resid_f16, const float* gamma, int64_t ldg, int rows_per_group,
                       int batch, int64_t sA, int64_t sW, int64_t sO, varhip_stream_t stream);
int varhip_gemm16_force_tile(int tile);
int varhip_gemm16_persistent(int on);
int varhip_gemm_qkv_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int M, int C, int K,
                        const float* scale_mul, float plain_scale, int l2norm,
                        void* q_out, void* kcache, void* vcache, int B2, int l, int H, int pos0, int Lmax, varhip_stream_t stream);
int varhip_attn_cached_bf16(const void* q, const void* kcache, const void* vcache, void* out,
                           int B2, int l, int H, int curL, int Lmax, varhip_stream_t stream);
int varhip_ln_modulate_bf16out(const float* x, const float* scale, int64_t ld_scale, const float* shift, int64_t ld_shift,
                              void* out, int M, int C, int rows_per_group, float eps, varhip_stream_t stream);
int varhip_adaln_block_bf16(float* x, float* x2, void* xn16, void* q16, void* att16, void* hid16, const float* ada, int64_t ld_ada,
                           const void* qkv_w16, const float* qkv_b, const float* scale_mul, float plain_scale, int l2norm,
                           const void* proj_w16, const float* proj_b, const void* fc1_w16, const float* fc1_b,
                           const void* fc2_w16, const float* fc2_b, void* kcache16, void* vcache16,
                           int B2, int l, int C, int H, int hidden, int pos0, int Lmax, float eps, varhip_stream_t stream);
int varhip_conv3x3_nhwc_bf16(const void* in, const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                            int B, int H, int W, int Cin, int Cout, int out_mode, varhip_stream_t stream);
int varhip_upconv_phase_bf16(const void* in, const void* w_phase, const float* bias, void* out, double* gn_part,
                            int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
int varhip_gnconv3x3_nhwc_bf16(const void* in, const float* table, int silu,
                               const void* w, const float* bias, const void* resid, void* out, double* gn_part,
                               int B, int H, int W, int Cin, int Cout, varhip_stream_t stream);
int varhip_conv16_force_tile(int wm);
int varhip_gn_stats_bf16(const void* x, float* stats, double* scratch, int B, int HW, int C, int G, float eps, varhip_stream_t stream);
int varhip_gn_apply_bf16(const void* x, const float* stats, const float* gamma, const float* beta, void* out,
                        int B, int HW, int C, int G, int silu, varhip_stream_t stream);
int varhip_gn_silu_conv_out_bf16(const void* x, const float* stats, const float* gamma, const float* beta, const void* w, const float* bias,
                                 float* out, int B, int H, int W, int Cin, int Cout, int G, int out_mode, varhip_stream_t stream);
int varhip_cast_f32_to_bf16(const float* in, void* out, int64_t n, varhip_stream_t stream);
int varhip_cast_bf16_to_f32(const void* in, float* out, int64_t n, varhip_stream_t stream);


/* ---- per-kernel timing (bench.py's roofline leg) -----------------------------------------------------------
 * When enabled, every launch is bracketed by hipEvents on its own stream and its algorithmic FLOPs and bytes are
 * accumulated per kernel family.  varhip_timing_read synchronises the recorded events.
 * families: 0 gemm (the 128x128-tile instantiation k_dma_gemm<4,4,false,2,false>), 1 conv3x3 (the 128x160-tile implicit-GEMM
 * instantiation k_dma_gemm<4,5,true,2,false>), 2 attn, 3 sampler, 4 ln, 5 qkv_prep, 6 gn, 7 other, 8 gemm_small (every other tile
 * of the transformer GEMMs and the element-wise-load fallback), 9 conv_small (every other conv tile: nearest-2x gather, Cout not
 * a multiple of 160); the 16-bit mode's kernels in families of their own (one arithmetic type, hence one MFMA peak, per family):
 * 10 gemm16 (k_gemm16p, the persistent 256x256-tile kernel), 11 gemm16_small (every k_gemm16 tile), 12 conv16h (k_conv16h<5,32>),
 * 13 conv16_small (every other fp16 conv kernel), 14 attn16.  Families 0, 1, 10 and 12 each map to exactly one kernel symbol, so their
 * averages can be checked against a rocprofv3 kernel trace.  Returns the number of families. */
#define VARHIP_NFAM 15
int varhip_timing_enable(int on);
/* restrict the timing to the families whose bit is set (default: all).  Every timed launch costs two event records on the stream —
 * about 2 % of a sampling call when all ~3000 launches are timed; bench.py times only what its roofline object reports. */
int varhip_timing_select(int family_mask);
int varhip_timing_reset(void);
int varhip_timing_read(double* ms, double* flops, double* bytes, int64_t* launches);
const char* varhip_timing_name(int family);

#ifdef __cplusplus
}
#endif
#endif /* VAR_HIP_H */
