// smooth.hip — VAR.smooth_sampling (fork, reference models/var.py:367-572): neighbour table of the codebook and the per-row
// candidate selection.  Both are integer/index work on top of a few fp32 reductions; the reductions use the canonical W256
// order so the CPU twins in oracle/var_oracle.c reproduce every bit.
#include "common.h"

__device__ __forceinline__ unsigned long long sm_shfl_xor_u64(unsigned long long v, int off) {
    return ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off, 64) << 32) | (unsigned)__shfl_xor((unsigned)v, off, 64);
}

// ---- neighbour table (var.py:459-462): for every code v the n nearest codes by L2 distance, ascending (ties -> smaller index).
// The reference builds it with torch.cdist (|a|^2+|b|^2-2ab form, BLAS rounding) + an unstable argsort; here the distance is the
// direct form, one fma chain over the channels, and the order is total, so the table is reproducible anywhere.
// One 256-thread workgroup per code: V keys (dist_key << 32 | index) in LDS, bitonic sort, first n written out.
__global__ void __launch_bounds__(256) k_neighbor_table(const float* __restrict__ cb, int V, int D, int n, int V2,
                                                        int* __restrict__ nbr_idx, float* __restrict__ nbr_dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [V2]
    const int tid = threadIdx.x, v = blockIdx.x;
    const float* a = cb + (int64_t)v * D;
    for (int u = tid; u < V2; u += 256) {
        unsigned long long k = ~0ull;
        if (u < V) {
            const float* b = cb + (int64_t)u * D;
            float acc = 0.f;
            for (int c = 0; c < D; ++c) { const float d = a[c] - b[c]; acc = __builtin_fmaf(d, d, acc); }
            k = ((unsigned long long)vm_float_key(vm_sqrt(acc)) << 32) | (unsigned)u;
        }
        keys[u] = k;
    }
    __syncthreads();
    for (int k = 2; k <= V2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < V2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int c = tid; c < n; c += 256) {
        const unsigned u = (unsigned)keys[c];
        const float* b = cb + (int64_t)u * D;
        float acc = 0.f;
        for (int q = 0; q < D; ++q) { const float d = a[q] - b[q]; acc = __builtin_fmaf(d, d, acc); }
        nbr_idx[(int64_t)v * n + c] = (int)u;
        nbr_dist[(int64_t)v * n + c] = vm_sqrt(acc);
    }
}

extern "C" int varhip_neighbor_table_f32(const float* codebook, int V, int D, int n, int32_t* nbr_idx, float* nbr_dist, varhip_stream_t stream) {
    if (V <= 0 || V > 8192 || D <= 0 || n <= 0 || n > V) return VARHIP_EINVAL;
    int V2 = 2; while (V2 < V) V2 <<= 1;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_neighbor_table, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192); attr_done = true; }
    VhScope sc(VH_FAM_OTHER, (hipStream_t)stream, 3.0 * V * (double)V * D, 4.0 * V * D + 8.0 * V * n);
    hipLaunchKernelGGL(k_neighbor_table, dim3(V), dim3(256), sizeof(unsigned long long) * V2, (hipStream_t)stream, codebook, V, D, n, V2, nbr_idx, nbr_dist);
    return vh_launch_status();
}

// ---- per-row candidate selection (var.py:482-537).  One 256-thread workgroup per (image, token) row:
//   x = (1+t)*cond - t*uncond;  lp_v = (x_v - max x) - log(sum exp(x - max x))            (torch.log_softmax)
//   candidates c < n: the nearest neighbours of the ground-truth token; valid if c < cand_count (count mode) or
//   dist_c <= d_0 + (thr - d_0)*ratio (threshold mode); the valid candidate with the largest lp wins, first one on ties
//   (all masked -> candidate 0 with lp = -inf, the reference's fallback).
//   dist_lp = log_softmax(-dist over all n candidates)[winner].
__global__ void __launch_bounds__(256) k_smooth_select(const float* __restrict__ logits, const int64_t* __restrict__ gt,
                                                       const int* __restrict__ nbr_idx, const float* __restrict__ nbr_dist, int n,
                                                       int cand_count, int use_thr, float thr, float ratio, int64_t rows, int V, float ca, float cb,
                                                       int64_t* __restrict__ idx_out, float* __restrict__ maxval_out, float* __restrict__ distlp_out,
                                                       float* __restrict__ cfg_out) {
    __shared__ float red[4];
    __shared__ unsigned long long s_k[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float* lc = logits + row * V;
    const float* lu = logits + (rows + row) * V;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) { const float a = ca * lc[i]; const float b = cb * lu[i]; const float x = a - b; m = fmaxf(m, x); if (cfg_out) cfg_out[row * V + i] = x; }
    m = vh_block_max256(m, red);
    float part = 0.f;
    for (int i = tid; i < V; i += 256) { const float a = ca * lc[i]; const float b = cb * lu[i]; part = part + vm_exp((a - b) - m); }
    const float ls = vm_log(vh_block_sum256(part, red));

    const int64_t g = gt[row];
    const int* ni = nbr_idx + g * n;
    const float* nd = nbr_dist + g * n;
    const float d0 = nd[0];
    const float eff = d0 + (thr - d0) * ratio;
    unsigned long long best = 0ull;
    float dmax = -INFINITY;
    for (int c = tid; c < n; c += 256) {
        const int v = ni[c];
        const float a = ca * lc[v]; const float b = cb * lu[v];
        float lp = ((a - b) - m) - ls;
        const bool valid = use_thr ? (nd[c] <= eff) : (c < cand_count);
        if (!valid) lp = -INFINITY;
        const unsigned long long k = ((unsigned long long)vm_float_key(lp) << 32) | (0xFFFFFFFFu - (unsigned)c);
        best = k > best ? k : best;
        dmax = fmaxf(dmax, -nd[c]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = sm_shfl_xor_u64(best, off); best = o > best ? o : best; }
    if ((tid & 63) == 0) s_k[tid >> 6] = best;
    dmax = vh_block_max256(dmax, red);                            // (its barriers also publish s_k)
    for (int w = 0; w < 4; ++w) best = s_k[w] > best ? s_k[w] : best;
    const int cbest = (int)(0xFFFFFFFFu - (unsigned)best);
    float dpart = 0.f;
    for (int c = tid; c < n; c += 256) dpart = dpart + vm_exp(-nd[c] - dmax);
    const float dls = vm_log(vh_block_sum256(dpart, red));
    if (tid == 0) {
        const int v = ni[cbest];
        const float a = ca * lc[v]; const float b = cb * lu[v];
        float lp = ((a - b) - m) - ls;
        const bool valid = use_thr ? (nd[cbest] <= eff) : (cbest < cand_count);
        idx_out[row] = v;
        maxval_out[row] = valid ? lp : -INFINITY;
        distlp_out[row] = (-nd[cbest] - dmax) - dls;
    }
}

extern "C" int varhip_smooth_select_f32(const float* logits, const int64_t* gt, const int32_t* nbr_idx, const float* nbr_dist, int n,
                                        int cand_count, int use_thr, float thr, float ratio, int B, int l, int V, double t_cfg,
                                        int64_t* idx_out, float* maxval_out, float* distlp_out, float* cfg_out, varhip_stream_t stream) {
    if (B <= 0 || l <= 0 || V <= 0 || n <= 0 || n > V || (!use_thr && (cand_count < 1 || cand_count > n))) return VARHIP_EINVAL;
    const int64_t rows = (int64_t)B * l;
    VhScope sc(VH_FAM_SAMPLER, (hipStream_t)stream, 0, 4.0 * rows * V * 4.0);
    hipLaunchKernelGGL(k_smooth_select, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, gt, nbr_idx, nbr_dist, n, cand_count, use_thr,
                       thr, ratio, rows, V, (float)(1.0 + t_cfg), (float)t_cfg, idx_out, maxval_out, distlp_out, cfg_out);
    return vh_launch_status();
}
