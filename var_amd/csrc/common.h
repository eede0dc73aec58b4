// common.h — shared plumbing of libvar_hip.so (gfx950 only; compile with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/var_hip.h"
#include "../../include/var_math.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Two exponentials at once with the packed fp32 ALU ops (v_pk_mul/fma/add_f32): element for element the operations of
// include/var_math.h's vm_exp in the same order, so the results are bit-identical to it for every non-NaN input (attention feeds it scores that are
// finite or -inf, the GELU epilogue finite pre-activations).  The scaling by 2^n uses v_ldexp_f32, exact like the multiplication it replaces (results are normal).
// About half of the attention kernel's non-matrix instructions were the 16 scalar exponentials per key tile.
__device__ __forceinline__ f32x2 vh_exp_pair(f32x2 x) {
    const f32x2 xc = {__builtin_fminf(x[0], 88.0f), __builtin_fminf(x[1], 88.0f)};
    const f32x2 t = xc * 1.44269504088896341f;
    const f32x2 n = {__builtin_rintf(t[0]), __builtin_rintf(t[1])};
    f32x2 r = __builtin_elementwise_fma(n, (f32x2)(-0.693145751953125f), xc);
    r = __builtin_elementwise_fma(n, (f32x2)(-1.42860682030941723212e-6f), r);
    f32x2 q = (f32x2)(1.9875691500e-4f);
    q = __builtin_elementwise_fma(q, r, (f32x2)(1.3981999507e-3f));
    q = __builtin_elementwise_fma(q, r, (f32x2)(8.3334519073e-3f));
    q = __builtin_elementwise_fma(q, r, (f32x2)(4.1665795894e-2f));
    q = __builtin_elementwise_fma(q, r, (f32x2)(1.6666665459e-1f));
    q = __builtin_elementwise_fma(q, r, (f32x2)(5.0000001201e-1f));
    const f32x2 y = __builtin_elementwise_fma(q, r * r, r) + 1.0f;
    f32x2 o;
    o[0] = (x[0] > -87.0f) ? __builtin_ldexpf(y[0], (int)n[0]) : 0.0f;
    o[1] = (x[1] > -87.0f) ? __builtin_ldexpf(y[1], (int)n[1]) : 0.0f;
    return o;
}

// vm_gelu_tanh (include/var_math.h) on two elements with packed multiplies/adds and the packed exponential; the two divisions stay
// scalar (correctly rounded, as the C expression x / d is).  Element for element the same operations in the same order.
__device__ __forceinline__ f32x2 vh_gelu_tanh_pair(f32x2 x) {
    const f32x2 x3 = (x * x) * x;
    const f32x2 u = (x + x3 * 0.044715f) * 0.7978845608028654f;
    const f32x2 d = vh_exp_pair(u * -2.0f) + 1.0f;
    return f32x2{x[0] / d[0], x[1] / d[1]};
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VH_FAM_GEMM 0
#define VH_FAM_CONV 1
#define VH_FAM_ATTN 2
#define VH_FAM_SAMPLER 3
#define VH_FAM_LN 4
#define VH_FAM_QKV 5
#define VH_FAM_GN 6
#define VH_FAM_OTHER 7
#define VH_FAM_GEMM_SMALL 8
#define VH_FAM_CONV_SMALL 9
// the 16-bit mode's kernels have families of their own: one arithmetic type (one MFMA peak) per family, and the three that can dominate a
// step map to one kernel each (k_gemm16p, k_conv16h<5,32>, k_attn16<NW>)
#define VH_FAM_GEMM16 10
#define VH_FAM_GEMM16_SMALL 11
#define VH_FAM_CONV16H 12
#define VH_FAM_CONV16_SMALL 13
#define VH_FAM_ATTN16 14

// ---- timing table (timing.cpp) ------------------------------------------------------------------------------------
int vh_timing_on(int fam);
void vh_timing_begin(int fam, hipStream_t s, double flops, double bytes);
void vh_timing_end(int fam, hipStream_t s);

struct VhScope {           // brackets one launch with events when timing is enabled
    int fam; hipStream_t s; bool on;
    VhScope(int f, hipStream_t st, double flops, double bytes) : fam(f), s(st), on(vh_timing_on(f) != 0) { if (on) vh_timing_begin(fam, s, flops, bytes); }
    ~VhScope() { if (on) vh_timing_end(fam, s); }
};

static inline int vh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

// ---- canonical reductions (CPU twins: canon_sum64 / canon_sum256 in oracle/var_oracle.c) -------------
// every lane ends with the same value: p[j] + p[j^off] is commutative, so both partners compute identical bits.
// The butterfly partners lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 are reached without the LDS crossbar (the compiler's __shfl_xor is a ds_bpermute: an
// LDS round trip per step, six dependent ones per sum): the half / row swaps of gfx950 for 32 and 16 (permlaneNN_swap on two copies of v leaves
// v in one result and the partner's v in the other: their sum / max is the step's result in both partners), DPP row operations for the rest
// (row_ror:8 IS lane ^ 8 inside a row of 16; lane ^ 4 = row_shl:4 for lanes with bit 2 clear, row_shr:4 for the others; quad_perm for 2 and 1).
// Same partners, same operand pairs as the shuffle form: bit-identical results (the LayerNorm / softmax / q-k-norm parity tests run on them).
template <int CTRL> __device__ __forceinline__ float vh_mov_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float vh_lane_xor4(float v) {            // value of lane ^ 4
    const float up = vh_mov_dpp<0x104>(v), dn = vh_mov_dpp<0x114>(v);      // row_shl:4 (from lane + 4), row_shr:4 (from lane - 4)
    return (threadIdx.x & 4) ? dn : up;
}
__device__ __forceinline__ float vh_wave_sum(float p) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false); p = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false); p = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    p = p + vh_mov_dpp<0x128>(p);                                    // row_ror:8
    p = p + vh_lane_xor4(p);
    p = p + vh_mov_dpp<0x4E>(p);                                     // quad_perm [2,3,0,1]
    p = p + vh_mov_dpp<0xB1>(p);                                     // quad_perm [1,0,3,2]
    return p;
}
__device__ __forceinline__ float vh_wave_max(float p) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false); p = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false); p = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    p = fmaxf(p, vh_mov_dpp<0x128>(p));
    p = fmaxf(p, vh_lane_xor4(p));
    p = fmaxf(p, vh_mov_dpp<0x4E>(p));
    p = fmaxf(p, vh_mov_dpp<0xB1>(p));
    return p;
}
// 256-thread block: wave butterflies, then ((w0+w1)+w2)+w3.  `red` is >= 4 floats of LDS; contains a barrier pair.
__device__ __forceinline__ float vh_block_sum256(float p, float* red) {
    p = vh_wave_sum(p);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
__device__ __forceinline__ float vh_block_max256(float p, float* red) {
    p = vh_wave_max(p);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
