// common.h — shared plumbing of libvar_hip.so (gfx950 only; compile with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/var_hip.h"
#include "../../include/var_math.h"

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
// every lane ends with the same value: p[j] + p[j^off] is commutative, so both partners compute identical bits
__device__ __forceinline__ float vh_wave_sum(float p) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}
__device__ __forceinline__ float vh_wave_max(float p) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = fmaxf(p, __shfl_xor(p, off, 64));
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
