// elem16.h — the element type of the 16-bit throughput mode.  gemm16.hip, attn16.hip, conv16.hip and rowops16.hip are compiled TWICE: as they
// stand for fp16 (the precision the reference's harness asks for, demo_sample.py:66-68) and with -DVH_BF16 for bfloat16 (the reference's
// other 16-bit option, utils/arg_util.py: `fp16: int  # 1: using fp16, 2: bf16`).  Same kernels, same data paths (a 16-bit element is a 16-bit
// element for LDS-DMA, swizzles and transposing reads); what changes is the MFMA opcode and the conversions at the rounding points.  Everything
// of a flavour lives in its own namespace; the extern "C" entry points carry the flavour in their name (varhip_gemm_nt_f16 / _bf16).
#pragma once
#ifdef VH_BF16
typedef __bf16 vh_e16;
#define VH16_NS vh_bf16
#define VH16_FN(stem) varhip_##stem##_bf16
#define VH16_FN_CAST_TO varhip_cast_f32_to_bf16
#define VH16_FN_CAST_FROM varhip_cast_bf16_to_f32
#define VH16_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define VH16_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VH16_MFMA_16x16x32_ASM "v_mfma_f32_16x16x32_bf16"
#else
typedef _Float16 vh_e16;
#define VH16_NS vh_f16
#define VH16_FN(stem) varhip_##stem##_f16
#define VH16_FN_CAST_TO varhip_cast_f32_to_f16
#define VH16_FN_CAST_FROM varhip_cast_f16_to_f32
#define VH16_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define VH16_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define VH16_MFMA_16x16x32_ASM "v_mfma_f32_16x16x32_f16"
#endif
// c += a . b with the accumulator pinned to ONE register tuple (dst == src C).  The builtin lets the register allocator rename an accumulator
// between two MFMAs (dst != src C); in a loop that holds 80 of its 128 registers in accumulators it did, failed to coalesce the loop-carried
// values and spilled them (k_conv16h).  Use only where the next reader of `c` is far behind the MFMA (>= 18 wait states: the compiler pads
// nothing for an instruction it cannot see; tools/check_kernel_asm.py checks the emitted code) — here, the same accumulator's next MFMA
// a whole tap (19 MFMAs) later, an in-place accumulate, which the hardware interlocks anyway.
#define VH16_MFMA_16x16x32_INPLACE(a, b, c) asm(VH16_MFMA_16x16x32_ASM " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
// test / experiment switches shared by both flavours (defined in timing.hip)
extern int vh_g_force_tile16, vh_g_gemm16_persist, vh_g_gemm16_deep, vh_g_conv16_force_wm;
