"""ctypes signature table of the C ABI declared in include/var_hip.h.

One table, two consumers: var_amd/hip.py binds `varhip_<name>(..., stream)` in libvar_hip.so (the product),
the CPU checker under oracle/ binds `varref_<name>(...)` in its own shared library.  Keeping the argument
lists in one place is what lets the parity tests drive both libraries with identical arguments.
"""
import ctypes as C

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double

# name -> argument ctypes (without the trailing stream of the HIP flavour); all return int
SIGNATURES = {
    'gemm_nt_f32':       [P, L, P, L, P, P, L, I, I, I, I, P, L, P, L, I, I, I, L, L, L],
    'silu_f32':          [P, P, L],
    'add_bcast_f32':     [P, P, P, I, I],
    'ln_modulate_f32':   [P, P, L, P, L, P, I, I, I, F],
    'qkv_prep_f32':      [P, P, F, I, P, P, P, I, I, I, I, I],
    'attn_cached_f32':   [P, P, P, P, I, I, I, I, I],
    'cfg_sample_f32':    [P, P, P, P, I, I, I, D, I, D],
    'quant_accum_f32':   [P, P, P, P, P, P, F, P, P, I, I, I, I],
    'next_map_f32':      [P, P, P, P, P, P, I, I, I, I, I],
    'lvl_pos_f32':       [P, P, P, P, I, I],
    'first_map_f32':     [P, P, I, P, P, P, P, I, I, I],
    'conv3x3_nhwc_f32':  [P, P, P, P, P, I, I, I, I, I, I, I],
    'gn_stats_f32':      [P, P, P, I, I, I, I, F],
    'gn_apply_f32':      [P, P, P, P, P, I, I, I, I, I],
    'softmax_rows_f32':  [P, P, L, I, F],
    'nchw_to_nhwc_f32':  [P, P, I, I, I],
    'nhwc_to_nchw_f32':  [P, P, I, I, I],
    'nearest_code_f32':  [P, P, P, I, I, I],
    'nearest_code_cos_f32': [P, P, P, I, I, I],
    'token_select_i64':  [P, P, P, P, L],
    'conv3x3_s2_nhwc_f32': [P, P, P, P, I, I, I, I, I],
    'nchw_to_nhwc_pad_f32': [P, P, I, I, I, I],
    'area_pool_f32':     [P, P, I, I, I, I],
    'word_embed_f32':    [P, P, P, P, P, I, I, I, I],
    'quant_residual_f32': [P, P, P, P, P, P, F, P, P, P, I, I, I, I],
    'upconv_pack_f32':   [P, P, I, I],
    'upconv_phase_f32':  [P, P, P, P, I, I, I, I, I],
    'quant_accum_h_f32': [P, P, P, P, P, F, P, P, I, I, I, I],
    'gumbel_softmax_f32': [P, P, P, L, I, F, F],
    'gemm_qkv_f32':      [P, L, P, L, P, I, I, I, P, F, I, P, P, P, I, I, I, I, I],
    'conv3x3_gn_nhwc_f32': [P, P, P, P, P, P, I, I, I, I, I, I],
    'upconv_phase_gn_f32': [P, P, P, P, P, I, I, I, I, I],
    'gn_stats_part_f32': [P, P, I, I, I, I, I, F],
    'adaln_block_f32':   [P, P, P, P, P, P, P, L, P, P, P, F, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F],
    'neighbor_table_f32': [P, I, I, I, P, P],
    'smooth_select_f32': [P, P, P, P, I, I, I, F, F, I, I, I, D, P, P, P, P],
}

# the 16-bit throughput mode: HIP library only (its CPU twin is oracle/var_oracle.py with f16=True: the fp32 functions + rounding points)
SIGNATURES_F16 = {
    'gemm_nt_f16':       [P, L, P, L, P, P, L, I, I, I, I, I, P, L, I, P, L, I, I, L, L, L],
    'gemm_qkv_f16':      [P, L, P, L, P, I, I, I, P, F, I, P, P, P, I, I, I, I, I],
    'attn_cached_f16':   [P, P, P, P, I, I, I, I, I],
    'ln_modulate_f16out': [P, P, L, P, L, P, I, I, I, F],
    'adaln_block_f16':   [P, P, P, P, P, P, P, L, P, P, P, F, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F],
    'conv3x3_nhwc_f16':  [P, P, P, P, P, P, I, I, I, I, I, I],
    'upconv_phase_f16':  [P, P, P, P, P, I, I, I, I, I],
    'gnconv3x3_nhwc_f16': [P, P, I, P, P, P, P, P, I, I, I, I, I],
    'gn_stats_f16':      [P, P, P, I, I, I, I, F],
    'gn_apply_f16':      [P, P, P, P, P, I, I, I, I, I],
    'gn_silu_conv_out_f16': [P, P, P, P, P, P, P, I, I, I, I, I, I, I],
    'cast_f32_to_f16':   [P, P, L],
    'cast_f16_to_f32':   [P, P, L],
}

# entry points of the HIP library that have no oracle twin of their own (they are pinned against other entry points bit for bit)
SIGNATURES_HIP_ONLY = {
    'gn_scale_shift_f32': [P, P, P, P, I, I, I],
    'gn_silu_conv_out_f32': [P, P, P, P, P, P, P, I, I, I, I, I, I, I],
}

# ... and with bfloat16 storage: one entry point per _f16 entry point, same arguments (include/var_hip.h, "bf16")
SIGNATURES_BF16 = {k.replace('_f16out', '_bf16out').replace('_to_f16', '_to_bf16').replace('cast_f16_', 'cast_bf16_').replace('_f16', '_bf16'): v
                   for k, v in SIGNATURES_F16.items()}

EPI_NONE, EPI_GELU, EPI_RESID = 0, 1, 2
EINVAL = -1


def bind(lib, prefix: str, with_stream: bool):
    """Attach argtypes/restype to every ABI function of `lib`; raises AttributeError naming a missing symbol."""
    fns = {}
    table = dict(SIGNATURES)
    if with_stream:                      # the HIP library also carries the 16-bit mode
        table.update(SIGNATURES_F16)
        table.update(SIGNATURES_BF16)
        table.update(SIGNATURES_HIP_ONLY)
    for name, args in table.items():
        fn = getattr(lib, prefix + name)
        fn.argtypes = list(args) + ([P] if with_stream else [])
        fn.restype = I
        fns[name] = fn
    gs = getattr(lib, prefix + 'gn_scratch_elems')
    gs.argtypes = [I, I, I, I]
    gs.restype = L
    fns['gn_scratch_elems'] = gs
    cb = getattr(lib, prefix + 'conv_gn_blocks')
    cb.argtypes = [I, I, I, I]
    cb.restype = I
    fns['conv_gn_blocks'] = cb
    if with_stream:
        cf = getattr(lib, prefix + 'conv16_gn_fusable')
        cf.argtypes = [I, I, I, I, I]
        cf.restype = I
        fns['conv16_gn_fusable'] = cf
    ver = getattr(lib, prefix + 'version')
    ver.argtypes = []
    ver.restype = C.c_char_p
    fns['version'] = ver
    return fns
