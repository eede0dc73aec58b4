"""Per-kernel parity on a real MI355X: every entry point of libvar_hip.so against its CPU twin in oracle/ on the same
seeded inputs, called through the C ABI on both sides (include/var_hip.h; var_amd/abi.py holds the one signature table).

Bar: bit-exact (`exact=True`) for everything on the token path — GEMM, AdaLN, q/k/v prep, attention, sampler, quantizer
step — because the loop feeds its own tokens back; GroupNorm statistics (fp64 partial sums in a different order) and
therefore the decoder are held to 1e-5 absolute instead (north_star asks 1e-3 for pixels).
"""
import ctypes

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _setup():
    util.ensure_oracle_built()
    from oracle import var_oracle
    from var_amd import hip
    return var_oracle.lib(), hip


def both(name, args, outs):
    """Call varref_<name> on numpy copies and varhip_<name> on device copies of `args`.
    An arg may be (array, element_offset) to pass an interior pointer.  Returns ([hip outs], [ref outs])."""
    L, hip = _setup()
    ref_arrays, dev_arrays, ref_args, dev_args = [], [], [], []
    for a in args:
        off = 0
        if isinstance(a, tuple):
            a, off = a
        if isinstance(a, np.ndarray):
            ra = np.ascontiguousarray(a).copy()
            da = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            ref_arrays.append(ra); dev_arrays.append(da)
            ref_args.append(ctypes.c_void_p(ra.ctypes.data + off * ra.itemsize))
            dev_args.append(ctypes.c_void_p(da.data_ptr() + off * da.element_size()))
        else:
            ref_arrays.append(None); dev_arrays.append(None)
            ref_args.append(a); dev_args.append(a)
    rc = L[name](*ref_args)
    assert rc == 0, f'oracle {name} rc={rc}'
    hip.call(name, *dev_args)
    torch.cuda.synchronize()
    return [dev_arrays[i].cpu().numpy() for i in outs], [ref_arrays[i] for i in outs]


def check(name, got, want, exact=True, atol=0.0, rtol=0.0):
    ok, msg = util.diff_report(name, got, want, atol=0.0 if exact else atol, rtol=0.0 if exact else rtol)
    print(msg)
    assert ok, msg


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
def test_library_loads_and_reports_version():
    _, hip = _setup()
    assert 'gfx950' in hip.lib().version()
    assert torch.cuda.is_available()


def test_timing_table_counts_only_the_selected_families():
    """bench.py's roofline numbers come from this table: launches, algorithmic FLOPs and event time per kernel family, restricted
    to the families asked for (every timed launch costs two event records)"""
    _, hip = _setup()
    a = torch.randn(256, 128, device='cuda'); w = torch.randn(256, 128, device='cuda'); o = torch.empty(256, 256, device='cuda')
    x = torch.randn(1 << 16, device='cuda'); y = torch.empty_like(x)
    def work():
        for _ in range(3):
            hip.call('gemm_nt_f32', a, 128, w, 128, None, o, 256, 256, 256, 128, 0, None, 0, None, 0, 1, 0, 1, 0, 0, 0)
            hip.call('silu_f32', x, y, x.numel())
    hip.timing_reset(); hip.timing_enable(True); work(); hip.timing_enable(False)
    t = hip.timing_read()
    small = t['gemm_small']
    assert small['launches'] == 3 and small['flops'] == 3 * 2.0 * 256 * 256 * 128 and small['ms'] > 0 and t['other']['launches'] == 3
    hip.timing_reset(); hip.timing_enable(True, ['other']); work(); hip.timing_enable(False)
    t = hip.timing_read()
    assert t['other']['launches'] == 3 and t['gemm_small']['launches'] == 0 and t['gemm']['launches'] == 0
    hip.timing_reset(); hip.timing_enable(False, None); work()
    assert all(v['launches'] == 0 for v in hip.timing_read().values())
    # the two conv families follow the tile instantiation: Cout % 160 == 0 without the nearest-2x gather is 'conv3x3'
    xin = torch.randn(1, 8, 8, 32, device='cuda'); bias = torch.zeros(160, device='cuda'); wt = torch.randn(160, 9 * 32, device='cuda')
    out = torch.empty(1, 16, 16, 160, device='cuda')
    hip.timing_reset(); hip.timing_enable(True)
    hip.call('conv3x3_nhwc_f32', xin, wt, bias, None, out, 1, 8, 8, 32, 160, 0, 0)
    hip.call('conv3x3_nhwc_f32', xin, wt, bias, None, out, 1, 8, 8, 32, 64, 0, 0)
    hip.call('conv3x3_nhwc_f32', xin, wt, bias, None, out, 1, 16, 16, 32, 160, 1, 0)
    hip.timing_enable(False)
    t = hip.timing_read()
    assert t['conv3x3']['launches'] == 1 and t['conv_small']['launches'] == 2 and t['conv3x3']['flops'] == 2.0 * 64 * 160 * 9 * 32
    # the 16-bit mode's kernels are counted in families of their own (one arithmetic type = one MFMA peak per family)
    a16, w16 = torch.randn(512, 128, device='cuda').half(), torch.randn(512, 128, device='cuda').half()
    o16 = torch.empty(512, 512, device='cuda', dtype=torch.float16)
    hip.timing_reset(); hip.timing_enable(True)
    hip.call('gemm_nt_f16', a16, 128, w16, 128, None, o16, 512, 1, 512, 512, 128, 0, None, 0, 0, None, 0, 1, 1, 0, 0, 0)
    hip.lib().so.varhip_gemm16_force_tile(2)
    try: hip.call('gemm_nt_f16', a16, 128, w16, 128, None, o16, 512, 1, 512, 512, 128, 0, None, 0, 0, None, 0, 1, 1, 0, 0, 0)
    finally: hip.lib().so.varhip_gemm16_force_tile(-1)
    hip.timing_enable(False)
    t = hip.timing_read()
    assert t['gemm16_small']['launches'] == 1 and t['gemm16']['launches'] == 1 and t['gemm']['launches'] == 0 and t['gemm_small']['launches'] == 0      # (forced 256x256 on whole tiles: the persistent kernel)


@pytest.mark.parametrize('M,N,K', [(128, 128, 128), (4, 384, 128), (36, 512, 128), (300, 320, 640), (1152, 1024, 1024), (2048, 3072, 1024), (64, 4096, 256), (1, 128, 32), (130, 40, 8), (9, 128, 9), (25, 33, 25), (70, 70, 13), (70, 50, 64), (3, 52, 96), (200, 17, 32)])
@pytest.mark.parametrize('epi', [0, 1, 2])
def test_gemm_exact(M, N, K, epi):
    rng = np.random.default_rng(M * 7 + N * 3 + K + epi)
    A, W, bias = rnd(rng, M, K), rnd(rng, N, K, scale=0.05), rnd(rng, N, scale=0.1)
    resid = rnd(rng, M, N); groups = max(1, M // 4); rpg = (M + groups - 1) // groups
    gamma = rnd(rng, groups + 1, 2 * N)
    out = np.zeros((M, N), np.float32)
    args = [A, K, W, K, bias, out, N, M, N, K, epi, resid if epi == 2 else None, N, (gamma, N) if epi == 2 else None, 2 * N, rpg, 0, 1, 0, 0, 0]
    (g,), (w,) = both('gemm_nt_f32', args, [5])
    check(f'gemm {M}x{N}x{K} epi{epi}', g, w)


def test_gemm_random_shapes_exact():
    """30 seeded random shapes around the tile edges (ragged M and N, K on and off the 32-multiple fast path, every epilogue):
    every tile configuration, idle waves, the element-wise epilogue and the fallback kernel against the oracle, bit for bit"""
    rs = np.random.default_rng(20240)
    for case in range(30):
        M = int(rs.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 300, 513, 700]))
        N = int(rs.choice([1, 3, 4, 8, 15, 16, 20, 33, 60, 64, 68, 100, 128, 132, 260, 516]))
        K = int(rs.choice([32, 64, 96, 160, 256, 1024, 40, 7, 100]))
        epi = int(rs.integers(0, 3))
        rng = np.random.default_rng(case)
        A, W, bias = rnd(rng, M, K), rnd(rng, N, K, scale=0.05), rnd(rng, N, scale=0.1)
        resid = rnd(rng, M, N); rpg = int(rs.integers(1, M + 1)); G = (M + rpg - 1) // rpg
        gamma = rnd(rng, G, N)
        out = np.zeros((M, N), np.float32)
        args = [A, K, W, K, bias if case % 5 else None, out, N, M, N, K, epi, resid if epi == 2 else None, N, gamma if (epi == 2 and case % 3) else None, N, rpg, 0, 1, 0, 0, 0]
        (g,), (w,) = both('gemm_nt_f32', args, [5])
        check(f'gemm random #{case} {M}x{N}x{K} epi{epi}', g, w)


def test_gemm_batched_bias_per_row_and_shared_operand():
    rng = np.random.default_rng(5)
    Bt, M, N, K = 3, 96, 256, 64                    # V^T = Wv . x^T per sample: A shared (sA = 0), W per sample
    A, W, bias = rnd(rng, M, K), rnd(rng, Bt, N, K), rnd(rng, M)
    out = np.zeros((Bt, M, N), np.float32)
    (g,), (w,) = both('gemm_nt_f32', [A, K, W, K, bias, out, N, M, N, K, 0, None, 0, None, 0, 1, 1, Bt, 0, N * K, M * N], [5])
    check('gemm batched bias_per_row', g, w)


def test_gemm_rejects_bad_shapes():
    _, hip = _setup()
    from var_amd.hip import VarHipError
    a = torch.zeros(64, 36, device='cuda')
    with pytest.raises(VarHipError):
        hip.call('gemm_nt_f32', a, 36, a, 36, None, a, 64, 64, 64, 0, 0, None, 0, None, 0, 1, 0, 1, 0, 0, 0)      # K == 0
    with pytest.raises(VarHipError):
        hip.call('gemm_nt_f32', a, 36, a, 36, None, a, 64, 64, 64, 32, 2, None, 0, None, 0, 1, 0, 1, 0, 0, 0)     # RESID without resid
    # the DMA requests carry 32-bit offsets: operands beyond those limits are refused before anything is launched (var_hip.h)
    x = torch.zeros(1, 8, 8, 32, device='cuda'); w = torch.zeros(32, 9 * 32, device='cuda'); b = torch.zeros(32, device='cuda')
    with pytest.raises(VarHipError):
        hip.call('conv3x3_nhwc_f32', x, w, b, None, x, 1, 4096, 4096, 32, 32, 0, 0)      # one sample of 2 GiB: beyond the descriptor window
    q = torch.zeros(8, device='cuda')
    with pytest.raises(VarHipError):                                                      # A rows spanning 4 GiB
        hip.call('gemm_qkv_f32', a, 1 << 22, a, 64, q, 257, 64, 64, q, 1.0, 0, a, a, a, 257, 1, 1, 0, 1)

@pytest.mark.parametrize('M,C,rpg', [(4, 128, 1), (37, 1024, 9), (512, 1024, 4), (10, 1920, 5), (6, 2304, 2)])
def test_ln_modulate_exact(M, C, rpg):
    rng = np.random.default_rng(M + C)
    G = (M + rpg - 1) // rpg
    x = rnd(rng, M, C, scale=2.0) + 0.3
    ada = rnd(rng, G, 6 * C, scale=0.5)
    out = np.zeros_like(x)
    (g,), (w,) = both('ln_modulate_f32', [x, (ada, 2 * C), 6 * C, (ada, 4 * C), 6 * C, out, M, C, rpg, 1e-6], [5])
    check(f'ln_modulate {M}x{C}', g, w)


@pytest.mark.parametrize('B2,l,H,pos0,Lmax,l2', [(4, 1, 2, 0, 14, 1), (4, 9, 2, 5, 14, 1), (2, 16, 16, 14, 55, 1), (4, 4, 2, 1, 14, 0)])
def test_qkv_prep_exact(B2, l, H, pos0, Lmax, l2):
    rng = np.random.default_rng(B2 * l + H)
    C = 64 * H
    qkv = rnd(rng, B2 * l, 3 * C)
    sm = (np.log(4.0) + rnd(rng, H, scale=0.5)).astype(np.float32); sm[0] = 6.0      # exercises clamp_max(log 100)
    q = np.zeros((B2 * l, C), np.float32)
    kc = rnd(rng, B2, H, Lmax, 64); vc = rnd(rng, B2, H, Lmax, 64)                   # pre-existing cache content must survive
    outs_h, outs_r = both('qkv_prep_f32', [qkv, sm, 0.03125, l2, q, kc, vc, B2, l, H, pos0, Lmax], [4, 5, 6])
    for nm, g, w in zip(('q', 'kcache', 'vcache'), outs_h, outs_r):
        check(f'qkv_prep {nm}', g, w)


@pytest.mark.parametrize('B2,l,H,K,pos0,Lmax,l2', [(4, 1, 2, 128, 0, 14, 1), (4, 9, 2, 128, 5, 14, 1), (2, 16, 16, 1024, 14, 55, 1), (4, 4, 3, 192, 1, 14, 0),
                                                   (6, 100, 5, 320, 30, 130, 1), (128, 4, 16, 1024, 1, 5, 1), (3, 169, 1, 64, 0, 169, 1)])
def test_gemm_qkv_fused_epilogue_exact(B2, l, H, K, pos0, Lmax, l2):
    """The fused entry point is DEFINED as gemm_nt followed by qkv_prep: same bits, both tile shapes, ragged M, odd H (3C % 128 != 0)."""
    rng = np.random.default_rng(B2 * l + H + K)
    C, M = 64 * H, B2 * l
    A, W, bias = rnd(rng, M, K), rnd(rng, 3 * C, K, scale=0.05), rnd(rng, 3 * C, scale=0.1)
    sm = (np.log(4.0) + rnd(rng, H, scale=0.5)).astype(np.float32); sm[0] = 6.0
    q = np.zeros((M, C), np.float32)
    kc = rnd(rng, B2, H, Lmax, 64); vc = rnd(rng, B2, H, Lmax, 64)
    outs_h, outs_r = both('gemm_qkv_f32', [A, K, W, K, bias, M, C, K, sm if l2 else None, 0.03125, l2, q, kc, vc, B2, l, H, pos0, Lmax], [11, 12, 13])
    for nm, g, w in zip(('q', 'kcache', 'vcache'), outs_h, outs_r):
        check(f'gemm_qkv {nm}', g, w)


@pytest.mark.parametrize('B2,l,H,hidden,pos0,Lmax,l2', [(4, 9, 2, 512, 5, 14, 1), (2, 16, 3, 768, 0, 16, 0), (6, 1, 16, 4096, 0, 5, 1)])
def test_adaln_block_composite_exact(B2, l, H, hidden, pos0, Lmax, l2):
    """one AdaLNSelfAttn block behind a single call == its CPU twin (which is the same seven steps on the oracle's functions)"""
    rng = np.random.default_rng(B2 * l + H)
    C, M = 64 * H, B2 * l
    x = rnd(rng, M, C); ada = rnd(rng, B2, 6 * C, scale=0.3)
    ws = [np.zeros((M, C), np.float32) for _ in range(4)]; hid = np.zeros((M, hidden), np.float32)
    qkv_w = rnd(rng, 3 * C, C, scale=0.05); qkv_b = rnd(rng, 3 * C, scale=0.1)
    sm = (np.log(4.0) + rnd(rng, H, scale=0.5)).astype(np.float32)
    proj_w = rnd(rng, C, C, scale=0.05); proj_b = rnd(rng, C, scale=0.1)
    fc1_w = rnd(rng, hidden, C, scale=0.05); fc1_b = rnd(rng, hidden, scale=0.1)
    fc2_w = rnd(rng, C, hidden, scale=0.03); fc2_b = rnd(rng, C, scale=0.1)
    kc = rnd(rng, B2, H, Lmax, 64, scale=0.2); vc = rnd(rng, B2, H, Lmax, 64)
    if l2:                                           # cached keys of earlier scales are unit vectors in the real loop
        kc /= np.maximum(np.linalg.norm(kc, axis=-1, keepdims=True), 1e-6)
    args = [x, ws[0], ws[1], ws[2], ws[3], hid, ada, 6 * C, qkv_w, qkv_b, sm if l2 else None, 0.03125, l2, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b,
            kc, vc, B2, l, C, H, hidden, pos0, Lmax, 1e-6]
    outs_h, outs_r = both('adaln_block_f32', args, [0, 19, 20])
    for nm, g, w in zip(('x', 'kcache', 'vcache'), outs_h, outs_r):
        check(f'adaln_block {nm}', g, w)


@pytest.mark.parametrize('V,D,n', [(4096, 32, 6), (512, 8, 512), (300, 5, 17), (8192, 16, 3)])
def test_neighbor_table_exact(V, D, n):
    """smooth_sampling's neighbour table: ascending distance, ties by index (duplicated codes force ties), self first"""
    rng = np.random.default_rng(V + n)
    cb = rnd(rng, V, D)
    cb[7] = cb[3]; cb[V - 1] = cb[3]                                   # exact duplicates -> zero distances and ties
    ni = np.zeros((V, n), np.int32); nd = np.zeros((V, n), np.float32)
    (gi, gd), (wi, wd) = both('neighbor_table_f32', [cb, V, D, n, ni, nd], [4, 5])
    check('neighbor idx', gi, wi); check('neighbor dist', gd, wd)
    assert gi[3, 0] == 3 and (n < 3 or (gi[3, 1] == 7 and gi[3, 2] == V - 1)) and np.all(np.diff(gd, axis=1) >= 0)


@pytest.mark.parametrize('B,l,V,n,cc,thr', [(2, 9, 4096, 6, 3, None), (3, 4, 4096, 8, 8, 2.5), (2, 1, 512, 300, 300, None), (2, 5, 1024, 16, 1, 0.0), (1, 3, 4096, 5, 5, 1e9)])
def test_smooth_select_exact(B, l, V, n, cc, thr):
    rng = np.random.default_rng(B * l + n)
    D = 8
    cb = rnd(rng, V, D)
    ni = np.zeros((V, n), np.int32); nd = np.zeros((V, n), np.float32)
    L, _ = _setup()
    assert L['neighbor_table_f32'](ctypes.c_void_p(cb.ctypes.data), V, D, n, ctypes.c_void_p(ni.ctypes.data), ctypes.c_void_p(nd.ctypes.data)) == 0
    logits = rnd(rng, 2 * B * l, V, scale=3.0)
    gt = rng.integers(0, V, size=B * l).astype(np.int64)
    idx = np.zeros(B * l, np.int64); mv = np.zeros(B * l, np.float32); dl = np.zeros(B * l, np.float32); cfgo = np.zeros((B * l, V), np.float32)
    outs_h, outs_r = both('smooth_select_f32', [logits, gt, ni, nd, n, cc, int(thr is not None), float(thr or 0.0), 0.625, B, l, V, 0.9375, idx, mv, dl, cfgo],
                          [13, 14, 15, 16])
    for nm, g, w in zip(('idx', 'maxval', 'distlp', 'cfg logits'), outs_h, outs_r):
        check(f'smooth_select {nm}', g, w)


@pytest.mark.parametrize('B2,l,H,curL,Lmax', [(4, 1, 2, 1, 14), (4, 4, 2, 5, 14), (4, 9, 2, 14, 14), (2, 25, 2, 55, 55), (2, 100, 3, 255, 300), (2, 256, 2, 680, 680), (1, 169, 1, 424, 680)])
def test_attn_cached_exact(B2, l, H, curL, Lmax):
    rng = np.random.default_rng(l * 3 + curL)
    C = 64 * H
    q = rnd(rng, B2 * l, C, scale=0.6)
    kc = rnd(rng, B2, H, Lmax, 64, scale=0.5); vc = rnd(rng, B2, H, Lmax, 64)
    kc[:, :, curL:] = np.nan; vc[:, :, curL:] = np.nan        # anything beyond curL must never be read into the result
    out = np.zeros((B2 * l, C), np.float32)
    (g,), (w,) = both('attn_cached_f32', [q, kc, vc, out, B2, l, H, curL, Lmax], [3])
    check(f'attn l={l} curL={curL}', g, w)


@pytest.mark.parametrize('l,curL', [(324, 536), (576, 1112), (1024, 2240)])
@pytest.mark.parametrize('H', [1, 36])
def test_attn_cached_exact_at_d36_512_shapes(l, curL, H):
    """BASELINE.json configs[4] (VAR-d36 512x512: patch_nums up to 32, L = 2240, 36 heads): the three largest scales' (l, curL), which
    no d16-sized case reaches — 8 to 32 workgroups of queries per head, 17 to 70 key tiles, a 2240-row cache stride — against the oracle, bit for bit"""
    rng = np.random.default_rng(l + curL + H)
    B2, Lmax, C = (2 if H == 1 else 1), 2240, 64 * H
    q = rnd(rng, B2 * l, C, scale=0.6)
    kc = rnd(rng, B2, H, Lmax, 64, scale=0.5); vc = rnd(rng, B2, H, Lmax, 64)
    kc[:, :, curL:] = np.nan; vc[:, :, curL:] = np.nan
    out = np.zeros((B2 * l, C), np.float32)
    (g,), (w,) = both('attn_cached_f32', [q, kc, vc, out, B2, l, H, curL, Lmax], [3])
    check(f'attn d36-512 l={l} curL={curL} H={H}', g, w)


def test_attn_random_ragged_shapes_exact():
    """12 seeded random (l, curL) pairs off the 32/128 tile edges, wide score ranges included"""
    rs = np.random.default_rng(77)
    for case in range(12):
        l = int(rs.choice([1, 2, 5, 31, 32, 33, 63, 65, 100, 127, 129, 169, 200]))
        curL = l + int(rs.choice([0, 1, 14, 31, 32, 33, 100, 255]))
        B2, H = int(rs.integers(1, 4)), int(rs.integers(1, 4))
        rng = np.random.default_rng(1000 + case)
        q = rnd(rng, B2 * l, 64 * H, scale=float(rs.choice([0.3, 1.0, 4.0])))
        kc = rnd(rng, B2, H, curL, 64, scale=0.7); vc = rnd(rng, B2, H, curL, 64)
        out = np.zeros((B2 * l, 64 * H), np.float32)
        (g,), (w,) = both('attn_cached_f32', [q, kc, vc, out, B2, l, H, curL, curL], [3])
        check(f'attn random #{case} l={l} curL={curL} B2={B2} H={H}', g, w)


def test_attn_peaked_rows():
    """one key dominating (softmax ~ one-hot) and large negative scores: exp underflow path (vm_exp -> 0)"""
    rng = np.random.default_rng(11)
    B2, l, H, curL = 2, 40, 2, 200
    q = rnd(rng, B2 * l, 128, scale=6.0); kc = rnd(rng, B2, H, curL, 64, scale=6.0); vc = rnd(rng, B2, H, curL, 64)
    out = np.zeros((B2 * l, 128), np.float32)
    (g,), (w,) = both('attn_cached_f32', [q, kc, vc, out, B2, l, H, curL, curL], [3])
    check('attn peaked', g, w)


@pytest.fixture(params=[0, 1], ids=['parallel_cut', 'walk'])
def sampler_mode(request):
    """the sampler decides the top-p cut by a parallel prefix sum wherever that provably equals the sequential fp64 walk (the definition) and by
    the walk itself otherwise; the test hook forces the walk everywhere — both must reproduce the oracle bit for bit"""
    _, hip = _setup()
    hip.lib().so.varhip_sampler_force_walk(request.param)
    yield request.param
    hip.lib().so.varhip_sampler_force_walk(0)


def test_cfg_sample_cut_exactly_on_the_threshold(sampler_mode):
    """all logits equal: every probability is exactly 1/V and the running sum j/V is exact; top_p is chosen so that 1 - top_p IS such a value
    (100/4096): the entry whose running sum equals the threshold is removed ('<='), the next one is not.  Second case: the threshold half
    an ulp-of-float above a reachable sum, and a threshold no sum reaches before the last entry (everything but the largest goes)."""
    V, B, l = 4096, 1, 3
    rng = np.random.default_rng(3)
    noise = rng.exponential(1.0, (B * l, V)).astype(np.float32)
    lg = np.full((2 * B * l, V), 0.75, np.float32)
    for top_p in (1.0 - 100.0 / 4096.0, 1.0 - 100.5 / 4096.0, 1.0 - 4095.0 / 4096.0, 1e-9):
        idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
        (gi, gm), (wi, wm) = both('cfg_sample_f32', [lg, noise, idx, masked, B, l, V, 0.0, 0, top_p], [2, 3])
        check(f'threshold cut top_p={top_p} kept-set', np.isfinite(gm), np.isfinite(wm)); check('threshold cut tokens', gi, wi)
    assert int(np.isfinite(wm).sum(1).min()) >= 1


@pytest.mark.parametrize('B,l,V,t,top_k,top_p,scale', [(2, 5, 4096, 0.75, 900, 0.96, 2.5), (2, 5, 4096, 0.0, 0, 0.0, 2.5), (3, 4, 4096, 1.5, 1, 0.0, 2.5),
                                                       (2, 5, 4096, 0.3, 0, 0.5, 2.5), (2, 3, 4096, 4.0, 50, 0.999, 8.0), (1, 7, 512, 1.0, 100, 0.9, 2.0),
                                                       (1, 2, 8192, 0.5, 8192, 0.0001, 3.0), (2, 64, 4096, 1.5, 900, 0.96, 2.0)])
def test_cfg_sample_exact(B, l, V, t, top_k, top_p, scale, sampler_mode):
    rng = np.random.default_rng(V + l + top_k)
    logits = rnd(rng, 2 * B * l, V, scale=scale)
    noise = rng.exponential(1.0, (B * l, V)).astype(np.float32)
    idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
    (gi, gm), (wi, wm) = both('cfg_sample_f32', [logits, noise, idx, masked, B, l, V, t, top_k, top_p], [2, 3])
    check('sampler kept-set', np.isfinite(gm), np.isfinite(wm))
    check('sampler masked logits', gm, wm)
    check('sampler tokens', gi, wi)


@pytest.mark.parametrize('V,top_k,top_p,n_above,n_tie', [(4096, 900, 0.96, 500, 2000), (4096, 900, 0.3, 899, 3197), (4096, 900, 0.96, 0, 4096),
                                                         (768, 0, 0.9, 0, 0), (4096, 100, 0.999, 99, 300)])
def test_cfg_sample_tie_crowd_beyond_sort_buffer(V, top_k, top_p, n_above, n_tie, sampler_mode):
    """more exact ties with the k-th value than the kernel's sort buffer holds (cap = pow2 >= top_k): the tie group is walked in
    index order unsorted.  Also a vocabulary that is not a power of two.  t = 0, unconditional half zero: x == cond exactly."""
    rng = np.random.default_rng(V + n_tie)
    B, l = 2, 3
    lg = rnd(rng, B * l, V, scale=2.0)
    if n_tie:
        for r in range(B * l):
            perm = rng.permutation(V)
            lg[r, perm[:n_above]] = np.abs(lg[r, perm[:n_above]]) + 1.5          # strictly above the tie value
            lg[r, perm[n_above:n_above + n_tie]] = 1.25                           # the crowd
            rest = perm[n_above + n_tie:]
            lg[r, rest] = -np.abs(lg[r, rest]) - 0.5
    two = np.concatenate([lg, np.zeros_like(lg)], 0)
    noise = rng.exponential(1.0, (B * l, V)).astype(np.float32)
    idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
    (gi, gm), (wi, wm) = both('cfg_sample_f32', [two, noise, idx, masked, B, l, V, 0.0, top_k, top_p], [2, 3])
    check('tie crowd kept-set', np.isfinite(gm), np.isfinite(wm))
    check('tie crowd masked logits', gm, wm)
    check('tie crowd tokens', gi, wi)
    if n_tie:
        assert np.isfinite(wm).sum(axis=1).min() >= 1


def test_cfg_sample_random_quantised_logits_exact(sampler_mode):
    """14 seeded random sampler configurations on logits rounded to a coarse grid (exact ties everywhere: at the top-k threshold,
    inside the top-p walk, at the arg-max), several vocabulary sizes, t != 0 so the CFG combine creates the ties' values"""
    rs = np.random.default_rng(99)
    for case in range(14):
        V = int(rs.choice([256, 512, 768, 1024, 4096])); B = int(rs.integers(1, 3)); l = int(rs.integers(1, 6))
        top_k = int(rs.choice([0, 1, 7, 100, 255, V // 2, V])); top_p = float(rs.choice([0.0, 0.1, 0.5, 0.9, 0.96, 0.9999]))
        grid = float(rs.choice([0.5, 0.125, 1.0]))
        rng = np.random.default_rng(3000 + case)
        logits = (np.round(rnd(rng, 2 * B * l, V, scale=2.0) / grid) * grid).astype(np.float32)
        noise = rng.exponential(1.0, (B * l, V)).astype(np.float32)
        idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
        (gi, gm), (wi, wm) = both('cfg_sample_f32', [logits, noise, idx, masked, B, l, V, 0.5, top_k, top_p], [2, 3])
        check(f'sampler random #{case} V={V} k={top_k} p={top_p} kept-set', np.isfinite(gm), np.isfinite(wm))
        check(f'sampler random #{case} masked', gm, wm)
        check(f'sampler random #{case} tokens', gi, wi)


def test_cfg_sample_ties_and_golden(golden_dir, sampler_mode):
    """rows full of exact ties, and the reference's own sampler fixtures (tests/golden/sampler.npz) straight through the HIP kernel"""
    import json
    z = np.load(f'{golden_dir}/sampler.npz')
    cases = json.loads(str(z['meta']))
    for ci, c in enumerate(cases):
        B, l, V = c['B'], c['l'], c['V']
        lg = z[f'logits_{ci}'].reshape(B * l, V)
        two = np.concatenate([lg, np.zeros_like(lg)], 0)
        idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
        (gi, gm), (wi, wm) = both('cfg_sample_f32', [two, z[f'noise_{ci}'], idx, masked, B, l, V, 0.0, c['top_k'], c['top_p']], [2, 3])
        check(f'golden sampler case {ci} masked (vs oracle)', gm, wm)
        check(f'golden sampler case {ci} tokens (vs oracle)', gi, wi)
        if c['kind'] != 'ties':
            check(f'golden sampler case {ci} tokens (vs reference)', gi.reshape(B, l).astype(np.int32), z[f'idx_{ci}'])


@pytest.mark.parametrize('B,pn,P', [(2, 1, 3), (2, 2, 3), (2, 3, 3), (3, 5, 16), (2, 13, 16), (2, 16, 16)])
def test_quant_step_exact(B, pn, P):
    from oracle.var_oracle import bicubic_taps
    rng = np.random.default_rng(pn * 17 + P)
    V, Cv, C = 4096, 32, 128
    idx = rng.integers(0, V, (B, pn * pn)).astype(np.int64)
    cb = rnd(rng, V, Cv); pw = rnd(rng, Cv, 3, 3, Cv, scale=0.1); pb = rnd(rng, Cv, scale=0.05)
    f_hat = rnd(rng, B, P, P, Cv); up = np.zeros_like(f_hat)
    ti, tw = bicubic_taps(pn, P) if pn != P else (None, None)
    (gu, gf), (wu, wf) = both('quant_accum_f32', [idx, cb, ti, tw, pw, pb, 0.5, up, f_hat, B, pn, P, Cv], [7, 8])
    check(f'quant up {pn}->{P}', gu, wu); check(f'quant f_hat {pn}->{P}', gf, wf)
    pq = max(1, min(P, pn + 1))
    ww, wb, lp = rnd(rng, C, Cv, scale=0.2), rnd(rng, C, scale=0.1), rnd(rng, pq * pq, C)
    x = np.zeros((2 * B * pq * pq, C), np.float32); pooled = np.zeros((B, pq * pq, Cv), np.float32)
    (gx, gp), (wx, wp) = both('next_map_f32', [wf, ww, wb, lp, x, pooled, B, P, pq, C, Cv], [4, 5])
    check(f'next_map pooled P={P}->{pq}', gp, wp); check('next_map x', gx, wx)


def test_gumbel_softmax_quant_h_and_token_select_exact():
    """more_smooth / inpainting helpers: gumbel softmax (shared vm_log), quantizer step from embeddings, token select"""
    from oracle.var_oracle import bicubic_taps
    rng = np.random.default_rng(77)
    rows, V = 12, 4096
    x = rnd(rng, rows, V, scale=3.0); x[:, ::3] = -np.inf                    # filtered logits: a third masked
    noise = rng.exponential(1.0, (rows, V)).astype(np.float32); noise[0, 1] = 1e-30; noise[0, 5] = 80.0
    for mul, tau in ((1.0, 0.27), (1.5, 0.14), (2.0, 0.0135)):
        y = np.zeros_like(x)
        (g,), (w,) = both('gumbel_softmax_f32', [x, noise, y, rows, V, mul, tau], [2]); check(f'gumbel softmax tau={tau}', g, w)
        assert abs(float(w.sum(1).max()) - 1) < 1e-4
    B, pn, P, Cv = 2, 5, 16, 32
    h = rnd(rng, B, pn * pn, Cv); pw = rnd(rng, Cv, 3, 3, Cv, scale=0.1); pb = rnd(rng, Cv, scale=0.05)
    f_hat = rnd(rng, B, P, P, Cv); up = np.zeros_like(f_hat)
    ti, tw = bicubic_taps(pn, P)
    (gu, gf), (wu, wf) = both('quant_accum_h_f32', [h, ti, tw, pw, pb, 0.5, up, f_hat, B, pn, P, Cv], [6, 7])
    check('quant_accum_h up', gu, wu); check('quant_accum_h f_hat', gf, wf)
    n = 1000
    keep = (rng.random(n) < 0.4).astype(np.uint8); gt = rng.integers(0, 4096, n).astype(np.int64); sm = rng.integers(0, 4096, n).astype(np.int64)
    out = np.zeros(n, np.int64)
    (g,), (w,) = both('token_select_i64', [keep, gt, sm, out, n], [3]); check('token_select', g, w)
    assert np.array_equal(w, np.where(keep.astype(bool), gt, sm))


def test_encode_side_kernels_exact():
    """stride-2 conv (Downsample2x), padded NCHW->NHWC, area pool, word embed, residual quantiser step vs their CPU twins"""
    from oracle.var_oracle import bicubic_taps
    rng = np.random.default_rng(123)
    for (B, H, W, Cin, Cout) in [(2, 8, 8, 32, 64), (1, 40, 24, 160, 160), (2, 5, 5, 64, 32)]:
        x = rnd(rng, B, 2 * H, 2 * W, Cin); w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
        out = np.zeros((B, H, W, Cout), np.float32)
        (g,), (r,) = both('conv3x3_s2_nhwc_f32', [x, w, bias, out, B, H, W, Cin, Cout], [3]); check(f'conv s2 {Cin}->{Cout} {H}x{W}', g, r)
    img = rnd(rng, 2, 3, 30); o = np.zeros((2, 30, 16), np.float32)
    (g,), (r,) = both('nchw_to_nhwc_pad_f32', [img, o, 2, 3, 30, 16], [1]); check('nchw_to_nhwc_pad', g, r)
    assert np.array_equal(r[:, :, 3:], np.zeros((2, 30, 13), np.float32)) and np.array_equal(r[:, :, :3], img.transpose(0, 2, 1))
    B, P, Cv, C = 3, 16, 32, 128
    f = rnd(rng, B, P, P, Cv)
    for pq in (1, 3, 5, 13, 16):
        pooled = np.zeros((B, pq * pq, Cv), np.float32)
        (g,), (r,) = both('area_pool_f32', [f, pooled, B, P, pq, Cv], [1]); check(f'area_pool ->{pq}', g, r)
        ww, wb, lp = rnd(rng, C, Cv, scale=0.2), rnd(rng, C, scale=0.1), rnd(rng, pq * pq, C)
        xo = np.zeros((2 * B * pq * pq, C), np.float32)
        (g2,), (r2,) = both('word_embed_f32', [r, ww, wb, lp, xo, B, pq * pq, C, Cv], [4]); check(f'word_embed lq={pq * pq}', g2, r2)
    V, pn = 4096, 6
    idx = rng.integers(0, V, (B, pn * pn)).astype(np.int64); cb = rnd(rng, V, Cv); pw = rnd(rng, Cv, 3, 3, Cv, scale=0.1); pb = rnd(rng, Cv, scale=0.05)
    f_hat, f_rest, up = rnd(rng, B, P, P, Cv), rnd(rng, B, P, P, Cv), np.zeros((B, P, P, Cv), np.float32)
    ti, tw = bicubic_taps(pn, P)
    (gh, gr), (rh, rr) = both('quant_residual_f32', [idx, cb, ti, tw, pw, pb, 0.5, up, f_hat, f_rest, B, pn, P, Cv], [8, 9])
    check('quant_residual f_hat', gh, rh); check('quant_residual f_rest', gr, rr)
    assert np.allclose((rh - f_hat), -(rr - f_rest), atol=1e-6)


def test_prologue_and_small_ops_exact():
    rng = np.random.default_rng(3)
    B, C, L, S = 3, 128, 14, 3
    lvl = np.concatenate([np.full(p * p, i) for i, p in enumerate((1, 2, 3))]).astype(np.int64)
    le, pos = rnd(rng, S, C), rnd(rng, L, C)
    out = np.zeros((L, C), np.float32)
    (g,), (w,) = both('lvl_pos_f32', [le, lvl, pos, out, L, C], [3]); check('lvl_pos', g, w)
    ce = rnd(rng, 1001, C); labels = np.array([0, 999, 1000], np.int64); ps = rnd(rng, 1, C)
    cond = np.zeros((2 * B, C), np.float32); x = np.zeros((2 * B, C), np.float32)
    (gc, gx), (wc, wx) = both('first_map_f32', [ce, labels, 1000, ps, w, cond, x, B, C, 1], [5, 6])
    check('first_map cond', gc, wc); check('first_map x', gx, wx)
    v = rnd(rng, 5000, scale=4.0); y = np.zeros_like(v)
    (g,), (w2,) = both('silu_f32', [v, y, v.size], [1]); check('silu', g, w2)
    base, cnd = rnd(rng, 768), rnd(rng, 6, 768); o = np.zeros_like(cnd)
    (g,), (w3,) = both('add_bcast_f32', [base, cnd, o, 6, 768], [2]); check('add_bcast', g, w3)
    a = rnd(rng, 2, 32, 25); o = np.zeros((2, 25, 32), np.float32)
    (g,), (w4,) = both('nchw_to_nhwc_f32', [a, o, 2, 32, 25], [1]); check('nchw_to_nhwc', g, w4)
    o2 = np.zeros_like(a)
    (g,), (w5,) = both('nhwc_to_nchw_f32', [w4, o2, 2, 32, 25], [1]); check('nhwc_to_nchw', g, w5)
    assert np.array_equal(w5, a)


@pytest.mark.parametrize('B,H,W,Cin,Cout,up2,resid,mode', [(2, 3, 3, 32, 32, 0, 0, 0), (2, 16, 16, 32, 640, 0, 0, 0), (1, 16, 16, 640, 640, 0, 1, 0),
                                                          (2, 32, 32, 128, 64, 1, 0, 0), (1, 64, 64, 320, 320, 1, 0, 0), (2, 24, 24, 160, 160, 0, 1, 0),
                                                          (2, 48, 48, 32, 3, 0, 0, 1), (1, 40, 40, 160, 3, 0, 0, 1), (3, 5, 7, 64, 128, 0, 1, 0)])
def test_conv3x3_exact(B, H, W, Cin, Cout, up2, resid, mode):
    rng = np.random.default_rng(H * W + Cin + Cout)
    Hi, Wi = (H // 2, W // 2) if up2 else (H, W)
    x = rnd(rng, B, Hi, Wi, Cin); w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
    rs = rnd(rng, B, H, W, Cout) if resid else None
    out = np.zeros((B, Cout, H, W) if mode else (B, H, W, Cout), np.float32)
    (g,), (wv,) = both('conv3x3_nhwc_f32', [x, w, bias, rs, out, B, H, W, Cin, Cout, up2, mode], [4])
    check(f'conv3x3 {Cin}->{Cout} {H}x{W} up{up2} mode{mode}', g, wv)


def test_conv3x3_random_shapes_exact():
    """16 seeded random convolutions: ragged images (pixel counts off the 128-row tile), channel counts off the 16/32-wide tiles,
    all input modes (plain, nearest-2x gather, stride 2), residual on and off, the NCHW clamp/denorm output"""
    rs = np.random.default_rng(4242)
    for case in range(16):
        B = int(rs.integers(1, 4)); H = int(rs.choice([2, 4, 6, 10, 16, 22])); W = int(rs.choice([2, 4, 8, 14, 16, 30]))
        Cin = int(rs.choice([32, 64, 96, 160])); Cout = int(rs.choice([1, 3, 5, 16, 20, 33, 64, 100, 160]))
        kind = int(rs.integers(0, 4))                   # 0 plain, 1 gather, 2 stride-2, 3 NCHW clamp output
        rng = np.random.default_rng(500 + case)
        w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
        if kind == 2:
            x = rnd(rng, B, 2 * H, 2 * W, Cin); out = np.zeros((B, H, W, Cout), np.float32)
            (g,), (wv,) = both('conv3x3_s2_nhwc_f32', [x, w, bias, out, B, H, W, Cin, Cout], [3])
        else:
            up2 = 1 if kind == 1 else 0
            x = rnd(rng, B, H // 2 if up2 else H, W // 2 if up2 else W, Cin)
            mode = int(rs.integers(1, 3)) if kind == 3 else 0
            res = rnd(rng, B, H, W, Cout) if (mode == 0 and case % 2) else None
            out = np.zeros((B, Cout, H, W) if mode else (B, H, W, Cout), np.float32)
            (g,), (wv,) = both('conv3x3_nhwc_f32', [x, w, bias, res, out, B, H, W, Cin, Cout, up2, mode], [4])
        check(f'conv random #{case} kind{kind} {Cin}->{Cout} {H}x{W} B{B}', g, wv)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 6, 6, 32, 32), (1, 32, 32, 640, 640), (2, 64, 48, 160, 160), (1, 16, 16, 64, 128)])
def test_upconv_phase(B, H, W, Cin, Cout):
    """Upsample2x as four 2x2 phase convs: bit-exact vs its CPU twin, and equal to the plain nearest-2x + 3x3 conv up to the
    rounding of the pre-summed weights (tolerance; the decoder is off the token path)"""
    rng = np.random.default_rng(H + W + Cin)
    x = rnd(rng, B, H // 2, W // 2, Cin); w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
    wp = np.zeros((4, Cout, 2, 2, Cin), np.float32)
    (gp,), (rp,) = both('upconv_pack_f32', [w, wp, Cin, Cout], [1]); check('upconv pack', gp, rp)
    out = np.zeros((B, H, W, Cout), np.float32)
    (g,), (r,) = both('upconv_phase_f32', [x, rp, bias, out, B, H, W, Cin, Cout], [3]); check('upconv phase vs twin', g, r)
    (g9,), (r9,) = both('conv3x3_nhwc_f32', [x, w, bias, None, np.zeros_like(out), B, H, W, Cin, Cout, 1, 0], [4])
    check('upconv phase vs 9-tap definition', g, r9, exact=False, atol=5e-5, rtol=1e-5)      # 1.9e-5 measured at K = 9*640, |out| <= 4.6


@pytest.mark.parametrize('B,H,W,Cin,Cout,resid', [(2, 16, 16, 32, 640, 0), (2, 32, 32, 320, 320, 1), (1, 32, 64, 160, 160, 1), (3, 16, 8, 64, 64, 0), (2, 16, 16, 32, 32, 1)])
def test_conv3x3_with_groupnorm_partials(B, H, W, Cin, Cout, resid):
    """the conv that also leaves per-block channel sums for the GroupNorm after it: same output bits as the plain conv, partials and
    the statistics made from them equal to the CPU twin's up to fp64 summation order, and to the stand-alone statistics kernel"""
    L, hip = _setup()
    rng = np.random.default_rng(H * W + Cin + Cout)
    nblk = hip.conv_gn_blocks(H, W, Cout)
    assert nblk == H * W // 128 and L['conv_gn_blocks'](H, W, Cout, 0) == nblk and hip.conv_gn_blocks(10, 10, Cout) == 0
    x = rnd(rng, B, H, W, Cin); w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
    r = rnd(rng, B, H, W, Cout) if resid else None
    out = np.zeros((B, H, W, Cout), np.float32); part = np.zeros((B, nblk, Cout, 2), np.float64)
    (g, gp), (wv, wp) = both('conv3x3_gn_nhwc_f32', [x, w, bias, r, out, part, B, H, W, Cin, Cout, 0], [4, 5])
    check('conv+gn output', g, wv)
    (g0,), _ = both('conv3x3_nhwc_f32', [x, w, bias, r, np.zeros_like(out), B, H, W, Cin, Cout, 0, 0], [4])
    check('conv+gn output vs plain conv', g, g0)
    check('gn partials', gp, wp, exact=False, atol=1e-9, rtol=1e-12)
    st = np.zeros((B, 32, 2), np.float32)
    (gs,), (ws,) = both('gn_stats_part_f32', [gp, st, B, nblk, H * W, Cout, 32, 1e-6], [1])
    check('stats from partials', gs, ws, exact=False, atol=1e-6, rtol=2e-6)
    scratch = np.zeros(hip.gn_scratch_elems(B, H * W, Cout, 32), np.float64)
    (g1,), _ = both('gn_stats_f32', [g, np.zeros_like(st), scratch, B, H * W, Cout, 32, 1e-6], [1])
    check('stats from partials vs statistics kernel', gs, g1, exact=False, atol=1e-6, rtol=2e-6)


def test_upconv_phase_with_groupnorm_partials():
    L, hip = _setup()
    rng = np.random.default_rng(77)
    B, H, W, Cin, Cout = 2, 32, 32, 64, 160
    nblk = hip.conv_gn_blocks(H, W, Cout, phase=True)
    assert nblk == 4 * (H // 2) * (W // 2) // 128
    x = rnd(rng, B, H // 2, W // 2, Cin); w = rnd(rng, Cout, 3, 3, Cin, scale=(1.0 / (9 * Cin)) ** 0.5); bias = rnd(rng, Cout, scale=0.1)
    wp = np.zeros((4, Cout, 2, 2, Cin), np.float32)
    _, (rp,) = both('upconv_pack_f32', [w, wp, Cin, Cout], [1])
    out = np.zeros((B, H, W, Cout), np.float32); part = np.zeros((B, nblk, Cout, 2), np.float64)
    (g, gp), (wv, wpp) = both('upconv_phase_gn_f32', [x, rp, bias, out, part, B, H, W, Cin, Cout], [3, 4])
    check('upconv+gn output', g, wv)
    check('upconv gn partials', gp, wpp, exact=False, atol=1e-9, rtol=1e-12)
    st = np.zeros((B, 32, 2), np.float32)
    (gs,), _ = both('gn_stats_part_f32', [gp, st, B, nblk, H * W, Cout, 32, 1e-6], [1])
    scratch = np.zeros(hip.gn_scratch_elems(B, H * W, Cout, 32), np.float64)
    (g1,), _ = both('gn_stats_f32', [g, np.zeros_like(st), scratch, B, H * W, Cout, 32, 1e-6], [1])
    check('upconv stats from partials vs statistics kernel', gs, g1, exact=False, atol=1e-6, rtol=2e-6)


@pytest.mark.parametrize('B,HW,C', [(2, 9, 32), (2, 256, 640), (1, 4096, 320), (2, 2304, 160), (3, 100, 64)])
def test_groupnorm(B, HW, C):
    _, hip = _setup()
    rng = np.random.default_rng(HW + C)
    x = rnd(rng, B, HW, C, scale=1.7) + 0.4
    st = np.zeros((B, 32, 2), np.float32)
    scratch = np.zeros(hip.gn_scratch_elems(B, HW, C, 32), np.float64)
    (g,), (w,) = both('gn_stats_f32', [x, st, scratch, B, HW, C, 32, 1e-6], [1])
    check('gn stats', g, w, exact=False, atol=1e-6, rtol=2e-6)
    gam, bet = 1 + rnd(rng, C, scale=0.2), rnd(rng, C, scale=0.2)
    for silu in (0, 1):
        out = np.zeros_like(x)
        (go,), (wo,) = both('gn_apply_f32', [x, w, gam, bet, out, B, HW, C, 32, silu], [4])
        # the affine part is exact; the fused SiLU uses the hardware exp2/rcp (decoder only, off the token path): ~1e-7 relative
        check(f'gn apply silu={silu} (same stats)', go, wo, exact=not silu, atol=2e-6, rtol=2e-6)


def test_softmax_rows_exact():
    rng = np.random.default_rng(9)
    x = rnd(rng, 300, 256, scale=30.0); out = np.zeros_like(x)
    (g,), (w,) = both('softmax_rows_f32', [x, out, 300, 256, 640 ** -0.5], [1]); check('softmax rows', g, w)
    x = rnd(rng, 7, 9, scale=3.0); out = np.zeros_like(x)
    (g,), (w,) = both('softmax_rows_f32', [x, out, 7, 9, 1.0], [1]); check('softmax rows n=9', g, w)


@pytest.mark.parametrize('name', ['nearest_code_f32', 'nearest_code_cos_f32'])
@pytest.mark.parametrize('N,V,Cv', [(333, 4096, 32), (1, 4096, 32), (130, 1000, 32), (4097, 512, 32), (50, 4096, 16), (7, 300, 8)])
def test_nearest_code_exact(name, N, V, Cv):
    """f_to_idxBl_or_fhat's arg-min / arg-max (quant.py:151-157), both variants: the MFMA kernel (Cvae = 32: 128 queries per workgroup,
    512-code shards in LDS; partial last workgroup, partial last shard and block) and the fallback for other widths; exact ties inside a
    lane's codes, across the two lane halves of a query, across blocks and across shards — the first index must win"""
    rng = np.random.default_rng(21 + N + V)
    z, cb = rnd(rng, N, Cv, scale=1.5), rnd(rng, V, Cv)
    for dup in (9, 77, min(600, V - 1)):                # code 5 again at 9 (other lane half), 77 (another block), 600 (another shard)
        cb[dup] = cb[5]
    z[0] = cb[77]
    if N > 40: z[40] = cb[5] * 1.0
    idx = np.zeros(N, np.int64)
    (g,), (w,) = both(name, [z, cb, idx, N, V, Cv], [2]); check(name, g, w)
    assert g[0] == 5 and (N <= 40 or g[40] == 5)


def test_nearest_code_at_encode_sizes():
    """every scale of a B=8 256x256 encode (N = 8 * pn^2 up to 2048 rows against the 4096 x 32 codebook), data with many near-ties
    (queries are codebook rows plus small noise)"""
    rng = np.random.default_rng(5)
    cb = rnd(rng, 4096, 32)
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        N = 8 * pn * pn
        z = cb[rng.integers(0, 4096, N)] + rnd(rng, N, 32, scale=0.3)
        idx = np.zeros(N, np.int64)
        for name in ('nearest_code_f32', 'nearest_code_cos_f32'):
            (g,), (w,) = both(name, [z, cb, idx, N, 4096, 32], [2]); check(f'{name} pn={pn}', g, w)


def test_vm_exp_matches_cpu_bit_for_bit():
    """the shared exp of include/var_math.h through SiLU on a dense sweep incl. the clamp/underflow edges"""
    v = np.concatenate([np.linspace(-100, 100, 200001), [-87.0, -86.99999, 88.0, 88.5, 0.0, -0.0, np.inf, -np.inf]]).astype(np.float32)
    y = np.zeros_like(v)
    (g,), (w,) = both('silu_f32', [v, y, v.size], [1])
    check('silu sweep', g, w)


@pytest.mark.parametrize('B,H,W,Cin,Cout,omode', [(2, 8, 32, 32, 3, 1), (1, 64, 64, 160, 3, 2), (3, 16, 96, 64, 3, 1), (1, 24, 32, 96, 4, 2), (2, 16, 32, 64, 1, 1), (2, 256, 256, 160, 3, 1)])
def test_gn_silu_conv_out_f32_fused_equals_two_launches(B, H, W, Cin, Cout, omode):
    """the fp32 decoder's tail (norm_out -> swish -> conv_out -> clamp, basic_vae.py:224-226) in one pass (one output pixel per thread, the
    convolution as per-thread fma chains in the chunk / tap / channel order) against varhip_gn_apply_f32 + varhip_conv3x3_nhwc_f32: identical
    bits — the image border and every patch edge included; the conv itself is pinned against the oracle by test_conv3x3_exact"""
    from var_amd import hip
    g = torch.Generator().manual_seed(H * 7 + W + Cin + Cout)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.3 + 0.2).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin) ** 0.5)).cuda()
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    gamma, beta = (torch.randn(Cin, generator=g) * 0.2 + 1.0).cuda(), (torch.randn(Cin, generator=g) * 0.2).cuda()
    stats = torch.empty(B, 32, 2, dtype=torch.float32, device='cuda')
    scratch = torch.empty(hip.gn_scratch_elems(B, H * W, Cin, 32), dtype=torch.float64, device='cuda')
    hip.call('gn_stats_f32', x, stats, scratch, B, H * W, Cin, 32, 1e-6)
    fused = torch.full((B, Cout, H, W), float('nan'), dtype=torch.float32, device='cuda')
    hip.call('gn_silu_conv_out_f32', x, stats, gamma, beta, w, bias, fused, B, H, W, Cin, Cout, 32, omode)
    xn = torch.empty_like(x)
    hip.call('gn_apply_f32', x, stats, gamma, beta, xn, B, H * W, Cin, 32, 1)
    two = torch.empty_like(fused)
    hip.call('conv3x3_nhwc_f32', xn, w, bias, None, two, B, H, W, Cin, Cout, 0, omode)
    assert torch.equal(fused, two), f'fused tail differs from the two launches in {int((fused != two).sum())} elements, max {float((fused - two).abs().max()):.3e}'
    if B * H * W <= 70000:
        ref = torch.nn.functional.conv2d(xn.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), bias.double().cpu(), padding=1).clamp(-1, 1)
        if omode == 1: ref = (ref + 1) * 0.5
        assert float((fused.double().cpu() - ref).abs().max()) <= 1e-5
    with pytest.raises(Exception):
        hip.call('gn_silu_conv_out_f32', x, stats, gamma, beta, w, bias, fused, B, H, W - 1, Cin, Cout, 32, omode)


def test_decoder_f32_fused_tail_equals_unfused():
    """VQVAE.fhat_to_img (d16-size decoder, 256x256, B=2): the one-pass tail and the two-launch tail give the same image bit for bit"""
    from tests.test_f16_gpu import _models
    z, meta = util.load_case('d16_full')
    vae, var = _models(meta)
    g = torch.Generator().manual_seed(4)
    f_hat = (torch.randn(2, 32, 16, 16, generator=g) * 1.5).cuda()
    eng = vae._decoder_engine()
    with torch.inference_mode():
        a = vae.fhat_to_img(f_hat).clone()
        eng.unfused_tail = True
        try: b = vae.fhat_to_img(f_hat).clone()
        finally: eng.unfused_tail = False
    assert torch.equal(a, b) and torch.isfinite(a).all() and float(a.abs().max()) <= 1.0


def test_cfg_sample_randomised_stress_exact(sampler_mode):
    """120 random sampler configurations against the oracle, bit for bit: V in {512, 4096, 8192}, top_k from 0 to V, top_p from 1e-7 to 1 - 1e-7 and
    0, CFG scales, logit spreads from 0.01 (near-uniform rows: long removed prefixes) to 30 (one dominant token), logits quantised to a few
    values in a third of the cases (tie groups across the top-k and the top-p cuts), rows whose whole mass sits on one token"""
    rng = np.random.default_rng(20261005)
    for case in range(120):
        V = int(rng.choice([512, 4096, 4096, 4096, 8192]))
        B, l = 1, int(rng.integers(1, 5))
        top_k = min(int(rng.choice([0, 1, 2, 50, 900, V // 2, V])), V)
        top_p = float(rng.choice([0.0, 1e-7, 0.01, 0.5, 0.9, 0.96, 0.999, 1.0 - 1e-7]))
        t = float(rng.choice([0.0, 0.3, 1.5, 4.0]))
        scale = float(rng.choice([0.01, 0.5, 2.5, 8.0, 30.0]))
        logits = rnd(rng, 2 * B * l, V, scale=scale)
        if case % 3 == 0:
            q = float(rng.choice([0.25, 1.0, 4.0]))
            logits = (np.round(logits / q) * q).astype(np.float32)
        if case % 10 == 7:
            logits[:, 1:] = -1e30; logits[:, 0] = 1.0                        # one token carries all the mass (the others' exponentials are exactly 0)
        noise = rng.exponential(1.0, (B * l, V)).astype(np.float32)
        idx = np.zeros(B * l, np.int64); masked = np.zeros((B * l, V), np.float32)
        (gi, gm), (wi, wm) = both('cfg_sample_f32', [logits, noise, idx, masked, B, l, V, t, top_k, top_p], [2, 3])
        tag = f'case {case}: V={V} top_k={top_k} top_p={top_p} t={t} scale={scale}'
        check(tag + ' kept-set', np.isfinite(gm), np.isfinite(wm))
        check(tag + ' tokens', gi, wi)
