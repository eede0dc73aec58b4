"""CPU oracle driver: the VAR sampling loop restated on numpy arrays over oracle/libvar_oracle.so.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product path (var_amd/, models/) must never do so.

It follows the reference's control flow line by line (citations inline; paths relative to the reference repo)
and delegates all arithmetic to the scalar C restatement in var_oracle.c.  Weights are taken in the
reference's state-dict layout (name -> ndarray), exactly what `VAR.state_dict()` / `VQVAE.state_dict()` hold.
"""
import ctypes
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from var_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libvar_oracle.so')
        if not os.path.exists(path):
            raise FileNotFoundError(f'{path} missing: run `make -C oracle` (or __graft_entry__.build())')
        so = ctypes.CDLL(path)
        _LIB = abi.bind(so, 'varref_', with_stream=False)
        kn = so.varref_gemm_kn_f32
        kn.argtypes = [abi.P, abi.L, abi.P, abi.L, abi.P, abi.P, abi.L, abi.I, abi.I, abi.I, abi.I, abi.P, abi.L, abi.P, abi.L, abi.I]
        kn.restype = abi.I
        _LIB['gemm_kn_f32'] = kn
        a16 = so.varref_attn_cached_p16_f32          # twin of the 16-bit mode's attention (p rounded to fp16 for p.v)
        a16.argtypes = [abi.P, abi.P, abi.P, abi.P, abi.I, abi.I, abi.I, abi.I, abi.I]
        a16.restype = abi.I
        _LIB['attn_cached_p16_f32'] = a16
        ab16 = so.varref_attn_cached_pbf16_f32       # ... and of its bfloat16 flavour
        ab16.argtypes = list(a16.argtypes)
        ab16.restype = abi.I
        _LIB['attn_cached_pbf16_f32'] = ab16
    return _LIB


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS'], 'oracle wants C-contiguous arrays'
    return a.ctypes.data_as(ctypes.c_void_p)


def _vp(a: Optional[np.ndarray]):
    """address of a (possibly strided) view's first element"""
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def r16(a: np.ndarray, kind: str = 'f16') -> np.ndarray:
    """round to fp16 (kind='f16') or bfloat16 ('bf16'), nearest-even, and back: the rounding points of the 16-bit throughput mode
    (include/var_hip.h, "f16" / "bf16")"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if kind == 'bf16':                               # the upper 16 bits of the fp32 pattern (finite values)
        u = a.view(np.uint32)
        return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return a.astype(np.float16).astype(np.float32)


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f'oracle {what} failed with code {rc}')


# ---------------------------------------------------------------------------------------------------------------------
def bicubic_taps(pn: int, P: int):
    """4-tap table of F.interpolate(mode='bicubic', align_corners=False) from size pn to P (quant.py:190).

    ATen upsample_bicubic2d: src = scale*(dst+0.5)-0.5 (not clamped), taps floor(src)-1..+2 index-clamped, Keys A=-0.75,
    all in fp32.  Returns (idx int32 [P,4], w float32 [P,4])."""
    A = np.float32(-0.75)
    one = np.float32(1.0)
    idx = np.zeros((P, 4), np.int32)
    w = np.zeros((P, 4), np.float32)
    scale = np.float32(pn) / np.float32(P)
    for o in range(P):
        src = scale * (np.float32(o) + np.float32(0.5)) - np.float32(0.5)
        i0 = np.floor(src)
        t = np.float32(src - i0)

        def c1(x):   # |x| <= 1
            return ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one

        def c2(x):   # 1 < |x| < 2
            return ((A * x - np.float32(5) * A) * x + np.float32(8) * A) * x - np.float32(4) * A
        w[o] = [c2(t + one), c1(t), c1(one - t), c2((one - t) + one)]
        for k in range(4):
            idx[o, k] = min(max(int(i0) - 1 + k, 0), pn - 1)
    return idx, w


def phi_index(si: int, S: int, K: int) -> int:
    """PhiPartiallyShared.__getitem__ (quant.py:218-226): argmin |ticks - si/(S-1)|."""
    ticks = np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)
    return int(np.argmin(np.abs(ticks - si / (S - 1))))


class OracleVAR:
    """Numpy/C restatement of VAR.autoregressive_infer_cfg (var.py:126-190) and VQVAE.fhat_to_img (vqvae.py:62-63)."""

    def __init__(self, var_sd: Dict[str, np.ndarray], vae_sd: Dict[str, np.ndarray], patch_nums: Sequence[int], depth: int,
                 attn_l2_norm: bool = True, shared_aln: bool = False, num_classes: int = 1000, norm_eps: float = 1e-6,
                 share_quant_resi: int = 4, quant_resi: float = 0.5, f16=False):
        """f16=True (or 'f16') / 'bf16': the CPU twin of the 16-bit throughput mode — the same fp32 restatement with the rounding points of
        that mode (16-bit block / head weights, LayerNorm output, q, k, v, attention probabilities and output, MLP hidden) in that flavour."""
        self.L_ = lib()
        self.f16 = bool(f16)
        self.kind16 = 'bf16' if f16 == 'bf16' else 'f16'
        self.sd = {k: (f32(v) if np.issubdtype(np.asarray(v).dtype, np.floating) else np.ascontiguousarray(v)) for k, v in var_sd.items()}
        self.vd = {k: f32(v) for k, v in vae_sd.items() if not k.startswith('encoder.') and not k.startswith('quant_conv.')}
        self.pns = tuple(patch_nums)
        self.depth, self.l2, self.saln = depth, attn_l2_norm, shared_aln
        self.C = self.sd['pos_start'].shape[-1]
        self.H = self.C // 64
        self.V, self.Cv = self.vd['quantize.embedding.weight'].shape
        self.num_classes, self.eps = num_classes, float(norm_eps)
        self.K_phi, self.ratio = share_quant_resi, abs(quant_resi)
        self.L = sum(p * p for p in self.pns)
        self._wt = {}
        assert self.sd['pos_1LC'].shape[1] == self.L

    # ---- small wrappers ---------------------------------------------------------------------------------------
    def _wT(self, key, w=None, w16=False):
        if key not in self._wt:
            w = self.sd[key] if w is None else w
            w = w.reshape(w.shape[0], -1)
            self._wt[key] = np.ascontiguousarray((r16(w, self.kind16) if w16 else w).T)
        return self._wt[key]

    def linear(self, x, wkey, bias, epi=abi.EPI_NONE, resid=None, gamma=None, ldg=0, rows_per_group=1, w=None, w16=False):
        """F.linear with the epilogues of include/var_hip.h; x: [M,K].  w16: the weight is rounded to fp16 first (16-bit mode)"""
        wt = self._wT(wkey, w, w16)
        K, N = wt.shape
        M = x.shape[0]
        assert x.shape[1] == K
        out = np.empty((M, N), np.float32)
        _ck(self.L_['gemm_kn_f32'](_p(x), K, _p(wt), N, _p(bias), _p(out), N, M, N, K, epi,
                                   _p(resid), N, _vp(gamma), ldg, rows_per_group), 'gemm')
        return out

    def ln_mod(self, x, scale, shift, rows_per_group):
        M, C = x.shape
        out = np.empty_like(x)
        # scale/shift are column-slice views of a row-major [2B, 6C] (or 2C) buffer: pass their base address + row stride
        _ck(self.L_['ln_modulate_f32'](_p(x), _vp(scale), scale.strides[0] // 4, _vp(shift), shift.strides[0] // 4,
                                       _p(out), M, C, rows_per_group, self.eps), 'ln_modulate')
        return out

    # ---- the loop ---------------------------------------------------------------------------------------------
    def ada(self, b: int, cond_silu: np.ndarray, shared: Optional[np.ndarray]):
        """AdaLNSelfAttn ada_lin / ada_gss (basic_var.py:153-156) -> [2B, 6C]"""
        if self.saln:
            out = np.empty_like(shared)
            g = f32(self.sd[f'blocks.{b}.ada_gss'].reshape(-1))
            _ck(self.L_['add_bcast_f32'](_p(g), _p(shared), _p(out), shared.shape[0], shared.shape[1]), 'add_bcast')
            return out
        return self.linear(cond_silu, f'blocks.{b}.ada_lin.1.weight', self.sd[f'blocks.{b}.ada_lin.1.bias'])

    def run(self, labels: Sequence[int], noises: List[np.ndarray], cfg: float, top_k: int, top_p: float,
            force_idx: Optional[np.ndarray] = None, decode: bool = True, keep_masked: bool = False,
            gt_tokens: Optional[np.ndarray] = None, keep_mask: Optional[np.ndarray] = None,
            more_smooth: bool = False, gumbel_noises: Optional[List[np.ndarray]] = None, smooth: Optional[dict] = None):
        """Returns dict(img, idx [B,L], logits [per scale 2B,l,V], f_hat [per scale, NCHW], pooled [per scale, NCHW]).
        smooth = dict(gt=[B,L] tokens, n=int, thr=float|None): VAR.smooth_sampling (var.py:367-572) — the sampler is replaced by
        the neighbour-candidate selection; adds sum_ll / sum_dist_ll / maxval / distlp to the result."""
        Lf, sd = self.L_, self.sd
        B, C, H, S = len(labels), self.C, self.H, len(self.pns)
        B2, P, Cv, V = 2 * B, self.pns[-1], self.Cv, self.V
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        lvl_pos = np.empty((self.L, C), np.float32)
        _ck(Lf['lvl_pos_f32'](_p(f32(sd['lvl_embed.weight'])), _p(np.ascontiguousarray(sd['lvl_1L'].reshape(-1), dtype=np.int64)),
                              _p(f32(sd['pos_1LC'].reshape(self.L, C))), _p(lvl_pos), self.L, C), 'lvl_pos')
        first_l = self.pns[0] ** 2
        cond = np.empty((B2, C), np.float32)
        x = np.empty((B2 * first_l, C), np.float32)
        _ck(Lf['first_map_f32'](_p(f32(sd['class_emb.weight'])), _p(labels), self.num_classes, _p(f32(sd['pos_start'].reshape(first_l, C))),
                                _p(lvl_pos), _p(cond), _p(x), B, C, first_l), 'first_map')
        cond_silu = np.empty_like(cond)
        _ck(Lf['silu_f32'](_p(cond), _p(cond_silu), cond.size), 'silu')
        f_hat = np.zeros((B, P, P, Cv), np.float32)                       # var.py:157 (channels-last here)
        up = np.empty_like(f_hat)
        kc = [np.zeros((B2, H, self.L, 64), np.float32) for _ in range(self.depth)]   # b.attn.kv_caching(True): var.py:159
        vc = [np.zeros((B2, H, self.L, 64), np.float32) for _ in range(self.depth)]
        codebook = self.vd['quantize.embedding.weight']
        out = dict(idx=[], logits=[], f_hat=[], pooled=[], masked=[])
        cur = 0
        draws = 0          # Exp(1) fills consumed so far (inpainting skips the draw on fully kept scales)
        for si, pn in enumerate(self.pns):                                 # var.py:160
            l = pn * pn
            shared = self.linear(cond_silu, 'shared_ada_lin.1.weight', sd['shared_ada_lin.1.bias']) if self.saln else None   # var.py:165
            for b in range(self.depth):                                    # var.py:168-169 -> AdaLNSelfAttn.forward basic_var.py:152-159
                ada = self.ada(b, cond_silu, shared)
                g1, g2, s1, s2, h1, h2 = (ada[:, i * C:(i + 1) * C] for i in range(6))
                h16 = self.f16
                hN = self.ln_mod(x, s1, h1, l)
                if h16: hN = r16(hN, self.kind16)
                bias_qkv = np.concatenate([sd[f'blocks.{b}.attn.q_bias'], np.zeros(C, np.float32), sd[f'blocks.{b}.attn.v_bias']]).astype(np.float32)
                qkv = self.linear(hN, f'blocks.{b}.attn.mat_qkv.weight', bias_qkv, w16=h16)          # basic_var.py:93
                q = np.empty((B2 * l, C), np.float32)
                sm = f32(sd[f'blocks.{b}.attn.scale_mul_1H11'].reshape(-1)) if self.l2 else None
                _ck(Lf['qkv_prep_f32'](_p(qkv), _p(sm), 0.25 / 8.0, int(self.l2), _p(q), _p(kc[b]), _p(vc[b]), B2, l, H, cur, self.L), 'qkv_prep')
                att = np.empty((B2 * l, C), np.float32)
                if h16:            # fp16 q and fp16 cache rows; p rounded to fp16 for p.v, fp16 output
                    k16 = self.kind16
                    q = r16(q, k16)
                    kc[b][:, :, cur:cur + l] = r16(kc[b][:, :, cur:cur + l], k16); vc[b][:, :, cur:cur + l] = r16(vc[b][:, :, cur:cur + l], k16)
                    _ck(Lf['attn_cached_pbf16_f32' if k16 == 'bf16' else 'attn_cached_p16_f32'](_p(q), _p(kc[b]), _p(vc[b]), _p(att), B2, l, H, cur + l, self.L), 'attn p16')
                else:
                    _ck(Lf['attn_cached_f32'](_p(q), _p(kc[b]), _p(vc[b]), _p(att), B2, l, H, cur + l, self.L), 'attn')
                x = self.linear(att, f'blocks.{b}.attn.proj.weight', sd[f'blocks.{b}.attn.proj.bias'], abi.EPI_RESID, resid=x,
                                gamma=g1, ldg=ada.shape[1], rows_per_group=l, w16=h16)                 # basic_var.py:157
                hN = self.ln_mod(x, s2, h2, l)
                if h16: hN = r16(hN, self.kind16)
                hid = self.linear(hN, f'blocks.{b}.ffn.fc1.weight', sd[f'blocks.{b}.ffn.fc1.bias'], abi.EPI_GELU, w16=h16)
                if h16: hid = r16(hid, self.kind16)
                x = self.linear(hid, f'blocks.{b}.ffn.fc2.weight', sd[f'blocks.{b}.ffn.fc2.bias'], abi.EPI_RESID, resid=x,
                                gamma=g2, ldg=ada.shape[1], rows_per_group=l, w16=h16)                 # basic_var.py:158
            cur += l
            if gt_tokens is not None and bool(np.all(keep_mask[:, cur - l:cur])):
                # VAR.inpainting, whole scale kept (var.py:312-313): ground-truth tokens, no logits, no RNG draw
                idx = np.ascontiguousarray(gt_tokens[:, cur - l:cur], dtype=np.int64)
                out['logits'].append(None); out['idx'].append(idx.copy())
            else:
                # get_logits: AdaLNBeforeHead + head (var.py:118-124, basic_var.py:172-174)
                hm = self.linear(cond_silu, 'head_nm.ada_lin.1.weight', sd['head_nm.ada_lin.1.bias'])
                hN = self.ln_mod(x, hm[:, :C], hm[:, C:], l)
                if self.f16: hN = r16(hN, self.kind16)
                logits = self.linear(hN, 'head.weight', sd['head.bias'], w16=self.f16)
                out['logits'].append(logits.reshape(B2, l, V))
                idx = np.empty((B * l,), np.int64)
                masked = np.empty((B * l, V), np.float32) if (keep_masked or more_smooth) else None
                t = cfg * (si / (S - 1))                                       # var.py:161,172
                if smooth is not None:                                         # var.py:484-537 (fork): candidates = neighbours of the gt token
                    n = int(smooth['n']); thr = smooth.get('thr'); ratio = si / (S - 1)
                    if 'nbr' not in self._wt:
                        ni = np.empty((V, n), np.int32); nd = np.empty((V, n), np.float32)
                        _ck(Lf['neighbor_table_f32'](_p(f32(codebook)), V, Cv, n, _p(ni), _p(nd)), 'neighbor_table')
                        self._wt['nbr'] = (n, ni, nd)
                    n0, ni, nd = self._wt['nbr']
                    assert n0 == n
                    gs = np.ascontiguousarray(smooth['gt'][:, cur - l:cur], dtype=np.int64).reshape(-1)
                    mv = np.empty(B * l, np.float32); dl = np.empty(B * l, np.float32)
                    _ck(Lf['smooth_select_f32'](_p(logits), _p(gs), _p(ni), _p(nd), n, 1 + int((n - 1) * ratio), int(thr is not None),
                                                float(thr if thr is not None else 0.0), float(ratio), B, l, V, float(t), _p(idx), _p(mv), _p(dl), _p(masked)), 'smooth_select')
                    out.setdefault('maxval', []).append(mv); out.setdefault('distlp', []).append(dl)
                    # var.py:537: new_tensor(max_vals) takes sampled_tokens' dtype (int64): each value is truncated toward zero before the sum
                    out['sum_ll'] = np.float32(out.get('sum_ll', np.float32(0.0)) + np.float32(np.trunc(mv.astype(np.float64)).sum()))
                    out['sum_dist_ll'] = np.float32(out.get('sum_dist_ll', np.float32(0.0)) + dl.sum(dtype=np.float32))
                else:
                    _ck(Lf['cfg_sample_f32'](_p(logits), _p(f32(noises[draws])), _p(idx), _p(masked), B, l, V, float(t), int(top_k), float(top_p)), 'cfg_sample')
                    draws += 1
                if gt_tokens is not None:                                      # torch.where(mask, gt, sampled): var.py:326-328
                    km = np.ascontiguousarray(keep_mask[:, cur - l:cur]).astype(np.uint8).reshape(-1)
                    gs = np.ascontiguousarray(gt_tokens[:, cur - l:cur], dtype=np.int64).reshape(-1)
                    _ck(Lf['token_select_i64'](_p(km), _p(gs), _p(idx), _p(idx), B * l), 'token_select')
                idx = idx.reshape(B, l)
                out['idx'].append(idx.copy())
                if keep_masked: out['masked'].append(masked.reshape(B, l, V))
            if force_idx is not None:                                      # teacher forcing for drift-free comparison
                idx = np.ascontiguousarray(force_idx[:, cur - l:cur], dtype=np.int64)
            # quantizer step (var.py:177-183, quant.py:187-196)
            if pn != P: ti, tw = bicubic_taps(pn, P)
            else: ti, tw = None, None
            k = phi_index(si, S, self.K_phi)
            pw = np.ascontiguousarray(self.vd[f'quantize.quant_resi.qresi_ls.{k}.weight'].transpose(0, 2, 3, 1))
            if more_smooth:                                                # var.py:178-180: gumbel softmax of the FILTERED logits @ codebook
                r_ = si / (S - 1)
                gum_t = max(0.27 * (1 - r_ * 0.95), 0.005)
                probs = np.empty((B * l, V), np.float32)
                _ck(Lf['gumbel_softmax_f32'](_p(masked), _p(f32(gumbel_noises[si])), _p(probs), B * l, V, float(1 + r_), float(gum_t)), 'gumbel_softmax')
                h = self.linear(probs, 'vae:codebook_T', None, w=np.ascontiguousarray(codebook.T))
                out.setdefault('h', []).append(h.reshape(B, l, Cv).copy())
                _ck(Lf['quant_accum_h_f32'](_p(h), _p(ti), _p(tw), _p(pw), _p(self.vd[f'quantize.quant_resi.qresi_ls.{k}.bias']),
                                            self.ratio, _p(up), _p(f_hat), B, pn, P, Cv), 'quant_accum_h')
            else:
                _ck(Lf['quant_accum_f32'](_p(idx), _p(codebook), _p(ti), _p(tw), _p(pw), _p(self.vd[f'quantize.quant_resi.qresi_ls.{k}.bias']),
                                          self.ratio, _p(up), _p(f_hat), B, pn, P, Cv), 'quant_accum')
            out['f_hat'].append(f_hat.transpose(0, 3, 1, 2).copy())
            if si != S - 1:
                pq = self.pns[si + 1]
                x = np.empty((B2 * pq * pq, C), np.float32)
                pooled = np.empty((B, pq * pq, Cv), np.float32)
                _ck(Lf['next_map_f32'](_p(f_hat), _p(f32(sd['word_embed.weight'])), _p(f32(sd['word_embed.bias'])), _p(lvl_pos[cur:]),
                                       _p(x), _p(pooled), B, P, pq, C, Cv), 'next_map')
                out['pooled'].append(pooled.reshape(B, pq, pq, Cv).transpose(0, 3, 1, 2).copy())
        out['idx'] = np.concatenate(out['idx'], axis=1)
        out['img'] = self.decode(f_hat) if decode else None
        return out

    # ---- VQVAE decoder (vqvae.py:62-63, basic_vae.py:163-226), channels-last ----------------------------------------
    def conv3(self, x, key, B, Hh, Ww, up2=0, resid=None, out_mode=0):
        w = self.vd[key + '.weight']
        Cout, Cin = w.shape[:2]
        wp = self._wt.get(key)
        if wp is None:
            wp = self._wt[key] = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
        out = np.empty((B, Cout, Hh, Ww) if out_mode else (B, Hh, Ww, Cout), np.float32)
        _ck(self.L_['conv3x3_nhwc_f32'](_p(x), _p(wp), _p(self.vd[key + '.bias']), _p(resid), _p(out), B, Hh, Ww, Cin, Cout, up2, out_mode), 'conv3x3')
        return out

    def gn(self, x, key, B, HW, silu):
        Cc = x.shape[-1]
        st = np.empty((B, 32, 2), np.float32)
        _ck(self.L_['gn_stats_f32'](_p(x), _p(st), None, B, HW, Cc, 32, 1e-6), 'gn_stats')
        out = np.empty_like(x)
        _ck(self.L_['gn_apply_f32'](_p(x), _p(st), _p(self.vd[key + '.weight']), _p(self.vd[key + '.bias']), _p(out), B, HW, Cc, 32, int(silu)), 'gn_apply')
        return out

    def conv1(self, x2d, key, epi=abi.EPI_NONE, resid=None):
        w = self.vd[key + '.weight']
        return self.linear(x2d, 'vae:' + key, self.vd[key + '.bias'], epi, resid=resid, w=w.reshape(w.shape[0], -1))

    def resblock(self, x, pre, B, Hh, Ww):
        """ResnetBlock.forward (basic_vae.py:57-60)"""
        HW = Hh * Ww
        h = self.conv3(self.gn(x, pre + '.norm1', B, HW, True), pre + '.conv1', B, Hh, Ww)
        if (pre + '.nin_shortcut.weight') in self.vd:
            sc = self.conv1(x.reshape(B * HW, -1), pre + '.nin_shortcut').reshape(B, Hh, Ww, -1)
        else:
            sc = x
        return self.conv3(self.gn(h, pre + '.norm2', B, HW, True), pre + '.conv2', B, Hh, Ww, resid=sc)

    def attnblock(self, x, pre, B, Hh, Ww):
        """AttnBlock.forward (basic_vae.py:73-92): single head over HW tokens"""
        HW, Cc = Hh * Ww, x.shape[-1]
        xn = self.gn(x, pre + '.norm', B, HW, False).reshape(B * HW, Cc)
        qkv = self.conv1(xn, pre + '.qkv').reshape(B, HW, 3 * Cc)
        hs = np.empty((B, HW, Cc), np.float32)
        scale = float(np.float32(int(Cc) ** (-0.5)))
        for b in range(B):
            q, k, v = (np.ascontiguousarray(qkv[b, :, i * Cc:(i + 1) * Cc]) for i in range(3))
            s = np.empty((HW, HW), np.float32)
            _ck(self.L_['gemm_nt_f32'](_p(q), Cc, _p(k), Cc, None, _p(s), HW, HW, HW, Cc, 0, None, 0, None, 0, 1, 0, 1, 0, 0, 0), 'bmm qk')
            p = np.empty_like(s)
            _ck(self.L_['softmax_rows_f32'](_p(s), _p(p), HW, HW, scale), 'softmax')
            vt = np.ascontiguousarray(v.T)                                  # [C][HW]
            o = np.empty((HW, Cc), np.float32)
            _ck(self.L_['gemm_nt_f32'](_p(p), HW, _p(vt), HW, None, _p(o), Cc, HW, Cc, HW, 0, None, 0, None, 0, 1, 0, 1, 0, 0, 0), 'bmm pv')
            hs[b] = o
        out = self.conv1(hs.reshape(B * HW, Cc), pre + '.proj_out', abi.EPI_RESID, resid=x.reshape(B * HW, Cc))
        return out.reshape(B, Hh, Ww, Cc)

    def decode(self, f_hat_nhwc: np.ndarray) -> np.ndarray:
        """fhat_to_img(f_hat).add_(1).mul_(0.5): [B,P,P,Cv] -> [B,3,16P,16P] in [0,1]"""
        B, P = f_hat_nhwc.shape[0], f_hat_nhwc.shape[1]
        Hh = Ww = P
        h = self.conv3(f32(f_hat_nhwc), 'post_quant_conv', B, Hh, Ww)
        h = self.conv3(h, 'decoder.conv_in', B, Hh, Ww)
        h = self.resblock(h, 'decoder.mid.block_1', B, Hh, Ww)
        h = self.attnblock(h, 'decoder.mid.attn_1', B, Hh, Ww)
        h = self.resblock(h, 'decoder.mid.block_2', B, Hh, Ww)
        nlev = 1 + max(int(k.split('.')[2]) for k in self.vd if k.startswith('decoder.up.'))
        for lev in reversed(range(nlev)):
            for ib in range(3):
                h = self.resblock(h, f'decoder.up.{lev}.block.{ib}', B, Hh, Ww)
                if f'decoder.up.{lev}.attn.{ib}.norm.weight' in self.vd:
                    h = self.attnblock(h, f'decoder.up.{lev}.attn.{ib}', B, Hh, Ww)
            if lev != 0:
                Hh, Ww = 2 * Hh, 2 * Ww
                h = self.conv3(h, f'decoder.up.{lev}.upsample.conv', B, Hh, Ww, up2=1)
        h = self.gn(h, 'decoder.norm_out', B, Hh * Ww, True)
        return self.conv3(h, 'decoder.conv_out', B, Hh, Ww, out_mode=1)
