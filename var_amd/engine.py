"""Host side of the MI355X sampling path: drives the HIP kernels of libvar_hip.so for one (VAR, VQVAE) pair.

This is what `VAR.autoregressive_infer_cfg` (reference models/var.py:126-190) and `VQVAE.fhat_to_img`
(models/vqvae.py:62-63) run on a GPU.  PyTorch supplies device memory, the RNG stream and the current HIP stream;
every floating-point operation of the path happens in a kernel reached through `hip.call` (include/var_hip.h).
There is no CPU / eager fallback here: without the library `hip.lib()` raises.

Differences from the reference's schedule (results unchanged):
  * the per-block AdaLN projection `ada_lin(cond)` is computed once per call instead of once per scale
    (it only depends on the class embedding: basic_var.py:156, var.py:165-169);
  * the KV cache is pre-allocated [2B, H, L, 64] per block and appended in place (no torch.cat, basic_var.py:107-109);
  * activations of the decoder are channels-last; f_hat is kept [B, P, P, Cvae] and only transposed at the API edge.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import hip
from .abi import EPI_GELU, EPI_NONE, EPI_RESID


def bicubic_taps(pn: int, P: int):
    """4-tap index/weight table of F.interpolate(mode='bicubic', align_corners=False) for pn -> P (reference quant.py:190).
    Follows ATen's upsample_bicubic2d in fp32: src = (pn/P)*(dst+0.5)-0.5, taps floor(src)-1..+2 clamped, Keys kernel A=-0.75."""
    f = np.float32
    A = f(-0.75)
    scale = f(pn) / f(P)
    src = scale * (np.arange(P, dtype=np.float32) + f(0.5)) - f(0.5)
    i0 = np.floor(src)
    t = (src - i0).astype(np.float32)

    def near(x):
        return ((A + f(2)) * x - (A + f(3))) * x * x + f(1)

    def far(x):
        return ((A * x - f(5) * A) * x + f(8) * A) * x - f(4) * A
    w = np.stack([far(t + f(1)), near(t), near(f(1) - t), far((f(1) - t) + f(1))], axis=1).astype(np.float32)
    idx = np.clip(i0.astype(np.int64)[:, None] - 1 + np.arange(4)[None, :], 0, pn - 1).astype(np.int32)
    return idx, w


def phi_index(si: int, S: int, K: int) -> int:
    """which shared Phi conv serves scale si (reference quant.py:218-226)"""
    ticks = np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)
    return int(np.argmin(np.abs(ticks - si / (S - 1)))) if S > 1 else 0


PRECISIONS = ('f32', 'f16', 'bf16')
DT16 = {'f16': torch.float16, 'bf16': torch.bfloat16}      # storage type of the two 16-bit flavours (include/var_hip.h "f16" / "bf16")


def _ver(p: torch.Tensor) -> int:
    """version counter of a parameter; tensors created under torch.inference_mode() have none"""
    try:
        return p._version
    except RuntimeError:
        return -1


def _signature(params) -> tuple:
    """what the engines key their re-laid weight copies on: storage address + version counter of every parameter.  In-place writes
    through `.data` (p.data.mul_(), p.data.copy_()) bump no counter: after such surgery call `engine.invalidate()` (VAR / VQVAE
    do it from load_state_dict and init_weights)."""
    return tuple((p.data_ptr(), _ver(p)) for p in params)


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise hip.VarHipError(f'{name}: the MI355X sampling path needs fp32 CUDA parameters, got {t.dtype} on {t.device}')
    return t if t.is_contiguous() else t.contiguous()


class _VaeOps:
    """channels-last building blocks shared by the decoder and encoder engines (reference basic_vae.py:18-92)"""

    PREFIXES = ()

    def __init__(self, vae):
        self.vae = vae
        self._sig = None
        self.w: Dict[str, torch.Tensor] = {}
        self._gn_part = None            # (tensor, partial sums, blocks per sample) left by the last conv for the GroupNorm after it
        self._ready = None              # event recorded behind the kernels that built the packed copies (they run on whichever stream called first)

    def _signature(self):
        return _signature(self.vae.parameters())

    def _built(self):
        """packed copies were just (re)built on the current stream: calls arriving on OTHER streams wait for this event before they read them"""
        self._ready = torch.cuda.Event()
        self._ready.record()

    def _wait_ready(self):
        if self._ready is not None:
            torch.cuda.current_stream().wait_event(self._ready)

    def _retire(self):
        """weights changed: before the old packed copies go back to the allocator, every stream that may still read them has to finish (a weight
        change is rare — checkpoint load, EMA swap — so a device-wide wait is the simple safe form)"""
        if self.w and torch.cuda.is_available():
            torch.cuda.synchronize()

    def _stats_from(self, x, B, HW, fn_full):
        """(mean, rstd) per (sample, group) of a channels-last map: folded from the producing conv's per-block partial sums when it left them for
        exactly this tensor, else by a statistics pass (`fn_full`) over x"""
        Cc = x.shape[-1]
        stats = torch.empty((B, 32, 2), dtype=torch.float32, device=x.device)
        pend, self._gn_part = self._gn_part, None
        if pend is not None and pend[0].data_ptr() == x.data_ptr() and pend[0].numel() == x.numel() and tuple(pend[1].shape) == (B, pend[2], Cc, 2):
            hip.call('gn_stats_part_f32', pend[1], stats, B, pend[2], HW, Cc, 32, 1e-6)
        else:
            scratch = torch.empty(hip.gn_scratch_elems(B, HW, Cc, 32), dtype=torch.float64, device=x.device)
            hip.call(fn_full, x, stats, scratch, B, HW, Cc, 32, 1e-6)
        return stats

    def invalidate(self):
        """drop the packed weight copies: the next call re-reads the module's parameters"""
        self._sig = None

    def _pack(self):
        """our copies of the weights this engine uses: 3x3 kernels re-laid [Cout][3][3][Cin] (Cin zero-padded to a multiple of 32),
        1x1 kernels as [Cout][Cin] matrices"""
        w = {}
        for k, v in self.vae.state_dict().items():
            if not k.startswith(self.PREFIXES) or not torch.is_floating_point(v):
                continue
            v = _chk(v.detach(), k)
            if v.dim() == 4 and v.shape[-1] == 3:
                v = v.permute(0, 2, 3, 1).contiguous()               # [Cout][Cin][3][3] -> [Cout][3][3][Cin]
                if v.shape[3] % 32:
                    pad = torch.zeros(v.shape[0], 3, 3, (v.shape[3] + 31) // 32 * 32, dtype=v.dtype, device=v.device)
                    pad[..., :v.shape[3]] = v
                    v = pad
                w[k] = v
            elif v.dim() == 4:
                w[k] = v.reshape(v.shape[0], v.shape[1])            # 1x1 conv == linear
            else:
                w[k] = v
        return w


class DecoderEngine(_VaeOps):
    """VQVAE.fhat_to_img on HIP kernels (reference vqvae.py:62-63, basic_vae.py:163-226)."""

    PREFIXES = ('decoder.', 'post_quant_conv.')
    # default precision of the VQVAE's own entry points (fhat_to_img, idxBl_to_img, ...): 'f32' unless the VQVAE's owner asks for 'f16'
    # here.  The sampling loop does NOT change it: SamplingEngine passes its own precision with every decode_nhwc call, so two VARs
    # sharing one VQVAE, or a VAR in the 16-bit mode next to direct VQVAE calls, never fight over a mode flag.
    precision = 'f32'
    unfused_tail = False         # tests: run norm_out / conv_out of the decoder as two launches

    def set_precision(self, precision: str):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.precision = precision

    def refresh(self):
        sig = self._signature()
        if sig == self._sig:
            return
        self._retire()
        w = self._pack()
        for k in [k for k in w if k.endswith('.upsample.conv.weight')]:        # Upsample2x convs: pre-summed 2x2 phase weights
            cout, _, _, cin = w[k].shape
            wp = torch.empty(4, cout, 2, 2, cin, dtype=torch.float32, device=w[k].device)
            hip.call('upconv_pack_f32', w[k], wp, cin, cout)
            w[k[:-len('weight')] + 'phase'] = wp
        self.w = w
        self.w16s = {}                       # {'f16' | 'bf16': 16-bit copies}: made by _ensure16() the first time such a decode runs on these weights
        self.nlev = 1 + max(int(k.split('.')[2]) for k in w if k.startswith('decoder.up.'))
        self._sig = sig
        self._built()

    def _ensure16(self, prec):
        """16-bit copies of every conv kernel (3x3, phase, 1x1 shortcut, attention projections) next to the fp32 ones; biases and GroupNorm affine
        stay fp32.  Selects them (self.w16) and the flavour's entry-point suffix / dtype for the decode that follows."""
        if prec not in self.w16s:
            self._wait_ready()
            self.w16s[prec] = {k: v.to(DT16[prec]).contiguous() for k, v in self.w.items()
                               if (k.endswith('.weight') or k.endswith('.phase')) and v.dim() >= 2 and '.norm' not in k}
            self._built()
        self.w16, self.sfx, self.dt16 = self.w16s[prec], prec, DT16[prec]

    # -- building blocks ---------------------------------------------------------------------------------------------
    def _part_buffer(self, B, nblk, Cout, dev):
        """scratch for the GroupNorm partials a conv leaves behind; remembered together with the tensor they describe"""
        return torch.empty((B, nblk, Cout, 2), dtype=torch.float64, device=dev)

    def conv3(self, x, key, B, Hh, Ww, up2=0, resid=None, out_mode=0, stats=False):
        """stats=True: the result feeds a GroupNorm next — the conv epilogue also emits per-block channel sums (when the shape
        allows), which gn() then uses instead of a statistics pass over the tensor."""
        wt = self.w[key + '.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        out = torch.empty((B, Cout, Hh, Ww) if out_mode else (B, Hh, Ww, Cout), dtype=torch.float32, device=x.device)
        nblk = hip.conv_gn_blocks(Hh, Ww, Cout) if (stats and out_mode == 0) else 0
        if nblk:
            part = self._part_buffer(B, nblk, Cout, x.device)
            hip.call('conv3x3_gn_nhwc_f32', x, wt, self.w[key + '.bias'], resid, out, part, B, Hh, Ww, Cin, Cout, up2)
            self._gn_part = (out, part, nblk)              # holds `out` so its address cannot be recycled before the next gn()
        else:
            hip.call('conv3x3_nhwc_f32', x, wt, self.w[key + '.bias'], resid, out, B, Hh, Ww, Cin, Cout, up2, out_mode)
        return out

    def gn_stats(self, x, B, HW):
        return self._stats_from(x, B, HW, 'gn_stats_f32')

    def gn(self, x, key, B, HW, silu):
        stats = self.gn_stats(x, B, HW)
        out = torch.empty_like(x)
        hip.call('gn_apply_f32', x, stats, self.w[key + '.weight'], self.w[key + '.bias'], out, B, HW, x.shape[-1], 32, int(silu))
        return out

    def tail(self, h, B, Hh, Ww, denorm):
        """norm_out -> swish -> conv_out -> clamp (-> (x + 1) / 2) (basic_vae.py:224-226, vqvae.py:63, var.py:190): one pass over the map
        (varhip_gn_silu_conv_out_f32) where it tiles into 8 x 32 patches, else GroupNorm apply + conv — the same bits either way"""
        wt = self.w['decoder.conv_out.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        if Hh % 8 == 0 and Ww % 32 == 0 and Cin % 32 == 0 and Cout <= 4 and (10 * 34 * 36 + 4 * Cin) * 4 <= 64 * 1024 and not self.unfused_tail:
            stats = self.gn_stats(h, B, Hh * Ww)
            out = torch.empty((B, Cout, Hh, Ww), dtype=torch.float32, device=h.device)
            hip.call('gn_silu_conv_out_f32', h, stats, self.w['decoder.norm_out.weight'], self.w['decoder.norm_out.bias'], wt,
                     self.w['decoder.conv_out.bias'], out, B, Hh, Ww, Cin, Cout, 32, 1 if denorm else 2)
            return out
        h = self.gn(h, 'decoder.norm_out', B, Hh * Ww, True)
        return self.conv3(h, 'decoder.conv_out', B, Hh, Ww, out_mode=1 if denorm else 2)

    def lin(self, x2d, key, resid=None):
        wt = self.w[key + '.weight']
        N, K = wt.shape
        M = x2d.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
        hip.call('gemm_nt_f32', x2d, K, wt, K, self.w[key + '.bias'], out, N, M, N, K, EPI_RESID if resid is not None else EPI_NONE,
                 resid, N, None, 0, 1, 0, 1, 0, 0, 0)
        return out

    def resblock(self, x, pre, B, Hh, Ww):
        HW = Hh * Ww
        h = self.conv3(self.gn(x, pre + '.norm1', B, HW, True), pre + '.conv1', B, Hh, Ww, stats=True)        # -> norm2
        hn = self.gn(h, pre + '.norm2', B, HW, True)
        sc = self.lin(x.view(B * HW, -1), pre + '.nin_shortcut').view(B, Hh, Ww, -1) if (pre + '.nin_shortcut.weight') in self.w else x
        return self.conv3(hn, pre + '.conv2', B, Hh, Ww, resid=sc, stats=True)                                  # -> the next block's norm

    def attnblock(self, x, pre, B, Hh, Ww):
        HW, Cc = Hh * Ww, x.shape[-1]
        dev = x.device
        xn = self.gn(x, pre + '.norm', B, HW, False).view(B * HW, Cc)
        wqkv, bqkv = self.w[pre + '.qkv.weight'], self.w[pre + '.qkv.bias']
        qk = torch.empty((B * HW, 2 * Cc), dtype=torch.float32, device=dev)
        hip.call('gemm_nt_f32', xn, Cc, wqkv, Cc, bqkv, qk, 2 * Cc, B * HW, 2 * Cc, Cc, EPI_NONE, None, 0, None, 0, 1, 0, 1, 0, 0, 0)
        vt = torch.empty((B, Cc, HW), dtype=torch.float32, device=dev)                   # V^T[b][c][j], bias per row (c)
        hip.call('gemm_nt_f32', wqkv[2 * Cc:], Cc, xn, Cc, bqkv[2 * Cc:], vt, HW, Cc, HW, Cc, EPI_NONE, None, 0, None, 0, 1, 1,
                 B, 0, HW * Cc, Cc * HW)
        s = torch.empty((B, HW, HW), dtype=torch.float32, device=dev)
        hip.call('gemm_nt_f32', qk, 2 * Cc, qk[:, Cc:], 2 * Cc, None, s, HW, HW, HW, Cc, EPI_NONE, None, 0, None, 0, 1, 0,
                 B, HW * 2 * Cc, HW * 2 * Cc, HW * HW)
        p = torch.empty_like(s)
        hip.call('softmax_rows_f32', s, p, B * HW, HW, float(np.float32(int(Cc) ** (-0.5))))
        o = torch.empty((B * HW, Cc), dtype=torch.float32, device=dev)
        hip.call('gemm_nt_f32', p, HW, vt, HW, None, o, Cc, HW, Cc, HW, EPI_NONE, None, 0, None, 0, 1, 0, B, HW * HW, Cc * HW, HW * Cc)
        return self.lin(o, pre + '.proj_out', resid=x.view(B * HW, Cc)).view(B, Hh, Ww, Cc)

    def flops_per_image_reference(self, P: int) -> float:
        """FLOPs (2/MAC) of one decode as the REFERENCE computes it (9-tap upsample convs; SURVEY.md §8d: 393.7 G at P=16, ch=160)"""
        self.refresh()
        w = self.w
        def c3(key, hw): co, _, _, ci = w[key + '.weight'].shape; return 2.0 * hw * co * 9 * ci
        def c1(key, hw): co, ci = w[key + '.weight'].shape; return 2.0 * hw * co * ci
        def res(pre, hw): return c3(pre + '.conv1', hw) + c3(pre + '.conv2', hw) + (c1(pre + '.nin_shortcut', hw) if (pre + '.nin_shortcut.weight') in w else 0.0)
        def att(pre, hw): c = w[pre + '.proj_out.weight'].shape[0]; return c1(pre + '.qkv', hw) + c1(pre + '.proj_out', hw) + 2.0 * 2 * hw * hw * c
        hw = P * P
        f = c3('post_quant_conv', hw) + c3('decoder.conv_in', hw) + res('decoder.mid.block_1', hw) + att('decoder.mid.attn_1', hw) + res('decoder.mid.block_2', hw)
        for lev in reversed(range(self.nlev)):
            for ib in range(3):
                f += res(f'decoder.up.{lev}.block.{ib}', hw)
                if f'decoder.up.{lev}.attn.{ib}.norm.weight' in w: f += att(f'decoder.up.{lev}.attn.{ib}', hw)
            if lev != 0:
                hw *= 4; f += c3(f'decoder.up.{lev}.upsample.conv', hw)
        return f + c3('decoder.conv_out', hw)

    def flops_per_image_executed(self, P: int) -> float:
        """FLOPs the kernels execute for one decode: as flops_per_image_reference, with the Upsample2x convolutions in their folded
        four-phase form (4 taps per output pixel instead of 9)"""
        f = self.flops_per_image_reference(P)
        hw = P * P
        for lev in reversed(range(self.nlev)):
            if lev != 0:
                hw *= 4
                co, _, _, ci = self.w[f'decoder.up.{lev}.upsample.conv.weight'].shape
                f -= 2.0 * hw * co * 5 * ci
        return f

    # -- the 16-bit throughput mode: fp16 activations, conv16.hip / rowops16.hip ----------------------------------------
    def conv3_16(self, x, key, B, Hh, Ww, resid=None, out_mode=0, stats=False):
        wt = self.w16[key + '.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        out = torch.empty((B, Cout, Hh, Ww), dtype=torch.float32, device=x.device) if out_mode else torch.empty((B, Hh, Ww, Cout), dtype=self.dt16, device=x.device)
        nblk = hip.conv_gn_blocks(Hh, Ww, Cout) if (stats and out_mode == 0 and Cout % 4 == 0) else 0
        part = self._part_buffer(B, nblk, Cout, x.device) if nblk else None
        hip.call('conv3x3_nhwc_' + self.sfx, x, wt, self.w[key + '.bias'], resid, out, part, B, Hh, Ww, Cin, Cout, out_mode)
        if nblk: self._gn_part = (out, part, nblk)
        return out

    def gn_stats16(self, x, B, HW):
        return self._stats_from(x, B, HW, 'gn_stats_' + self.sfx)

    def gn16(self, x, key, B, HW, silu):
        stats = self.gn_stats16(x, B, HW)
        out = torch.empty_like(x)
        hip.call('gn_apply_' + self.sfx, x, stats, self.w[key + '.weight'], self.w[key + '.bias'], out, B, HW, x.shape[-1], 32, int(silu))
        return out

    def tail16(self, h, B, Hh, Ww, denorm):
        """norm_out -> swish -> conv_out -> clamp (-> (x + 1) / 2) (basic_vae.py:224-226, vqvae.py:63, var.py:190): one pass over the map
        (varhip_gn_silu_conv_out_*) where it tiles into 8 x 32 patches, else GroupNorm apply + conv — the same bits either way"""
        wt = self.w16['decoder.conv_out.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        if Hh % 8 == 0 and Ww % 32 == 0 and Cin % 32 == 0 and Cout <= 16 and (2 * 22 * 1024 + Cout * 9 * Cin * 2 + 16 + 8 * Cin) <= 64 * 1024 and not self.unfused_tail:
            stats = self.gn_stats16(h, B, Hh * Ww)
            out = torch.empty((B, Cout, Hh, Ww), dtype=torch.float32, device=h.device)
            hip.call('gn_silu_conv_out_' + self.sfx, h, stats, self.w['decoder.norm_out.weight'], self.w['decoder.norm_out.bias'], wt,
                     self.w['decoder.conv_out.bias'], out, B, Hh, Ww, Cin, Cout, 32, 1 if denorm else 2)
            return out
        h = self.gn16(h, 'decoder.norm_out', B, Hh * Ww, True)
        return self.conv3_16(h, 'decoder.conv_out', B, Hh, Ww, out_mode=1 if denorm else 2)

    fuse_gn = os.environ.get('VARHIP_FUSE_GN', '1') != '0'      # False (tests, A/B runs): every GroupNorm + SiLU as its own pass in front of the conv (the same bits)

    def gnconv3_16(self, x, nkey, ckey, B, Hh, Ww, resid=None):
        """conv(swish(norm(x))) (basic_vae.py:57-60): one launch where the halo-patch conv can normalise its own input patch, else apply pass + conv"""
        wt = self.w16[ckey + '.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        if not (self.fuse_gn and hip.conv16_gn_fusable(B, Hh, Ww, Cin, Cout)):
            return self.conv3_16(self.gn16(x, nkey, B, Hh * Ww, True), ckey, B, Hh, Ww, resid=resid, stats=True)
        stats = self.gn_stats16(x, B, Hh * Ww)
        out = torch.empty((B, Hh, Ww, Cout), dtype=self.dt16, device=x.device)
        nblk = hip.conv_gn_blocks(Hh, Ww, Cout) if Cout % 4 == 0 else 0
        part = self._part_buffer(B, nblk, Cout, x.device) if nblk else None
        table = torch.empty((B, 2, Cin), dtype=torch.float32, device=x.device)
        hip.call('gn_scale_shift_f32', stats, self.w[nkey + '.weight'], self.w[nkey + '.bias'], table, B, Cin, 32)
        hip.call('gnconv3x3_nhwc_' + self.sfx, x, table, 1, wt, self.w[ckey + '.bias'], resid, out, part, B, Hh, Ww, Cin, Cout)
        if nblk: self._gn_part = (out, part, nblk)
        return out

    def resblock16(self, x, pre, B, Hh, Ww):
        HW = Hh * Ww
        h = self.gnconv3_16(x, pre + '.norm1', pre + '.conv1', B, Hh, Ww)
        sc = x
        if (pre + '.nin_shortcut.weight') in self.w16:              # 1x1 conv == fp16 GEMM over the pixels
            wt = self.w16[pre + '.nin_shortcut.weight']
            N, K = wt.shape
            sc = torch.empty((B, Hh, Ww, N), dtype=self.dt16, device=x.device)
            hip.call('gemm_nt_' + self.sfx, x, K, wt, K, self.w[pre + '.nin_shortcut.bias'], sc, N, 1, B * HW, N, K, EPI_NONE, None, 0, 0, None, 0, 1, 1, 0, 0, 0)
        return self.gnconv3_16(h, pre + '.norm2', pre + '.conv2', B, Hh, Ww, resid=sc)

    def attnblock16(self, x, pre, B, Hh, Ww):
        """AttnBlock (basic_vae.py:73-92) on fp16 activations: the five products (q/k projection, V^T projection, q.k^T, p.v, proj_out + residual)
        on the f16 MFMA GEMM with fp32 accumulation; scores and softmax in fp32, the probabilities rounded to fp16 for p.v"""
        HW, Cc = Hh * Ww, x.shape[-1]
        dev = x.device
        f16 = self.dt16
        if Cc % 64 or HW % 64:                     # (tiny test configurations: the f16 GEMM contracts 64 at a time) fp32 attention between two casts
            x32 = torch.empty(x.shape, dtype=torch.float32, device=dev)
            hip.call(f'cast_{self.sfx}_to_f32', x, x32, x.numel())
            self._gn_part = None
            y32 = self.attnblock(x32, pre, B, Hh, Ww)
            y = torch.empty(x.shape, dtype=f16, device=dev)
            hip.call('cast_f32_to_' + self.sfx, y32, y, y.numel())
            return y
        xn = self.gn16(x, pre + '.norm', B, HW, False).view(B * HW, Cc)
        wqkv, bqkv = self.w16[pre + '.qkv.weight'], self.w[pre + '.qkv.bias']
        g16 = lambda A, lda, W, ldw, bias, out, ldo, o16, M, N, K, epi=EPI_NONE, resid=None, ldr=0, r16=0, batch=1, sA=0, sW=0, sO=0: \
            hip.call('gemm_nt_' + self.sfx, A, lda, W, ldw, bias, out, ldo, o16, M, N, K, epi, resid, ldr, r16, None, 0, 1, batch, sA, sW, sO)
        qk = torch.empty((B * HW, 2 * Cc), dtype=f16, device=dev)
        g16(xn, Cc, wqkv, Cc, bqkv, qk, 2 * Cc, 1, B * HW, 2 * Cc, Cc)
        # V^T[b][c][j] WITHOUT its bias: the GEMM's bias is per column and here c is the row.  The rows of p sum to one, so the bias is added
        # to p.v instead (column c of that product) — equal up to the fp16 rounding of p (|sum p - 1| <= 1e-3, bias ~1e-2: 1e-5)
        vt = torch.empty((B, Cc, HW), dtype=f16, device=dev)
        g16(wqkv[2 * Cc:], Cc, xn, Cc, None, vt, HW, 1, Cc, HW, Cc, batch=B, sA=0, sW=HW * Cc, sO=Cc * HW)
        s = torch.empty((B, HW, HW), dtype=torch.float32, device=dev)
        g16(qk, 2 * Cc, qk[:, Cc:], 2 * Cc, None, s, HW, 0, HW, HW, Cc, batch=B, sA=HW * 2 * Cc, sW=HW * 2 * Cc, sO=HW * HW)
        p32 = torch.empty_like(s)
        hip.call('softmax_rows_f32', s, p32, B * HW, HW, float(np.float32(int(Cc) ** (-0.5))))
        p = torch.empty((B, HW, HW), dtype=f16, device=dev)
        hip.call('cast_f32_to_' + self.sfx, p32, p, p.numel())
        o = torch.empty((B * HW, Cc), dtype=f16, device=dev)
        g16(p, HW, vt, HW, bqkv[2 * Cc:], o, Cc, 1, HW, Cc, HW, batch=B, sA=HW * HW, sW=Cc * HW, sO=HW * Cc)
        y = torch.empty((B, Hh, Ww, Cc), dtype=f16, device=dev)
        g16(o, Cc, self.w16[pre + '.proj_out.weight'], Cc, self.w[pre + '.proj_out.bias'], y, Cc, 1, B * HW, Cc, Cc, epi=EPI_RESID, resid=x, ldr=Cc, r16=1)
        return y

    def _decode16(self, f_hat: torch.Tensor, denorm: bool) -> torch.Tensor:
        B, P = f_hat.shape[0], f_hat.shape[1]
        Hh = Ww = P
        x = torch.empty(f_hat.shape, dtype=self.dt16, device=f_hat.device)
        hip.call('cast_f32_to_' + self.sfx, f_hat.contiguous(), x, x.numel())
        h = self.conv3_16(x, 'post_quant_conv', B, Hh, Ww)
        h = self.conv3_16(h, 'decoder.conv_in', B, Hh, Ww, stats=True)
        h = self.resblock16(h, 'decoder.mid.block_1', B, Hh, Ww)
        h = self.attnblock16(h, 'decoder.mid.attn_1', B, Hh, Ww)
        h = self.resblock16(h, 'decoder.mid.block_2', B, Hh, Ww)
        for lev in reversed(range(self.nlev)):
            for ib in range(3):
                h = self.resblock16(h, f'decoder.up.{lev}.block.{ib}', B, Hh, Ww)
                if f'decoder.up.{lev}.attn.{ib}.norm.weight' in self.w:
                    h = self.attnblock16(h, f'decoder.up.{lev}.attn.{ib}', B, Hh, Ww)
            if lev != 0:
                Hh, Ww = 2 * Hh, 2 * Ww
                key = f'decoder.up.{lev}.upsample.conv'
                wp = self.w16[key + '.phase']
                up = torch.empty((B, Hh, Ww, wp.shape[1]), dtype=self.dt16, device=h.device)
                nblk = hip.conv_gn_blocks(Hh, Ww, wp.shape[1], phase=True)
                part = self._part_buffer(B, nblk, wp.shape[1], h.device) if nblk else None
                hip.call('upconv_phase_' + self.sfx, h, wp, self.w[key + '.bias'], up, part, B, Hh, Ww, wp.shape[4], wp.shape[1])
                self._gn_part = (up, part, nblk) if nblk else None
                h = up
        return self.tail16(h, B, Hh, Ww, denorm)

    def decode_nhwc(self, f_hat: torch.Tensor, denorm: bool = True, precision: Optional[str] = None) -> torch.Tensor:
        """[B,P,P,Cvae] channels-last -> [B,3,16P,16P]; denorm=True: in [0,1] (clamp and (x+1)/2 fused into the last conv, what
        autoregressive_infer_cfg returns); denorm=False: clamped to [-1,1] (VQVAE.fhat_to_img's contract).
        precision: 'f32' / 'f16' / 'bf16' for THIS call (None: the engine's default, `self.precision`)"""
        self.refresh()
        prec = precision or self.precision
        if prec != 'f32':
            self._ensure16(prec)
            self._wait_ready()
            return self._decode16(f_hat, denorm)
        self._wait_ready()
        B, P = f_hat.shape[0], f_hat.shape[1]
        Hh = Ww = P
        h = self.conv3(f_hat, 'post_quant_conv', B, Hh, Ww)
        h = self.conv3(h, 'decoder.conv_in', B, Hh, Ww, stats=True)
        h = self.resblock(h, 'decoder.mid.block_1', B, Hh, Ww)
        h = self.attnblock(h, 'decoder.mid.attn_1', B, Hh, Ww)
        h = self.resblock(h, 'decoder.mid.block_2', B, Hh, Ww)
        for lev in reversed(range(self.nlev)):
            for ib in range(3):
                h = self.resblock(h, f'decoder.up.{lev}.block.{ib}', B, Hh, Ww)
                if f'decoder.up.{lev}.attn.{ib}.norm.weight' in self.w:
                    h = self.attnblock(h, f'decoder.up.{lev}.attn.{ib}', B, Hh, Ww)
            if lev != 0:                                             # Upsample2x: nearest 2x + conv3x3, as 4 phase convs on the low-res map
                Hh, Ww = 2 * Hh, 2 * Ww
                key = f'decoder.up.{lev}.upsample.conv'
                wp = self.w[key + '.phase']
                up = torch.empty((B, Hh, Ww, wp.shape[1]), dtype=torch.float32, device=h.device)
                nblk = hip.conv_gn_blocks(Hh, Ww, wp.shape[1], phase=True)
                if nblk:                                             # its result is normalised by the next level's first ResnetBlock
                    part = self._part_buffer(B, nblk, wp.shape[1], h.device)
                    hip.call('upconv_phase_gn_f32', h, wp, self.w[key + '.bias'], up, part, B, Hh, Ww, wp.shape[4], wp.shape[1])
                    self._gn_part = (up, part, nblk)
                else:
                    hip.call('upconv_phase_f32', h, wp, self.w[key + '.bias'], up, B, Hh, Ww, wp.shape[4], wp.shape[1])
                h = up
        return self.tail(h, B, Hh, Ww, denorm)


class QuantizerEngine:
    """Encode-side methods of VectorQuantizer2 on HIP: f_to_idxBl_or_fhat (quant.py:135-166) and idxBl_to_var_input (quant.py:169-184)"""

    def __init__(self, quant):
        self.quant = quant
        self._sig = None

    def refresh(self):
        q = self.quant
        sig = _signature(q.parameters())
        if sig == self._sig:
            return
        self.codebook = _chk(q.embedding.weight.detach(), 'codebook')
        self.phi = [(_chk(p.weight.detach(), 'phi').permute(0, 2, 3, 1).contiguous(), _chk(p.bias.detach(), 'phi'), float(p.resi_ratio))
                    for p in q.quant_resi.phis()]
        self._taps = {}
        self._sig = sig

    def invalidate(self):
        self._sig = None

    def taps(self, pn, P, dev):
        if (pn, P) not in self._taps:
            ti, tw = bicubic_taps(pn, P)
            self._taps[(pn, P)] = (torch.from_numpy(ti).to(dev), torch.from_numpy(tw).to(dev))
        return self._taps[(pn, P)]

    @torch.no_grad()
    def quantize(self, f_nhwc: torch.Tensor, to_fhat: bool, patch_nums):
        """residual quantisation scale by scale (quant.py:147-164): idx lists (B, pn^2) int64, or the cumulative f_hat's (NCHW)"""
        self.refresh()
        B, P, _, Cv = f_nhwc.shape
        dev = f_nhwc.device
        f_rest = f_nhwc.clone()
        f_hat = torch.zeros_like(f_rest)
        up = torch.empty_like(f_rest)
        S, out = len(patch_nums), []
        for si, pn in enumerate(patch_nums):
            if si != S - 1:
                z = torch.empty((B, pn * pn, Cv), dtype=torch.float32, device=dev)
                hip.call('area_pool_f32', f_rest, z, B, P, pn, Cv)
            else:
                z = f_rest
            idx = torch.empty(B * pn * pn, dtype=torch.int64, device=dev)
            # argmin |z - e|^2, or argmax cos(z, e) when using_znorm (quant.py:151-157)
            hip.call('nearest_code_cos_f32' if self.quant.using_znorm else 'nearest_code_f32', z, self.codebook, idx, B * pn * pn, self.codebook.shape[0], Cv)
            ti, tw = self.taps(pn, P, dev) if pn != P else (None, None)
            pw, pb, ratio = self.phi[phi_index(si, S, len(self.phi))]
            hip.call('quant_residual_f32', idx, self.codebook, ti, tw, pw, pb, ratio, up, f_hat, f_rest, B, pn, P, Cv)
            out.append(f_hat.permute(0, 3, 1, 2).contiguous() if to_fhat else idx.view(B, pn * pn))
        return out

    @torch.no_grad()
    def fhat_from_scales(self, items, patch_nums, from_tokens: bool, last_one: bool):
        """f_hat (B, Cvae, P, P) accumulated over all scales (quant.py:107-133 embed_to_fhat with all_to_max_scale=True; with
        from_tokens=True the codebook lookup of vqvae.py:77-84 idxBl_to_img is fused in).  items[si]: (B, pn*pn) token ids, or the
        embedding maps (B, Cvae, pn, pn).  Same kernels, same order of operations as the sampling loop's incremental f_hat."""
        self.refresh()
        B, P, Cv, S = items[0].shape[0], patch_nums[-1], self.codebook.shape[1], len(patch_nums)
        dev = self.codebook.device
        f_hat = torch.zeros((B, P, P, Cv), dtype=torch.float32, device=dev)
        up = torch.empty_like(f_hat)
        outs = []
        for si, pn in enumerate(patch_nums):
            ti, tw = self.taps(pn, P, dev) if pn != P else (None, None)
            pw, pb, ratio = self.phi[phi_index(si, S, len(self.phi))]
            if from_tokens:
                hip.call('quant_accum_f32', items[si].to(dev, torch.int64).contiguous(), self.codebook, ti, tw, pw, pb, ratio, up, f_hat, B, pn, P, Cv)
            else:
                h = torch.empty((B, pn * pn, Cv), dtype=torch.float32, device=dev)
                hip.call('nchw_to_nhwc_f32', items[si].to(dev, torch.float32).contiguous(), h, B, Cv, pn * pn)
                hip.call('quant_accum_h_f32', h, ti, tw, pw, pb, ratio, up, f_hat, B, pn, P, Cv)
            if not last_one or si == S - 1:
                o = torch.empty((B, Cv, P, P), dtype=torch.float32, device=dev)
                hip.call('nhwc_to_nchw_f32', f_hat, o, B, Cv, P * P)
                outs.append(o)
        return outs[-1] if last_one else outs

    @torch.no_grad()
    def var_input(self, idx_list, patch_nums) -> torch.Tensor:
        """teacher-forcing input (B, L - first_l, Cvae) of VAR.forward from ground-truth token maps (quant.py:169-184)"""
        self.refresh()
        B, P, Cv, S = idx_list[0].shape[0], patch_nums[-1], self.codebook.shape[1], len(patch_nums)
        dev = idx_list[0].device
        f_hat = torch.zeros((B, P, P, Cv), dtype=torch.float32, device=dev)
        up = torch.empty_like(f_hat)
        L = sum(p * p for p in patch_nums)
        out = torch.empty((B, L - patch_nums[0] ** 2, Cv), dtype=torch.float32, device=dev)
        cur = 0
        for si in range(S - 1):
            pn, pq = patch_nums[si], patch_nums[si + 1]
            ti, tw = self.taps(pn, P, dev) if pn != P else (None, None)
            pw, pb, ratio = self.phi[phi_index(si, S, len(self.phi))]
            hip.call('quant_accum_f32', idx_list[si].to(torch.int64).contiguous(), self.codebook, ti, tw, pw, pb, ratio, up, f_hat, B, pn, P, Cv)
            pooled = torch.empty((B, pq * pq, Cv), dtype=torch.float32, device=dev)
            hip.call('area_pool_f32', f_hat, pooled, B, P, pq, Cv)
            out[:, cur:cur + pq * pq] = pooled
            cur += pq * pq
        return out


class EncoderEngine(DecoderEngine):
    """Encoder + quant_conv on HIP (reference vqvae.py:65-75 img_to_*: basic_vae.py:99-160).  Inherits the channels-last conv /
    GroupNorm / attention building blocks of DecoderEngine."""

    PREFIXES = ('encoder.', 'quant_conv.')

    def refresh(self):
        sig = self._signature()
        if sig == self._sig:
            return
        self._retire()
        self.w = self._pack()
        self.nlev = 1 + max(int(k.split('.')[2]) for k in self.w if k.startswith('encoder.down.'))
        self._sig = sig
        self._built()

    def conv_s2(self, x, key, B, Hh, Ww):
        wt = self.w[key + '.weight']
        Cout, Cin = wt.shape[0], wt.shape[3]
        out = torch.empty((B, Hh, Ww, Cout), dtype=torch.float32, device=x.device)
        hip.call('conv3x3_s2_nhwc_f32', x, wt, self.w[key + '.bias'], out, B, Hh, Ww, Cin, Cout)
        return out

    @torch.no_grad()
    def encode(self, img: torch.Tensor) -> torch.Tensor:
        """img (B,3,H,W) fp32 in [-1,1] -> f (B, H/16, W/16, Cvae) channels-last == quant_conv(encoder(img))"""
        self.refresh()
        self._wait_ready()
        B, Ci, Hh, Ww = img.shape
        cin_pad = self.w['encoder.conv_in.weight'].shape[3]
        x = torch.empty((B, Hh, Ww, cin_pad), dtype=torch.float32, device=img.device)
        hip.call('nchw_to_nhwc_pad_f32', img.contiguous(), x, B, Ci, Hh * Ww, cin_pad)
        h = self.conv3(x, 'encoder.conv_in', B, Hh, Ww)
        for lev in range(self.nlev):
            for ib in range(2):
                h = self.resblock(h, f'encoder.down.{lev}.block.{ib}', B, Hh, Ww)
                if f'encoder.down.{lev}.attn.{ib}.norm.weight' in self.w:
                    h = self.attnblock(h, f'encoder.down.{lev}.attn.{ib}', B, Hh, Ww)
            if lev != self.nlev - 1:
                Hh, Ww = Hh // 2, Ww // 2
                h = self.conv_s2(h, f'encoder.down.{lev}.downsample.conv', B, Hh, Ww)
        h = self.resblock(h, 'encoder.mid.block_1', B, Hh, Ww)
        h = self.attnblock(h, 'encoder.mid.attn_1', B, Hh, Ww)
        h = self.resblock(h, 'encoder.mid.block_2', B, Hh, Ww)
        h = self.conv3(self.gn(h, 'encoder.norm_out', B, Hh * Ww, True), 'encoder.conv_out', B, Hh, Ww)
        return self.conv3(h, 'quant_conv', B, Hh, Ww)



AUTOCAST_PRECISION = {torch.float16: 'f16', torch.bfloat16: 'bf16'}


class SamplingEngine:
    """The AR loop of VAR.autoregressive_infer_cfg on HIP kernels.  One engine per VAR module; calls on different HIP streams may be in flight
    together (per-stream workspaces), the host side is not thread-safe."""

    MAX_WORKSPACES = 6          # buffer sets kept alive per engine (workspace(): one per (batch size, stream, precision) in use)

    def __init__(self, var):
        self.var = var
        self._sig = None
        self.w: dict = {}
        self._ws: Dict[tuple, dict] = {}            # (batch size, HIP stream, precision) -> buffers of a call
        self._ws_tf: Dict[tuple, dict] = {}         # the same for teacher_forced_logits
        self._ready = None                          # event behind the kernels that built the derived weight copies (see _VaeOps._built)
        self._labels_ok = None                      # identity of the last label tensor whose range was checked (one host sync saved per repeated call)

        self.policy = 'auto' if os.environ.get('VARHIP_FOLLOW_AUTOCAST', '0') not in ('', '0') else 'f32'      # what set_precision() was given: 'f32' | 'f16' | 'bf16' | 'auto' (follow the caller's torch.autocast)
        self.precision = 'f32'              # the arithmetic of the call in progress (the policy, resolved): 'f16' / 'bf16' = the 16-bit throughput mode (include/var_hip.h)
        self.dec = var.vae_proxy[0]._decoder_engine()       # the VQVAE's own engine: one packed copy of the decoder weights, one place to invalidate
        self.last_trace: Optional[dict] = None

    # -- weights -----------------------------------------------------------------------------------------------------
    def refresh(self):
        var = self.var
        quant = var.vae_quant_proxy[0]
        sig = _signature(list(var.parameters()) + list(quant.parameters())) + (getattr(var.vae_proxy[0], '_hip_generation', 0),)
        if sig == self._sig:
            self._ensure16()
            return
        if var.C != 64 * var.num_heads:
            raise hip.VarHipError(f'the HIP attention kernels are built for head_dim 64, got embed_dim {var.C} / {var.num_heads} heads')
        if self.w and torch.cuda.is_available():
            torch.cuda.synchronize()         # weights changed: other streams may still read the old derived copies that are dropped below (rare: checkpoint load, EMA swap)
        dev = var.pos_start.device
        w = {}
        g = lambda t, n: _chk(t.detach(), n)
        C = var.C
        w['class_emb'] = g(var.class_emb.weight, 'class_emb')
        w['pos_start'] = g(var.pos_start, 'pos_start').view(var.first_l, C)
        w['pos_1LC'] = g(var.pos_1LC, 'pos_1LC').view(var.L, C)
        w['lvl_embed'] = g(var.lvl_embed.weight, 'lvl_embed')
        w['lvl_1L'] = var.lvl_1L.view(-1).to(torch.int64).contiguous()
        w['word_w'], w['word_b'] = g(var.word_embed.weight, 'word_embed.weight'), g(var.word_embed.bias, 'word_embed.bias')
        w['head_w'], w['head_b'] = g(var.head.weight, 'head.weight'), g(var.head.bias, 'head.bias')
        w['hn_w'], w['hn_b'] = g(var.head_nm.ada_lin[1].weight, 'head_nm'), g(var.head_nm.ada_lin[1].bias, 'head_nm')
        if var.shared_aln:
            w['sal_w'], w['sal_b'] = g(var.shared_ada_lin[1].weight, 'shared_ada_lin'), g(var.shared_ada_lin[1].bias, 'shared_ada_lin')
        blocks = []
        for b in var.blocks:
            a = b.attn
            d = dict(
                qkv_w=g(a.mat_qkv.weight, 'mat_qkv'), qkv_b=torch.cat((a.q_bias.detach(), a.zero_k_bias, a.v_bias.detach())).float().contiguous(),
                proj_w=g(a.proj.weight, 'proj'), proj_b=g(a.proj.bias, 'proj'),
                fc1_w=g(b.ffn.fc1.weight, 'fc1'), fc1_b=g(b.ffn.fc1.bias, 'fc1'), fc2_w=g(b.ffn.fc2.weight, 'fc2'), fc2_b=g(b.ffn.fc2.bias, 'fc2'),
                smul=g(a.scale_mul_1H11, 'scale_mul').view(-1) if a.attn_l2_norm else None, l2=bool(a.attn_l2_norm), plain_scale=float(a.scale))
            if var.shared_aln:
                d['gss'] = g(b.ada_gss, 'ada_gss').view(-1)
            blocks.append(d)
        w['blocks'] = blocks
        w['b16'], w['head_w16'] = {}, {}     # {'f16' | 'bf16': ...} 16-bit copies of the GEMM weights, made by _ensure16() when such a call first needs them
        if not var.shared_aln and var.depth * 6 * C * C * 4 < 3.5e9:          # (beyond: the weight rows would leave the GEMM's 32-bit request offsets)
            # all blocks' ada_lin projections as ONE GEMM per call (N = depth * 6C): sixteen launches of 192 workgroups each were bound by one
            # workgroup's K loop (29 us each at d16).  The packed copy (0.4 GB fp32 at d16, 2.65 GB at d30; rebuilt when weights change) is the only
            # copy the engine holds: the per-block path is not used beside it
            w['ada_w_all'] = torch.cat([g(b.ada_lin[1].weight, 'ada_lin') for b in var.blocks], dim=0)
            w['ada_b_all'] = torch.cat([g(b.ada_lin[1].bias, 'ada_lin') for b in var.blocks], dim=0)
        elif not var.shared_aln:
            for d, b in zip(blocks, var.blocks):
                d['ada_w'], d['ada_b'] = g(b.ada_lin[1].weight, 'ada_lin'), g(b.ada_lin[1].bias, 'ada_lin')
        w['codebook'] = g(quant.embedding.weight, 'codebook')
        w['codebook_T'] = w['codebook'].t().contiguous()          # [Cvae][V]: "probabilities @ codebook" as an NT GEMM (more_smooth)
        phis = list(quant.quant_resi.phis())
        w['phi'] = [(g(p.weight, 'phi').permute(0, 2, 3, 1).contiguous(), g(p.bias, 'phi'), float(p.resi_ratio)) for p in phis]
        w['taps'] = {}
        P = var.patch_nums[-1]
        for pn in var.patch_nums:
            if pn != P:
                ti, tw = bicubic_taps(pn, P)
                w['taps'][pn] = (torch.from_numpy(ti).to(dev), torch.from_numpy(tw).to(dev))
        self.w = w
        self._sig = sig
        self._labels_ok = None
        self._built()
        self._ensure16()

    _built = _VaeOps._built
    _wait_ready = _VaeOps._wait_ready

    def _ensure16(self):
        """16-bit copies of the four GEMM weights of every block and of the head (round-to-nearest-even, once per weight change and flavour), kept
        per flavour so that calls alternating between precisions ('auto' inside and outside an autocast region) convert nothing twice"""
        prec = self.precision
        if prec == 'f32' or prec in self.w['b16']:
            return
        self._wait_ready()
        dt = DT16[prec]
        self.w['b16'][prec] = [{k + '16': d[k].to(dt).contiguous() for k in ('qkv_w', 'proj_w', 'fc1_w', 'fc2_w')} for d in self.w['blocks']]
        self.w['head_w16'][prec] = self.w['head_w'].to(dt).contiguous()
        self._built()

    def resolve_precision(self) -> str:
        """the arithmetic of the call that starts now: the policy itself, or under 'auto' what the caller's torch.autocast('cuda', dtype=...) asks for
        (reference basic_var.py:97 branches on the dtype autocast hands it; demo_sample.py:66-68 is the caller)"""
        if self.policy == 'auto':
            self.precision = AUTOCAST_PRECISION.get(torch.get_autocast_dtype('cuda'), 'f32') if torch.is_autocast_enabled('cuda') else 'f32'
        else:
            self.precision = self.policy
        return self.precision

    def set_precision(self, precision: str):
        """'f32' (default; the parity contract: token ids bit-identical to the CPU oracle), 'f16' or 'bf16': 16-bit weights / GEMM operands / KV
        cache with fp32 accumulation on the f16 MFMAs — what the reference's harness asks for with torch.autocast(fp16)
        (demo_sample.py:66-68), and the decoder call that ends the loop runs on fp16 activations / conv weights too (conv16.hip).  LayerNorm /
        GroupNorm statistics, AdaLN parameters, the residual stream, softmax, logits, sampler and quantizer stay fp32.  The VQVAE's own entry
        points (fhat_to_img, idxBl_to_img, ...) are NOT switched: the shared DecoderEngine is told the precision per call."""
        if precision not in PRECISIONS + ('auto',):
            raise ValueError(f"precision must be one of {PRECISIONS + ('auto',)}")
        self.policy = precision
        if precision != 'auto' and precision != self.precision:
            self.precision = precision
            self._ws = {k: v for k, v in self._ws.items() if k[2] == precision}          # an explicit switch releases the other modes' buffers (6-11 GB each at d16 / B=64)
            self._ws_tf = {k: v for k, v in self._ws_tf.items() if k[2] == precision}

    def invalidate(self):
        """forget the packed weight copies of the sampling loop and of the decoder (call after editing parameters through `.data`)"""
        self._sig = None
        self.dec.invalidate()

    # -- workspaces --------------------------------------------------------------------------------------------------
    def workspace(self, B: int):
        """buffers of one call (residual stream, GEMM operands, KV caches ...), cached per (batch size, HIP stream): calls issued on
        DIFFERENT streams may be in flight together — each owns its buffers, everything else a call allocates comes from torch's
        stream-ordered allocator — and a call's latency-bound small scales then run beside another call's decoder"""
        sid = int(torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get((B, sid, self.precision))
        dev = self.var.pos_start.device
        if ws is not None and ws['dev'] == dev:
            return ws
        var = self.var
        C, L, H, V, Cv, P = var.C, var.L, var.num_heads, var.V, var.Cvae, var.patch_nums[-1]
        lmax = max(p * p for p in var.patch_nums)
        M = 2 * B * lmax
        hid = var.blocks[0].ffn.fc1.weight.shape[0]
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        act = DT16.get(self.precision, torch.float32)                           # GEMM operands and KV cache; x / x2 / logits stay fp32
        ws = dict(dev=dev, x=e(M, C), x2=e(M, C), xn=e(M, C, dt=act), q=e(M, C, dt=act), att=e(M, C, dt=act), hid=e(M, hid, dt=act), logits=e(M, V),
                  idx=e(B * lmax, dt=torch.int64), lvl_pos=e(L, C), cond=e(2 * B, C), cond_silu=e(2 * B, C), hn=e(2 * B, 2 * C),
                  ada=e(var.depth, 2 * B, 6 * C) if var.shared_aln else e(2 * B, var.depth * 6 * C), shared=e(2 * B, 6 * C) if var.shared_aln else None,
                  kc=[torch.zeros(2 * B, H, L, 64, dtype=act, device=dev) for _ in range(var.depth)],
                  vc=[torch.zeros(2 * B, H, L, 64, dtype=act, device=dev) for _ in range(var.depth)],
                  f_hat=e(B, P, P, Cv), up=e(B, P, P, Cv), pooled=e(B * lmax, Cv))
        self._ws = self._evict(self._ws, sid)
        self._ws[(B, sid, self.precision)] = ws
        return ws

    def _evict(self, cache: dict, sid: int) -> dict:
        """make room for a new buffer set: one batch size resident at a time per (stream, precision), and a bounded number of sets (oldest first: a
        d16 / B=64 set is 6-11 GB; its memory returns to the stream it was allocated on, so work still queued there is safe)"""
        cache = {k: v for k, v in cache.items() if (k[1], k[2]) != (sid, self.precision)}
        while len(cache) >= self.MAX_WORKSPACES:
            cache.pop(next(iter(cache)))
        return cache

    def _check_labels(self, label_B: torch.Tensor):
        """labels index class_emb: out-of-range ones must not reach the kernel.  One device-side reduction and ONE host sync (a sync drains the
        stream, so back-to-back calls would otherwise never overlap their host side with the previous call's kernels); a tensor that was checked
        before and has not been written since (same storage, same version counter) is not checked again."""
        ident = (label_B.data_ptr(), _ver(label_B), label_B.numel(), label_B.device)
        if ident == self._labels_ok and ident[1] >= 0:
            return
        lo, hi = torch.aminmax(label_B)
        lo, hi = torch.stack((lo, hi)).tolist()
        if lo < 0 or hi > self.var.num_classes:
            raise ValueError(f'labels must lie in [0, {self.var.num_classes}]')
        self._labels_ok = ident

    def gemm(self, A, W, bias, out, M, epi=EPI_NONE, resid=None, gamma=None, ldg=0, rpg=1):
        N, K = W.shape
        hip.call('gemm_nt_f32', A, K, W, K, bias, out, N, M, N, K, epi, resid, N, gamma, ldg, rpg, 0, 1, 0, 0, 0)

    def neighbor_table(self, n: int):
        """(idx int32 [V, n], dist float32 [V, n]) nearest codes of every code (var.py:459-462); cached until the codebook changes."""
        self.refresh()
        tabs = self.w.setdefault('nbr', {})
        if n not in tabs:
            cb = self.w['codebook']
            V, D = cb.shape
            idx = torch.empty(V, n, dtype=torch.int32, device=cb.device)
            dist = torch.empty(V, n, dtype=torch.float32, device=cb.device)
            self._wait_ready()
            hip.call('neighbor_table_f32', cb, V, D, n, idx, dist)
            tabs.clear()                                        # one n resident at a time
            tabs[n] = (idx, dist)
            self._built()
        return tabs[n]

    def block(self, blk, ws, bi, x, x2, rows, l, cur):
        """one AdaLNSelfAttn block (basic_var.py:152-159): seven launches behind one library call; the result is left in x"""
        var = self.var
        C = var.C
        if self.precision != 'f32':
            b16 = self.w['b16'][self.precision][bi]
            hip.call('adaln_block_' + self.precision, x, x2, ws['xn'], ws['q'], ws['att'], ws['hid'], ws['ada_view'][bi][0], ws['ada_view'][bi][1],
                     b16['qkv_w16'], blk['qkv_b'], blk['smul'], blk['plain_scale'], int(blk['l2']), b16['proj_w16'], blk['proj_b'],
                     b16['fc1_w16'], blk['fc1_b'], b16['fc2_w16'], blk['fc2_b'], ws['kc'][bi], ws['vc'][bi],
                     rows, l, C, var.num_heads, blk['fc1_w'].shape[0], cur, var.L, var.norm_eps)
            return
        hip.call('adaln_block_f32', x, x2, ws['xn'], ws['q'], ws['att'], ws['hid'], ws['ada_view'][bi][0], ws['ada_view'][bi][1],
                 blk['qkv_w'], blk['qkv_b'], blk['smul'], blk['plain_scale'], int(blk['l2']), blk['proj_w'], blk['proj_b'],
                 blk['fc1_w'], blk['fc1_b'], blk['fc2_w'], blk['fc2_b'], ws['kc'][bi], ws['vc'][bi],
                 rows, l, C, var.num_heads, blk['fc1_w'].shape[0], cur, var.L, var.norm_eps)

    def head(self, x, hn, xn, logits, M, l):
        """get_logits (var.py:118-124): AdaLNBeforeHead (LayerNorm + scale/shift) then the vocabulary projection -> fp32 logits"""
        var, w = self.var, self.w
        C, V = var.C, var.V
        if self.precision != 'f32':
            hip.call(f'ln_modulate_{self.precision}out', x, hn, 2 * C, hn[:, C:], 2 * C, xn, M, C, l, var.norm_eps)
            hip.call('gemm_nt_' + self.precision, xn, C, w['head_w16'][self.precision], C, w['head_b'], logits, V, 0, M, V, C, EPI_NONE, None, 0, 0, None, 0, 1, 1, 0, 0, 0)
        else:
            hip.call('ln_modulate_f32', x, hn, 2 * C, hn[:, C:], 2 * C, xn, M, C, l, var.norm_eps)
            self.gemm(xn, w['head_w'], w['head_b'], logits, M)

    def qkv(self, xn, blk, ws, bi, rows, l, cur):
        """mat_qkv + q/k normalisation + KV-cache append in one launch (basic_var.py:93-109)."""
        C, H = self.var.C, self.var.num_heads
        hip.call('gemm_qkv_f32', xn, C, blk['qkv_w'], C, blk['qkv_b'], rows * l, C, C, blk['smul'], blk['plain_scale'], int(blk['l2']),
                 ws['q'], ws['kc'][bi], ws['vc'][bi], rows, l, H, cur, self.var.L)

    # -- the loop ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, B: int, label_B: torch.Tensor, rng: Optional[torch.Generator], cfg: float, top_k: int, top_p: float,
               noises=None, force_idx: Optional[torch.Tensor] = None, trace: bool = False,
               decode: bool = True, gt_tokens: Optional[torch.Tensor] = None, keep_mask: Optional[torch.Tensor] = None,
               more_smooth: bool = False, gumbel_noises=None, smooth: Optional[dict] = None) -> torch.Tensor:
        """label_B: int64 [B] on the device.  noises: optional per-scale Exp(1) tensors [B*l, V] — a list, or a callable
        (si, l) -> tensor (tests inject the CPU generator's stream; var_amd.multi hands each rank its rows); by default they
        are drawn with `exponential_(generator=rng)` exactly as torch.multinomial (helpers.py:19) would.
        gt_tokens/keep_mask [B, L]: VAR.inpainting (var.py:236-364, fork) — kept positions take the given token, the others are
        sampled; a scale whose tokens are all kept skips the head, the sampler and the RNG draw, as the reference does.
        smooth = dict(gt=[B, L] tokens, n=int, thr=float|None): VAR.smooth_sampling (var.py:367-572, fork) — no sampler, no Exp(1)
        draw: every position takes the most likely of the nearest codebook neighbours of its ground-truth token; the two
        accumulated log-likelihoods are left in self.last_smooth.
        force_idx/trace are test hooks (teacher forcing; keep per-scale logits/tokens/f_hat)."""
        var = self.var
        self.resolve_precision()
        self.refresh()
        self._wait_ready()
        w, ws = self.w, self.workspace(B)
        dev = ws['dev']
        C, H, V, Cv, S = var.C, var.num_heads, var.V, var.Cvae, len(var.patch_nums)
        P, B2 = var.patch_nums[-1], 2 * B
        if label_B.dtype != torch.int64 or label_B.numel() != B:
            raise ValueError('label_B must be an int64 tensor of B labels')
        self._check_labels(label_B)
        label_B = label_B.to(dev).contiguous()
        tr = dict(logits=[], idx=[], f_hat=[], pooled=[]) if trace else None
        gt = keep_u8 = skip = masked = None
        draws = 0
        if more_smooth:
            lmax = max(p * p for p in var.patch_nums)
            if 'probs' not in ws:
                ws['masked'] = torch.empty(B * lmax, V, dtype=torch.float32, device=dev)
                ws['probs'] = torch.empty(B * lmax, V, dtype=torch.float32, device=dev)
                ws['h'] = torch.empty(B * lmax, Cv, dtype=torch.float32, device=dev)
            masked = ws['masked']
        if gt_tokens is not None:
            if keep_mask is None or tuple(keep_mask.shape) != tuple(gt_tokens.shape) or tuple(gt_tokens.shape) != (B, var.L):
                raise ValueError('Mask shape must match the latent token shape obtained from vae.img_to_idxBl')
            gt = gt_tokens.to(dev, torch.int64).contiguous()
            if int(gt.min()) < 0 or int(gt.max()) >= V:
                raise ValueError(f'gt_tokens must lie in [0, {V})')
            keep = keep_mask.to(dev).bool()
            keep_u8 = keep.to(torch.uint8).contiguous()
            skip = torch.stack([keep[:, b0:e0].all() for b0, e0 in var.begin_ends]).tolist()      # one host sync for all scales
            if more_smooth and any(skip):
                # var.py:312-341 (fork): on a fully kept scale the reference feeds the gumbel softmax the PREVIOUS scale's logits
                # (a NameError on the first scale, a shape error afterwards) — there is no behaviour to reproduce
                raise NotImplementedError('inpainting(more_smooth=True) with a fully kept scale: undefined in the reference (it reads logits it did not compute)')

        sm_gt = sm_ll = sm_dl = None
        if smooth is not None:
            if gt_tokens is not None:
                raise ValueError('smooth sampling and inpainting are separate entry points')
            sm_gt = smooth['gt'].to(dev, torch.int64).contiguous()
            sm_n, sm_thr = int(smooth['n']), smooth.get('thr')
            if tuple(sm_gt.shape) != (B, var.L) or int(sm_gt.min()) < 0 or int(sm_gt.max()) >= V:
                raise ValueError(f'gt_tokens must be (B, L) token ids in [0, {V})')
            if not 1 <= sm_n <= V:
                raise ValueError(f'n must lie in [1, {V}]')
            nbr_idx, nbr_dist = self.neighbor_table(sm_n)
            lmax = max(p * p for p in var.patch_nums)
            sm_val = torch.empty(B * lmax, dtype=torch.float32, device=dev)
            sm_dlp = torch.empty(B * lmax, dtype=torch.float32, device=dev)
            sm_ll = torch.zeros((), dtype=torch.float32, device=dev)
            sm_dl = torch.zeros((), dtype=torch.float32, device=dev)

        # prologue (var.py:151-157)
        hip.call('lvl_pos_f32', w['lvl_embed'], w['lvl_1L'], w['pos_1LC'], ws['lvl_pos'], var.L, C)
        hip.call('first_map_f32', w['class_emb'], label_B, var.num_classes, w['pos_start'], ws['lvl_pos'], ws['cond'], ws['x'], B, C, var.first_l)
        hip.call('silu_f32', ws['cond'], ws['cond_silu'], B2 * C)
        ws['f_hat'].zero_()
        # AdaLN parameters of every block, once per call
        if var.shared_aln:
            self.gemm(ws['cond_silu'], w['sal_w'], w['sal_b'], ws['shared'], B2)
        if var.shared_aln:
            for bi, blk in enumerate(w['blocks']):
                hip.call('add_bcast_f32', blk['gss'], ws['shared'], ws['ada'][bi], B2, 6 * C)
            ws['ada_view'] = [(ws['ada'][bi], 6 * C) for bi in range(var.depth)]
        elif 'ada_w_all' in w:
            self.gemm(ws['cond_silu'], w['ada_w_all'], w['ada_b_all'], ws['ada'], B2)          # row b: [block 0: 6C | block 1: 6C | ...]
            ws['ada_view'] = [(ws['ada'][:, bi * 6 * C:], var.depth * 6 * C) for bi in range(var.depth)]
        else:
            ws['ada_view'] = [(ws['ada'][:, bi * 6 * C:], var.depth * 6 * C) for bi in range(var.depth)]
            for bi, blk in enumerate(w['blocks']):
                hip.call('gemm_nt_f32', ws['cond_silu'], C, blk['ada_w'], C, blk['ada_b'], ws['ada_view'][bi][0], var.depth * 6 * C, B2, 6 * C, C, EPI_NONE,
                         None, 0, None, 0, 1, 0, 1, 0, 0, 0)
        self.gemm(ws['cond_silu'], w['hn_w'], w['hn_b'], ws['hn'], B2)

        x, x2 = ws['x'], ws['x2']
        cur = 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(S + 2)] if getattr(self, 'profile_scales', False) else None
        for si, pn in enumerate(var.patch_nums):                          # var.py:160
            l = pn * pn
            M = B2 * l
            if ev: ev[si].record()
            for bi, blk in enumerate(w['blocks']):                        # AdaLNSelfAttn.forward, basic_var.py:152-159
                self.block(blk, ws, bi, x, x2, B2, l, cur)
            cur += l
            if skip is not None and skip[si]:
                # inpainting, every token of this scale is kept: no head, no sampling, no RNG draw (var.py:312-313, fork)
                idx = gt[:, cur - l:cur].contiguous().view(-1)
                if trace: tr['logits'].append(None); tr['idx'].append(idx.view(B, l).clone())
            else:
                # get_logits (var.py:118-124): AdaLNBeforeHead + head
                self.head(x, ws['hn'], ws['xn'], ws['logits'], M, l)
                if trace: tr['logits'].append(ws['logits'][:M].view(B2, l, V).clone())
                t = cfg * (si / var.num_stages_minus_1) if var.num_stages_minus_1 > 0 else 0.0
                idx = ws['idx'][:B * l]
                if sm_gt is not None:
                    # neighbour-candidate selection instead of sampling (var.py:484-537, fork); `masked` receives the CFG logits
                    r_ = si / var.num_stages_minus_1 if var.num_stages_minus_1 > 0 else 0.0
                    hip.call('smooth_select_f32', ws['logits'], sm_gt[:, cur - l:cur].contiguous(), nbr_idx, nbr_dist, sm_n, 1 + int((sm_n - 1) * r_),
                             int(sm_thr is not None), float(sm_thr if sm_thr is not None else 0.0), float(r_), B, l, V, float(t),
                             idx, sm_val, sm_dlp, masked)
                    # var.py:537: new_tensor(max_vals) has the tokens' dtype, so every value is truncated to an integer before the sum
                    sm_ll = sm_ll + sm_val[:B * l].to(torch.int64).sum()
                    sm_dl = sm_dl + sm_dlp[:B * l].sum()
                else:
                    # CFG + top-k/top-p + multinomial (var.py:172-175)
                    if noises is not None:        # a list is indexed by draw count: skipped (fully kept) scales draw nothing
                        noise = (noises(si, l) if callable(noises) else noises[draws]).to(dev, torch.float32).contiguous()
                        draws += 1
                    else:
                        noise = torch.empty(B * l, V, dtype=torch.float32, device=dev).exponential_(1, generator=rng)
                    hip.call('cfg_sample_f32', ws['logits'], noise, idx, masked, B, l, V, float(t), int(top_k), float(top_p))
                if gt is not None:                                        # torch.where(mask, gt_tokens, sampled) (var.py:326-328)
                    hip.call('token_select_i64', keep_u8[:, cur - l:cur].contiguous(), gt[:, cur - l:cur].contiguous(), idx, idx, B * l)
                if trace: tr['idx'].append(idx.view(B, l).clone())
            if force_idx is not None:
                idx = force_idx[:, cur - l:cur].to(dev, torch.int64).contiguous().view(-1)
            # quantizer step (var.py:177-183)
            ti, tw = w['taps'].get(pn, (None, None))
            pw, pb, ratio = w['phi'][phi_index(si, S, len(w['phi']))]
            if more_smooth:
                # h = gumbel_softmax(filtered logits * (1+ratio), tau) @ codebook, a second Exp(1) fill per scale (var.py:178-180)
                r_ = si / var.num_stages_minus_1 if var.num_stages_minus_1 > 0 else 0.0
                gum_t = max(0.27 * (1 - r_ * 0.95), 0.005)
                if gumbel_noises is not None:
                    gn = gumbel_noises[si].to(dev, torch.float32).contiguous()
                else:
                    gn = torch.empty(B * l, V, dtype=torch.float32, device=dev).exponential_(generator=rng)
                hip.call('gumbel_softmax_f32', masked, gn, ws['probs'], B * l, V, float(1 + r_), float(gum_t))
                self.gemm(ws['probs'], w['codebook_T'], None, ws['h'], B * l)
                hip.call('quant_accum_h_f32', ws['h'], ti, tw, pw, pb, ratio, ws['up'], ws['f_hat'], B, pn, P, Cv)
            else:
                hip.call('quant_accum_f32', idx, w['codebook'], ti, tw, pw, pb, ratio, ws['up'], ws['f_hat'], B, pn, P, Cv)
            if trace: tr['f_hat'].append(ws['f_hat'].permute(0, 3, 1, 2).clone())
            if si != S - 1:
                pq = var.patch_nums[si + 1]
                hip.call('next_map_f32', ws['f_hat'], w['word_w'], w['word_b'], ws['lvl_pos'][cur:], x, ws['pooled'], B, P, pq, C, Cv)
                if trace: tr['pooled'].append(ws['pooled'][:B * pq * pq].view(B, pq, pq, Cv).permute(0, 3, 1, 2).clone())
        self.last_trace = tr
        self.last_smooth = (sm_ll, sm_dl) if sm_gt is not None else None
        if not decode:
            return ws['f_hat'].permute(0, 3, 1, 2).contiguous()
        if ev: ev[S].record()
        img = self.dec.decode_nhwc(ws['f_hat'], precision=self.precision)     # var.py:190
        if ev:                                                            # tools/per_scale.py: ms per scale (blocks + head + sampler + quantizer step), then the decoder
            ev[S + 1].record(); torch.cuda.synchronize()
            self.last_scale_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(S + 1)]
        return img

    # -- teacher-forced logits (VAR.forward without autograd) ------------------------------------------------------------
    @torch.no_grad()
    def teacher_forced_logits(self, label_B: torch.Tensor, x_BLCv_wo_first_l: Optional[torch.Tensor]) -> torch.Tensor:
        """logits (B, L, V) of VAR.forward (reference var.py:192-234) for given next-scale inputs, computed scale by scale over the
        KV cache instead of one masked pass: the cache holds exactly the scales <= the current one, which is what the block-causal
        mask `attn_bias_for_masking` allows (SURVEY.md §4 identity (i): identical to 3e-8 in the reference itself).
        No CFG doubling: B rows.  label_B may contain num_classes (dropped condition)."""
        var = self.var
        self.resolve_precision()
        self.refresh()
        self._wait_ready()
        w = self.w
        R = int(label_B.numel())                               # rows: one per image, no CFG pair
        dev = var.pos_start.device
        C, H, V, Cv, L = var.C, var.num_heads, var.V, var.Cvae, var.L
        self._check_labels(label_B)
        lmax = max(p * p for p in var.patch_nums)
        hid = var.blocks[0].ffn.fc1.weight.shape[0]
        sid = int(torch.cuda.current_stream().cuda_stream)
        ws = self._ws_tf.get((R, sid, self.precision))
        if ws is None or ws['dev'] != dev:
            e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
            act = DT16.get(self.precision, torch.float32)
            M = R * lmax
            # x is written by first_map_f32 / word_embed_f32, which also emit the CFG copy of every row (unused here): room for 2x
            ws = dict(dev=dev, x=e(2 * M, C), x2=e(M, C), xn=e(M, C, dt=act), q=e(M, C, dt=act), att=e(M, C, dt=act), hid=e(M, hid, dt=act), lg=e(R * lmax, V),
                      lvl_pos=e(L, C), cond=e(2 * R, C), cond_silu=e(2 * R, C), hn=e(R, 2 * C), ada=e(var.depth, R, 6 * C) if var.shared_aln else e(R, var.depth * 6 * C),
                      shared=e(R, 6 * C) if var.shared_aln else None,
                      kc=[torch.zeros(R, H, L, 64, dtype=act, device=dev) for _ in range(var.depth)],
                      vc=[torch.zeros(R, H, L, 64, dtype=act, device=dev) for _ in range(var.depth)])
            self._ws_tf = self._evict(self._ws_tf, sid)
            self._ws_tf[(R, sid, self.precision)] = ws
        lab = label_B.to(dev).long().contiguous()
        hip.call('lvl_pos_f32', w['lvl_embed'], w['lvl_1L'], w['pos_1LC'], ws['lvl_pos'], L, C)
        hip.call('first_map_f32', w['class_emb'], lab, var.num_classes, w['pos_start'], ws['lvl_pos'], ws['cond'], ws['x'], R, C, var.first_l)
        hip.call('silu_f32', ws['cond'], ws['cond_silu'], R * C)
        if var.shared_aln:
            self.gemm(ws['cond_silu'], w['sal_w'], w['sal_b'], ws['shared'], R)
        if var.shared_aln:
            for bi, blk in enumerate(w['blocks']):
                hip.call('add_bcast_f32', blk['gss'], ws['shared'], ws['ada'][bi], R, 6 * C)
            ws['ada_view'] = [(ws['ada'][bi], 6 * C) for bi in range(var.depth)]
        elif 'ada_w_all' in w:
            self.gemm(ws['cond_silu'], w['ada_w_all'], w['ada_b_all'], ws['ada'], R)
            ws['ada_view'] = [(ws['ada'][:, bi * 6 * C:], var.depth * 6 * C) for bi in range(var.depth)]
        else:
            ws['ada_view'] = [(ws['ada'][:, bi * 6 * C:], var.depth * 6 * C) for bi in range(var.depth)]
            for bi, blk in enumerate(w['blocks']):
                hip.call('gemm_nt_f32', ws['cond_silu'], C, blk['ada_w'], C, blk['ada_b'], ws['ada_view'][bi][0], var.depth * 6 * C, R, 6 * C, C, EPI_NONE,
                         None, 0, None, 0, 1, 0, 1, 0, 0, 0)
        self.gemm(ws['cond_silu'], w['hn_w'], w['hn_b'], ws['hn'], R)
        out = torch.empty(R, L, V, dtype=torch.float32, device=dev)
        x, x2 = ws['x'], ws['x2']
        cur = 0
        xin = None if x_BLCv_wo_first_l is None else x_BLCv_wo_first_l.to(dev, torch.float32).contiguous()
        for si, pn in enumerate(var.patch_nums):
            l = pn * pn
            M = R * l
            if si > 0:                                           # word_embed(teacher-forcing input) + lvl_pos  (var.py:206-207)
                seg = xin[:, cur - var.first_l:cur - var.first_l + l].contiguous()
                hip.call('word_embed_f32', seg, w['word_w'], w['word_b'], ws['lvl_pos'][cur:], x, R, l, C, Cv)
            for bi, blk in enumerate(w['blocks']):
                self.block(blk, ws, bi, x, x2, R, l, cur)
            self.head(x, ws['hn'], ws['xn'], ws['lg'], M, l)
            out[:, cur:cur + l] = ws['lg'][:M].view(R, l, V)
            cur += l
        return out

    # -- model arithmetic (for bench.py's roofline) -------------------------------------------------------------------
    def flops_per_image(self) -> float:
        """Algorithmic FLOPs of one image (2 per MAC, both CFG branches), SURVEY.md §8(d) formula; ada_lin hoisted."""
        var = self.var
        C, depth, V = var.C, var.depth, var.V
        L = var.L
        T = 2 * L
        sig = 0; cur = 0
        for pn in var.patch_nums:
            cur += pn * pn; sig += pn * pn * cur
        lin = 2 * 12 * C * C * depth * T
        att = 2 * 2 * sig * C * depth * 2
        head = 2 * C * V * T
        ada = 2 * 2 * depth * 6 * C * C
        return float(lin + att + head + ada)
