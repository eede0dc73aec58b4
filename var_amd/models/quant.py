"""Multi-scale residual vector quantizer with the reference's interface (models/quant.py:15-243).

The GPU sampling loop does NOT go through these methods: var_amd.engine runs the fused HIP quantizer step
(varhip_quant_accum_f32 / varhip_next_map_f32) on `embedding` and the Phi convolutions held here.  The PyTorch methods
below keep the encode / teacher-forcing API (trainer.py:105-111, fork scripts) available on any device."""
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

__all__ = ['VectorQuantizer2']


class Phi(nn.Conv2d):
    """h -> (1-r) h + r conv3x3(h)   (reference quant.py:199-206)"""
    def __init__(self, embed_dim, quant_resi):
        super().__init__(embed_dim, embed_dim, kernel_size=3, stride=1, padding=1)
        self.resi_ratio = abs(quant_resi)

    def forward(self, h_BChw):
        return h_BChw.mul(1 - self.resi_ratio) + super().forward(h_BChw).mul_(self.resi_ratio)


def _ticks(K):
    return np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)


class PhiShared(nn.Module):
    def __init__(self, qresi: Phi):
        super().__init__()
        self.qresi = qresi

    def __getitem__(self, _):
        return self.qresi

    def phis(self):
        return [self.qresi]


class PhiPartiallyShared(nn.Module):
    def __init__(self, qresi_ls: nn.ModuleList):
        super().__init__()
        self.qresi_ls = qresi_ls
        self.ticks = _ticks(len(qresi_ls))

    def __getitem__(self, at_from_0_to_1: float):
        return self.qresi_ls[np.argmin(np.abs(self.ticks - at_from_0_to_1)).item()]

    def phis(self):
        return list(self.qresi_ls)

    def extra_repr(self) -> str:
        return f'ticks={self.ticks}'


class PhiNonShared(nn.ModuleList):
    def __init__(self, qresi: List):
        super().__init__(qresi)
        self.ticks = _ticks(len(qresi))

    def __getitem__(self, at_from_0_to_1: float):
        return super().__getitem__(np.argmin(np.abs(self.ticks - at_from_0_to_1)).item())

    def phis(self):
        return [super(PhiNonShared, self).__getitem__(i) for i in range(len(self))]


class VectorQuantizer2(nn.Module):
    def __init__(self, vocab_size, Cvae, using_znorm, beta: float = 0.25, default_qresi_counts=0, v_patch_nums=None, quant_resi=0.5, share_quant_resi=4):
        super().__init__()
        self.vocab_size, self.Cvae, self.using_znorm, self.v_patch_nums = vocab_size, Cvae, using_znorm, v_patch_nums
        self.quant_resi_ratio = quant_resi
        mk = lambda: Phi(Cvae, quant_resi) if abs(quant_resi) > 1e-6 else nn.Identity()
        if share_quant_resi == 0:
            self.quant_resi = PhiNonShared([mk() for _ in range(default_qresi_counts or len(self.v_patch_nums))])
        elif share_quant_resi == 1:
            self.quant_resi = PhiShared(mk())
        else:
            self.quant_resi = PhiPartiallyShared(nn.ModuleList([mk() for _ in range(share_quant_resi)]))
        self.register_buffer('ema_vocab_hit_SV', torch.full((len(self.v_patch_nums), self.vocab_size), fill_value=0.0))
        self.record_hit = 0
        self.beta = beta
        self.embedding = nn.Embedding(self.vocab_size, self.Cvae)
        self.prog_si = -1

    def eini(self, eini):
        if eini > 0: nn.init.trunc_normal_(self.embedding.weight.data, std=eini)
        elif eini < 0: self.embedding.weight.data.uniform_(-abs(eini) / self.vocab_size, abs(eini) / self.vocab_size)

    def extra_repr(self) -> str:
        return f'{self.v_patch_nums}, znorm={self.using_znorm}, beta={self.beta}  |  S={len(self.v_patch_nums)}, quant_resi={self.quant_resi_ratio}'

    def forward(self, f_BChw, ret_usages=False):
        raise NotImplementedError('VectorQuantizer2.forward is VAE *training* (reference quant.py:52-104): out of scope of the sampling-path build')

    # ---- helpers shared by the methods below ---------------------------------------------------------------------
    def _nearest(self, z_NC: torch.Tensor) -> torch.Tensor:
        E = self.embedding.weight.data
        if self.using_znorm:
            return torch.argmax(F.normalize(z_NC, dim=-1) @ F.normalize(E.T, dim=0), dim=1)
        d = z_NC.square().sum(1, keepdim=True) + E.square().sum(1)
        d.addmm_(z_NC, E.T, alpha=-2, beta=1)
        return torch.argmin(d, dim=1)

    def _lift(self, h_BChw, si, SN, HW):
        """bicubic up to the final size (except the last scale) followed by Phi"""
        if si != SN - 1:
            h_BChw = F.interpolate(h_BChw, size=HW, mode='bicubic')
        return self.quant_resi[si / (SN - 1)](h_BChw.contiguous())

    def embed_to_fhat(self, ms_h_BChw: List[torch.Tensor], all_to_max_scale=True, last_one=False):
        """reference quant.py:107-133.  fp32 CUDA maps outside autograd run the HIP quantizer kernels (the same ones, in the same order,
        as the sampling loop's incremental f_hat: identical bits); anything else the PyTorch code below."""
        if (all_to_max_scale and len(ms_h_BChw) == len(self.v_patch_nums) and self._hip_eligible(ms_h_BChw[0])
                and all(h.dtype == torch.float32 and h.shape[-1] == pn and h.shape[-2] == pn for h, pn in zip(ms_h_BChw, self.v_patch_nums))):
            return self.hip_engine().fhat_from_scales(list(ms_h_BChw), tuple(self.v_patch_nums), from_tokens=False, last_one=last_one)
        B, SN = ms_h_BChw[0].shape[0], len(self.v_patch_nums)
        outs = []
        if all_to_max_scale:
            H = W = self.v_patch_nums[-1]
            f_hat = ms_h_BChw[0].new_zeros(B, self.Cvae, H, W, dtype=torch.float32)
            for si in range(SN):
                f_hat.add_(self._lift(ms_h_BChw[si], si, SN, (H, W)))
                outs.append(f_hat.clone())
        else:   # experimental path of the reference: grow the canvas scale by scale
            f_hat = ms_h_BChw[0].new_zeros(B, self.Cvae, self.v_patch_nums[0], self.v_patch_nums[0], dtype=torch.float32)
            for si, pn in enumerate(self.v_patch_nums):
                f_hat = F.interpolate(f_hat, size=(pn, pn), mode='bicubic')
                f_hat.add_(self.quant_resi[si / (SN - 1)](ms_h_BChw[si]))
                outs.append(f_hat)
        return outs[-1] if last_one else outs

    # ---- HIP routing of the encode-side methods (inference on fp32 CUDA tensors); everything else takes the PyTorch code below ----
    def hip_engine(self):
        if getattr(self, '_hip_engine', None) is None:
            from ..engine import QuantizerEngine
            object.__setattr__(self, '_hip_engine', QuantizerEngine(self))
        return self._hip_engine

    def _hip_eligible(self, t: torch.Tensor, v_patch_nums=None) -> bool:
        pns = v_patch_nums or self.v_patch_nums
        return (t.is_cuda and not torch.is_grad_enabled() and self.prog_si < 0 and self.embedding.weight.is_cuda
                and self.embedding.weight.dtype == torch.float32 and isinstance(self.quant_resi[0.0], Phi) and all(isinstance(p, int) for p in pns))

    def f_to_idxBl_or_fhat(self, f_BChw: torch.Tensor, to_fhat: bool, v_patch_nums: Optional[Sequence[Union[int, Tuple[int, int]]]] = None):
        """residual quantisation of an encoder feature map, scale by scale (reference quant.py:135-166)"""
        B, C, H, W = f_BChw.shape
        if f_BChw.dtype == torch.float32 and H == W and self._hip_eligible(f_BChw, v_patch_nums):
            from .. import hip
            pns = tuple(v_patch_nums or self.v_patch_nums)
            assert pns[-1] == H, f'{pns[-1]=} != ({H=}, {W=})'
            nhwc = torch.empty(B, H, W, C, dtype=torch.float32, device=f_BChw.device)
            hip.call('nchw_to_nhwc_f32', f_BChw.contiguous(), nhwc, B, C, H * W)
            return self.hip_engine().quantize(nhwc, to_fhat, pns)
        f_rest = f_BChw.detach().clone()
        f_hat = torch.zeros_like(f_rest)
        hws = [(pn, pn) if isinstance(pn, int) else (pn[0], pn[1]) for pn in (v_patch_nums or self.v_patch_nums)]
        assert hws[-1][0] == H and hws[-1][1] == W, f'{hws[-1]=} != ({H=}, {W=})'
        SN, out = len(hws), []
        for si, (ph, pw) in enumerate(hws):
            if 0 <= self.prog_si < si: break
            z = F.interpolate(f_rest, size=(ph, pw), mode='area') if si != SN - 1 else f_rest
            idx_N = self._nearest(z.permute(0, 2, 3, 1).reshape(-1, C))
            h = self._lift(self.embedding(idx_N.view(B, ph, pw)).permute(0, 3, 1, 2), si, SN, (H, W))
            f_hat.add_(h); f_rest.sub_(h)
            out.append(f_hat.clone() if to_fhat else idx_N.reshape(B, ph * pw))
        return out

    def idxBl_to_var_input(self, gt_ms_idx_Bl: List[torch.Tensor]) -> torch.Tensor:
        """teacher-forcing input of VAR.forward (reference quant.py:169-184)"""
        if len(gt_ms_idx_Bl) == len(self.v_patch_nums) and self._hip_eligible(gt_ms_idx_Bl[0]):
            return self.hip_engine().var_input(list(gt_ms_idx_Bl), tuple(self.v_patch_nums))
        B, C, SN = gt_ms_idx_Bl[0].shape[0], self.Cvae, len(self.v_patch_nums)
        H = W = self.v_patch_nums[-1]
        f_hat = gt_ms_idx_Bl[0].new_zeros(B, C, H, W, dtype=torch.float32)
        nxt = []
        for si in range(SN - 1):
            if self.prog_si == 0 or (0 <= self.prog_si - 1 < si): break
            pn, pq = self.v_patch_nums[si], self.v_patch_nums[si + 1]
            h = F.interpolate(self.embedding(gt_ms_idx_Bl[si]).transpose(1, 2).reshape(B, C, pn, pn), size=(H, W), mode='bicubic')
            f_hat.add_(self.quant_resi[si / (SN - 1)](h))
            nxt.append(F.interpolate(f_hat, size=(pq, pq), mode='area').view(B, C, -1).transpose(1, 2))
        return torch.cat(nxt, dim=1) if nxt else None

    def get_next_autoregressive_input(self, si: int, SN: int, f_hat: torch.Tensor, h_BChw: torch.Tensor):
        """one quantizer step in PyTorch (reference quant.py:187-196); the engine's HIP step is what sampling uses"""
        HW = self.v_patch_nums[-1]
        f_hat.add_(self._lift(h_BChw, si, SN, (HW, HW)))
        if si == SN - 1:
            return f_hat, f_hat
        pq = self.v_patch_nums[si + 1]
        return f_hat, F.interpolate(f_hat, size=(pq, pq), mode='area')
