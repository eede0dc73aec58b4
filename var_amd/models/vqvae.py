"""VQVAE wrapper with the reference's constructor, methods and state-dict (models/vqvae.py:16-103).

`fhat_to_img` — the decode half of the sampling path — runs the HIP decoder (var_amd.engine.DecoderEngine) whenever it is
handed an fp32 CUDA tensor outside autograd; anything else (CPU tensors, autograd, half precision) takes the PyTorch
modules, which exist for the encode side and for API users, not for the measured path."""
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from .basic_vae import Decoder, Encoder
from .quant import VectorQuantizer2


class VQVAE(nn.Module):
    def __init__(self, vocab_size=4096, z_channels=32, ch=128, dropout=0.0, beta=0.25, using_znorm=False, quant_conv_ks=3, quant_resi=0.5,
                 share_quant_resi=4, default_qresi_counts=0, v_patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), test_mode=True):
        super().__init__()
        self.test_mode = test_mode
        self.V, self.Cvae = vocab_size, z_channels
        ddconfig = dict(dropout=dropout, ch=ch, z_channels=z_channels, in_channels=3, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
                        using_sa=True, using_mid_sa=True)
        self.encoder = Encoder(double_z=False, **ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.vocab_size = vocab_size
        self.downsample = 2 ** (len(ddconfig['ch_mult']) - 1)
        self.quantize = VectorQuantizer2(vocab_size=vocab_size, Cvae=self.Cvae, using_znorm=using_znorm, beta=beta,
                                         default_qresi_counts=default_qresi_counts, v_patch_nums=v_patch_nums, quant_resi=quant_resi,
                                         share_quant_resi=share_quant_resi)
        self.quant_conv = nn.Conv2d(self.Cvae, self.Cvae, quant_conv_ks, stride=1, padding=quant_conv_ks // 2)
        self.post_quant_conv = nn.Conv2d(self.Cvae, self.Cvae, quant_conv_ks, stride=1, padding=quant_conv_ks // 2)
        self._hip_decoder = None
        self._hip_encoder = None
        self._hip_generation = 0            # bumped by invalidate_engines(); part of the sampling engine's weight signature
        if self.test_mode:
            self.eval()
            for p in self.parameters(): p.requires_grad_(False)

    # ---- decode -----------------------------------------------------------------------------------------------------
    def _decoder_engine(self):
        if self._hip_decoder is None:
            from ..engine import DecoderEngine
            self._hip_decoder = DecoderEngine(self)
        return self._hip_decoder

    def fhat_to_img(self, f_hat: torch.Tensor):
        """f_hat (B, Cvae, h, w) -> image (B, 3, 16h, 16w) clamped to [-1, 1]   (reference vqvae.py:62-63)"""
        if f_hat.is_cuda and f_hat.dtype == torch.float32 and not (torch.is_grad_enabled() and f_hat.requires_grad):
            B, C, H, W = f_hat.shape
            from .. import hip
            nhwc = torch.empty(B, H, W, C, dtype=torch.float32, device=f_hat.device)
            hip.call('nchw_to_nhwc_f32', f_hat.contiguous(), nhwc, B, C, H * W)
            return self._decoder_engine().decode_nhwc(nhwc, denorm=False)     # clamp(-1,1) fused into the last conv
        return self.decoder(self.post_quant_conv(f_hat)).clamp_(-1, 1)

    # ---- encode side / utilities: PyTorch, API kept (reference vqvae.py:56-98) ---------------------------------------
    def forward(self, inp, ret_usages=False):
        raise NotImplementedError('VQVAE.forward is VAE training (reference vqvae.py:56-59): out of scope of the sampling-path build')

    def _encoder_engine(self):
        if self._hip_encoder is None:
            from ..engine import EncoderEngine
            self._hip_encoder = EncoderEngine(self)
        return self._hip_encoder

    def img_to_post(self, inp_img_no_grad, v_patch_nums=None):
        """quant_conv(encoder(img)): (B, Cvae, H/16, W/16).  HIP encoder for fp32 CUDA images outside autograd, PyTorch otherwise."""
        x = inp_img_no_grad
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] % 16 == 0 and x.shape[-2] % 16 == 0
                and self.quant_conv.kernel_size == (3, 3)):
            f = self._encoder_engine().encode(x)                         # channels-last (B, h, w, Cvae)
            from .. import hip
            B, h, w, C = f.shape
            out = torch.empty(B, C, h, w, dtype=torch.float32, device=x.device)
            hip.call('nhwc_to_nchw_f32', f, out, B, C, h * w)
            return out
        return self.quant_conv(self.encoder(x))

    def img_to_idxBl(self, inp_img_no_grad, v_patch_nums=None) -> List[torch.LongTensor]:
        return self.quantize.f_to_idxBl_or_fhat(self.img_to_post(inp_img_no_grad), to_fhat=False, v_patch_nums=v_patch_nums)

    def img_to_fhat(self, inp_img_no_grad, v_patch_nums=None):
        return self.quantize.f_to_idxBl_or_fhat(self.img_to_post(inp_img_no_grad), to_fhat=True, v_patch_nums=v_patch_nums)

    def idxBl_to_img(self, ms_idx_Bl: List[torch.Tensor], same_shape: bool, last_one=False):
        q = self.quantize
        if (same_shape and len(ms_idx_Bl) == len(q.v_patch_nums) and q._hip_eligible(ms_idx_Bl[0])
                and all(i.shape[1] == pn * pn for i, pn in zip(ms_idx_Bl, q.v_patch_nums))):
            # codebook lookup + bicubic + Phi + accumulate fused on HIP (what the sampling loop runs per scale), then the HIP decoder
            f = q.hip_engine().fhat_from_scales(list(ms_idx_Bl), tuple(q.v_patch_nums), from_tokens=True, last_one=last_one)
            return self.fhat_to_img(f) if last_one else [self.fhat_to_img(x) for x in f]
        B = ms_idx_Bl[0].shape[0]
        hs = []
        for idx_Bl in ms_idx_Bl:
            pn = round(idx_Bl.shape[1] ** 0.5)
            hs.append(self.quantize.embedding(idx_Bl).transpose(1, 2).view(B, self.Cvae, pn, pn))
        return self.embed_to_img(ms_h_BChw=hs, all_to_max_scale=same_shape, last_one=last_one)

    def embed_to_img(self, ms_h_BChw: List[torch.Tensor], all_to_max_scale: bool, last_one=False):
        f = self.quantize.embed_to_fhat(ms_h_BChw, all_to_max_scale=all_to_max_scale, last_one=last_one)
        return self.fhat_to_img(f) if last_one else [self.fhat_to_img(x) for x in f]

    def img_to_reconstructed_img(self, x, v_patch_nums=None, last_one=False):
        fs = self.quantize.f_to_idxBl_or_fhat(self.img_to_post(x), to_fhat=True, v_patch_nums=v_patch_nums)
        return self.fhat_to_img(fs[-1]) if last_one else [self.fhat_to_img(f) for f in fs]

    def load_state_dict(self, state_dict: Dict[str, Any], strict=True, assign=False):
        k = 'quantize.ema_vocab_hit_SV'        # checkpoints trained with another number of scales: keep ours (reference vqvae.py:100-103)
        if k in state_dict and state_dict[k].shape[0] != self.quantize.ema_vocab_hit_SV.shape[0]:
            state_dict[k] = self.quantize.ema_vocab_hit_SV
        ret = super().load_state_dict(state_dict=state_dict, strict=strict, assign=assign)
        self.invalidate_engines()
        return ret

    def invalidate_engines(self):
        """the HIP engines keep re-laid copies of the conv / Phi / codebook weights keyed on (address, version counter); edits through
        `.data` bump no counter — call this after them (load_state_dict does)"""
        self._hip_generation += 1
        for e in (self._hip_decoder, self._hip_encoder, getattr(self.quantize, '_hip_engine', None)):
            if e is not None:
                e.invalidate()
