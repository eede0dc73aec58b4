"""Transformer blocks of VAR as parameter containers + a plain-PyTorch teacher-forced forward.

State-dict names/shapes are the reference's (models/basic_var.py:33-174), so its checkpoints load with strict=True.
Sampling never runs these `forward`s: VAR.autoregressive_infer_cfg reads the parameters and runs the HIP kernels of
var_amd/engine.py.  The `forward`s below exist for the autograd/teacher-forcing API that trainer.py uses (VAR.forward)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .helpers import DropPath

__all__ = ['FFN', 'AdaLNSelfAttn', 'AdaLNBeforeHead', 'SelfAttention']

_LOG_100 = math.log(100.0)          # cap of the learnt per-head temperature (reference basic_var.py:70)


def _dropout_or_identity(p: float) -> nn.Module:
    return nn.Dropout(p, inplace=True) if p > 0 else nn.Identity()


def _modulate(normed: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """AdaLN: LN(x) * (1 + scale) + shift (reference basic_var.py:157-158,174)"""
    return normed * (scale + 1) + shift


class FFN(nn.Module):
    """fc1 -> GELU(tanh) -> fc2 (reference basic_var.py:33-55); parameters fc1.{weight,bias}, fc2.{weight,bias}"""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0., fused_if_available=True):
        super().__init__()
        width_mid = in_features if not hidden_features else hidden_features
        width_out = in_features if not out_features else out_features
        self.fused_mlp_func = None                      # the reference's optional flash-attn fused MLP; never present here
        self.fc1 = nn.Linear(in_features, width_mid)
        self.act = nn.GELU(approximate='tanh')
        self.fc2 = nn.Linear(width_mid, width_out)
        self.drop = _dropout_or_identity(drop)

    def forward(self, x):
        hidden = self.act(self.fc1(x))
        return self.drop(self.fc2(hidden))


class SelfAttention(nn.Module):
    """Multi-head attention with optional q/k L2 normalisation and a learnt per-head temperature (reference basic_var.py:58-121).
    Parameters: scale_mul_1H11 (only with attn_l2_norm), mat_qkv.weight (no bias), q_bias, v_bias, buffer zero_k_bias, proj.*"""

    def __init__(self, block_idx, embed_dim=768, num_heads=12, attn_drop=0., proj_drop=0., attn_l2_norm=False, flash_if_available=True):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError('embed_dim must be a multiple of num_heads')
        self.block_idx = block_idx
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.attn_l2_norm = attn_l2_norm
        if not attn_l2_norm:
            self.scale = 0.25 / math.sqrt(self.head_dim)
        else:
            self.scale = 1
            self.scale_mul_1H11 = nn.Parameter(torch.full((1, num_heads, 1, 1), math.log(4.0)), requires_grad=True)
            self.max_scale_mul = _LOG_100
        self.mat_qkv = nn.Linear(embed_dim, 3 * embed_dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(embed_dim))
        self.v_bias = nn.Parameter(torch.zeros(embed_dim))
        self.register_buffer('zero_k_bias', torch.zeros(embed_dim))
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = _dropout_or_identity(proj_drop)
        self.attn_drop = attn_drop
        self.using_flash = self.using_xform = False     # third-party kernels of the reference: not used by this build
        self.kv_caching(False)

    def kv_caching(self, enable: bool):
        """API kept for callers that toggle it (notebooks); the HIP path owns a pre-allocated cache in the engine."""
        self.caching = enable
        self.cached_k = self.cached_v = None

    def _heads(self, x):
        """x (B, L, C) -> q, k, v each (B, H, L, c); the k part of the projection has no bias"""
        B, L, _ = x.shape
        bias = torch.cat((self.q_bias, self.zero_k_bias, self.v_bias))
        packed = F.linear(x, self.mat_qkv.weight, bias).view(B, L, 3, self.num_heads, self.head_dim)
        return packed.permute(2, 0, 3, 1, 4).unbind(0)

    def forward(self, x, attn_bias):
        B, L, C = x.shape
        q, k, v = self._heads(x)
        if self.attn_l2_norm:
            temperature = self.scale_mul_1H11.clamp_max(self.max_scale_mul).exp()
            q, k = F.normalize(q, dim=-1) * temperature, F.normalize(k, dim=-1)
        if self.caching:
            if self.cached_k is not None:
                k = torch.cat((self.cached_k, k), dim=2)
                v = torch.cat((self.cached_v, v), dim=2)
            self.cached_k, self.cached_v = k, v
        p_drop = self.attn_drop if self.training else 0.0
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_bias, dropout_p=p_drop, scale=self.scale)
        return self.proj_drop(self.proj(ctx.transpose(1, 2).reshape(B, L, C)))

    def extra_repr(self) -> str:
        return f'attn_l2_norm={self.attn_l2_norm}'


class AdaLNSelfAttn(nn.Module):
    """Pre-LN block whose two branches are modulated and gated by six class-conditional vectors (reference basic_var.py:124-163):
    either `ada_lin` (SiLU + Linear(D, 6C)) or, with shared_aln, the parameter `ada_gss` (1,1,6,C) added to a shared projection."""

    def __init__(self, block_idx, last_drop_p, embed_dim, cond_dim, shared_aln: bool, norm_layer, num_heads, mlp_ratio=4., drop=0.,
                 attn_drop=0., drop_path=0., attn_l2_norm=False, flash_if_available=False, fused_if_available=True):
        super().__init__()
        self.block_idx, self.last_drop_p = block_idx, last_drop_p
        self.C, self.D = embed_dim, cond_dim
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.attn = SelfAttention(block_idx=block_idx, embed_dim=embed_dim, num_heads=num_heads, attn_drop=attn_drop, proj_drop=drop,
                                  attn_l2_norm=attn_l2_norm, flash_if_available=flash_if_available)
        self.ffn = FFN(in_features=embed_dim, hidden_features=round(mlp_ratio * embed_dim), drop=drop, fused_if_available=fused_if_available)
        self.ln_wo_grad = norm_layer(embed_dim, elementwise_affine=False)
        self.shared_aln = shared_aln
        if not shared_aln:
            self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(cond_dim, 6 * embed_dim))
        else:
            self.ada_gss = nn.Parameter(torch.randn(1, 1, 6, embed_dim) / math.sqrt(embed_dim))
        self.fused_add_norm_fn = None

    def _six(self, cond_BD):
        """(gamma1, gamma2, scale1, scale2, shift1, shift2), each broadcastable to (B, L, C)"""
        if self.shared_aln:
            return (cond_BD + self.ada_gss).unbind(2)
        return self.ada_lin(cond_BD).view(-1, 1, 6, self.C).unbind(2)

    def forward(self, x, cond_BD, attn_bias):
        gamma1, gamma2, scale1, scale2, shift1, shift2 = self._six(cond_BD)
        branch = self.attn(_modulate(self.ln_wo_grad(x), scale1, shift1), attn_bias=attn_bias)
        x = x + self.drop_path(branch * gamma1)
        branch = self.ffn(_modulate(self.ln_wo_grad(x), scale2, shift2))
        return x + self.drop_path(branch * gamma2)

    def extra_repr(self) -> str:
        return f'shared_aln={self.shared_aln}'


class AdaLNBeforeHead(nn.Module):
    """Final modulated LayerNorm in front of the classifier head (reference basic_var.py:166-174)"""

    def __init__(self, C, D, norm_layer):
        super().__init__()
        self.C, self.D = C, D
        self.ln_wo_grad = norm_layer(C, elementwise_affine=False)
        self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(D, 2 * C))

    def forward(self, x_BLC, cond_BD):
        scale, shift = self.ada_lin(cond_BD).view(-1, 1, 2, self.C).unbind(2)
        return _modulate(self.ln_wo_grad(x_BLC), scale, shift)
