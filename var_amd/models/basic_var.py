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


class FFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0., fused_if_available=True):
        super().__init__()
        self.fused_mlp_func = None                      # the reference's optional flash-attn fused MLP; never present here
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU(approximate='tanh')
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop, inplace=True) if drop > 0 else nn.Identity()

    def forward(self, x):
        return self.drop(self.fc2(self.act(self.fc1(x))))


class SelfAttention(nn.Module):
    def __init__(self, block_idx, embed_dim=768, num_heads=12, attn_drop=0., proj_drop=0., attn_l2_norm=False, flash_if_available=True):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.block_idx, self.num_heads, self.head_dim = block_idx, num_heads, embed_dim // num_heads
        self.attn_l2_norm = attn_l2_norm
        if attn_l2_norm:
            self.scale = 1
            self.scale_mul_1H11 = nn.Parameter(torch.full((1, num_heads, 1, 1), 4.0).log(), requires_grad=True)
            self.max_scale_mul = torch.log(torch.tensor(100)).item()
        else:
            self.scale = 0.25 / math.sqrt(self.head_dim)
        self.mat_qkv = nn.Linear(embed_dim, embed_dim * 3, bias=False)
        self.q_bias, self.v_bias = nn.Parameter(torch.zeros(embed_dim)), nn.Parameter(torch.zeros(embed_dim))
        self.register_buffer('zero_k_bias', torch.zeros(embed_dim))
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(proj_drop, inplace=True) if proj_drop > 0 else nn.Identity()
        self.attn_drop = attn_drop
        self.using_flash = self.using_xform = False     # third-party kernels of the reference: not used by this build
        self.caching, self.cached_k, self.cached_v = False, None, None

    def kv_caching(self, enable: bool):
        """API kept for callers that toggle it (notebooks); the HIP path owns a pre-allocated cache in the engine."""
        self.caching, self.cached_k, self.cached_v = enable, None, None

    def forward(self, x, attn_bias):
        B, L, C = x.shape
        qkv = F.linear(x, self.mat_qkv.weight, torch.cat((self.q_bias, self.zero_k_bias, self.v_bias))).view(B, L, 3, self.num_heads, self.head_dim)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)                      # B H L c
        if self.attn_l2_norm:
            q = F.normalize(q, dim=-1) * self.scale_mul_1H11.clamp_max(self.max_scale_mul).exp()
            k = F.normalize(k, dim=-1)
        if self.caching:
            if self.cached_k is not None:
                k, v = torch.cat((self.cached_k, k), dim=2), torch.cat((self.cached_v, v), dim=2)
            self.cached_k, self.cached_v = k, v
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_bias, dropout_p=self.attn_drop if self.training else 0.0, scale=self.scale)
        return self.proj_drop(self.proj(o.transpose(1, 2).reshape(B, L, C)))

    def extra_repr(self) -> str:
        return f'attn_l2_norm={self.attn_l2_norm}'


class AdaLNSelfAttn(nn.Module):
    def __init__(self, block_idx, last_drop_p, embed_dim, cond_dim, shared_aln: bool, norm_layer, num_heads, mlp_ratio=4., drop=0.,
                 attn_drop=0., drop_path=0., attn_l2_norm=False, flash_if_available=False, fused_if_available=True):
        super().__init__()
        self.block_idx, self.last_drop_p, self.C, self.D = block_idx, last_drop_p, embed_dim, cond_dim
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.attn = SelfAttention(block_idx=block_idx, embed_dim=embed_dim, num_heads=num_heads, attn_drop=attn_drop, proj_drop=drop,
                                  attn_l2_norm=attn_l2_norm, flash_if_available=flash_if_available)
        self.ffn = FFN(in_features=embed_dim, hidden_features=round(embed_dim * mlp_ratio), drop=drop, fused_if_available=fused_if_available)
        self.ln_wo_grad = norm_layer(embed_dim, elementwise_affine=False)
        self.shared_aln = shared_aln
        if shared_aln:
            self.ada_gss = nn.Parameter(torch.randn(1, 1, 6, embed_dim) / embed_dim ** 0.5)
        else:
            self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(cond_dim, 6 * embed_dim))
        self.fused_add_norm_fn = None

    def forward(self, x, cond_BD, attn_bias):
        mod = (self.ada_gss + cond_BD) if self.shared_aln else self.ada_lin(cond_BD).view(-1, 1, 6, self.C)
        gamma1, gamma2, scale1, scale2, shift1, shift2 = mod.unbind(2)
        x = x + self.drop_path(self.attn(self.ln_wo_grad(x) * (scale1 + 1) + shift1, attn_bias=attn_bias) * gamma1)
        x = x + self.drop_path(self.ffn(self.ln_wo_grad(x) * (scale2 + 1) + shift2) * gamma2)
        return x

    def extra_repr(self) -> str:
        return f'shared_aln={self.shared_aln}'


class AdaLNBeforeHead(nn.Module):
    def __init__(self, C, D, norm_layer):
        super().__init__()
        self.C, self.D = C, D
        self.ln_wo_grad = norm_layer(C, elementwise_affine=False)
        self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(D, 2 * C))

    def forward(self, x_BLC, cond_BD):
        scale, shift = self.ada_lin(cond_BD).view(-1, 1, 2, self.C).unbind(2)
        return self.ln_wo_grad(x_BLC) * (scale + 1) + shift
