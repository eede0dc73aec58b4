"""Conv encoder/decoder of the VQVAE as parameter containers with the reference's state-dict layout
(models/basic_vae.py:99-226: LDM-style, ch_mult (1,1,2,2,4), 2 res blocks per level, attention at the lowest resolution).

On the GPU sampling path VQVAE.fhat_to_img bypasses these `forward`s and runs var_amd.engine.DecoderEngine on the same
parameters; the PyTorch `forward`s serve the encode side and autograd users (trainer.py, fork scripts)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['Encoder', 'Decoder']


def Normalize(in_channels, num_groups=32):
    """GroupNorm(32 groups, eps 1e-6, affine) — reference basic_vae.py:18-19"""
    return nn.GroupNorm(num_groups, in_channels, eps=1e-6, affine=True)


def _conv3(cin: int, cout: int, stride: int = 1, pad: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=pad)


def _act_norm(norm: nn.GroupNorm, x: torch.Tensor) -> torch.Tensor:
    return F.silu(norm(x))


class Upsample2x(nn.Module):
    """nearest-neighbour x2, then a 3x3 conv (parameter `conv`)"""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = _conv3(in_channels, in_channels)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='nearest'))


class Downsample2x(nn.Module):
    """zero row/column appended at the bottom/right, then a stride-2 3x3 conv without padding (parameter `conv`)"""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = _conv3(in_channels, in_channels, stride=2, pad=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class ResnetBlock(nn.Module):
    """norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> (dropout) -> conv2, plus the input (through the 1x1 `nin_shortcut` when the width changes)"""

    def __init__(self, *, in_channels, out_channels=None, dropout):
        super().__init__()
        width_out = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, width_out
        self.norm1 = Normalize(in_channels)
        self.conv1 = _conv3(in_channels, width_out)
        self.norm2 = Normalize(width_out)
        self.dropout = nn.Identity() if dropout <= 1e-6 else nn.Dropout(dropout)
        self.conv2 = _conv3(width_out, width_out)
        self.nin_shortcut = nn.Identity() if in_channels == width_out else nn.Conv2d(in_channels, width_out, kernel_size=1)

    def forward(self, x):
        y = self.conv1(_act_norm(self.norm1, x))
        y = self.conv2(self.dropout(_act_norm(self.norm2, y)))
        return self.nin_shortcut(x) + y


class AttnBlock(nn.Module):
    """single-head self-attention over the H*W positions (parameters norm, qkv (1x1 conv to 3C), proj_out (1x1))"""

    def __init__(self, in_channels):
        super().__init__()
        self.C = in_channels
        self.norm = Normalize(in_channels)
        self.qkv = nn.Conv2d(in_channels, 3 * in_channels, kernel_size=1)
        self.w_ratio = int(in_channels) ** (-0.5)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        B, C, H, W = x.shape
        n = H * W
        q, k, v = self.qkv(self.norm(x)).reshape(B, 3, C, n).unbind(1)              # each (B, C, n)
        weights = torch.softmax(torch.bmm(q.transpose(1, 2), k) * self.w_ratio, dim=2)   # (B, n queries, n keys)
        mixed = torch.bmm(v, weights.transpose(1, 2)).view(B, C, H, W)
        return x + self.proj_out(mixed)


def make_attn(in_channels, using_sa=True):
    return AttnBlock(in_channels) if using_sa else nn.Identity()


class _Level(nn.Module):
    """attribute bag for one resolution level (`block`, `attn`, `upsample`/`downsample`) or the middle stage"""


def _middle(width: int, dropout: float, with_attention: bool) -> _Level:
    mid = _Level()
    mid.block_1 = ResnetBlock(in_channels=width, out_channels=width, dropout=dropout)
    mid.attn_1 = make_attn(width, using_sa=with_attention)
    mid.block_2 = ResnetBlock(in_channels=width, out_channels=width, dropout=dropout)
    return mid


def _res_stack(cin: int, cout: int, count: int, dropout: float, with_attention: bool) -> _Level:
    stage = _Level()
    stage.block = nn.ModuleList(ResnetBlock(in_channels=(cin if i == 0 else cout), out_channels=cout, dropout=dropout) for i in range(count))
    stage.attn = nn.ModuleList(make_attn(cout) for _ in range(count)) if with_attention else nn.ModuleList()
    return stage


def _run_stage(stage: _Level, h: torch.Tensor) -> torch.Tensor:
    for i, blk in enumerate(stage.block):
        h = blk(h)
        if len(stage.attn):
            h = stage.attn[i](h)
    return h


class Encoder(nn.Module):
    def __init__(self, *, ch=128, ch_mult=(1, 2, 4, 8), num_res_blocks=2, dropout=0.0, in_channels=3, z_channels, double_z=False,
                 using_sa=True, using_mid_sa=True):
        super().__init__()
        self.ch, self.num_res_blocks, self.in_channels = ch, num_res_blocks, in_channels
        self.num_resolutions = len(ch_mult)
        self.downsample_ratio = 2 ** (self.num_resolutions - 1)
        self.conv_in = _conv3(in_channels, ch)
        widths = [ch] + [ch * m for m in ch_mult]
        self.down = nn.ModuleList()
        for lev in range(self.num_resolutions):
            is_last = lev == self.num_resolutions - 1
            stage = _res_stack(widths[lev], widths[lev + 1], num_res_blocks, dropout, is_last and using_sa)
            if not is_last:
                stage.downsample = Downsample2x(widths[lev + 1])
            self.down.append(stage)
        self.mid = _middle(widths[-1], dropout, using_mid_sa)
        self.norm_out = Normalize(widths[-1])
        self.conv_out = _conv3(widths[-1], (2 if double_z else 1) * z_channels)

    def forward(self, x):
        h = self.conv_in(x)
        for lev, stage in enumerate(self.down):
            h = _run_stage(stage, h)
            if lev + 1 < self.num_resolutions:
                h = stage.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(_act_norm(self.norm_out, h))


class Decoder(nn.Module):
    def __init__(self, *, ch=128, ch_mult=(1, 2, 4, 8), num_res_blocks=2, dropout=0.0, in_channels=3, z_channels, using_sa=True, using_mid_sa=True):
        super().__init__()
        self.ch, self.num_res_blocks, self.in_channels = ch, num_res_blocks, in_channels
        self.num_resolutions = len(ch_mult)
        top = ch * ch_mult[-1]
        self.conv_in = _conv3(z_channels, top)
        self.mid = _middle(top, dropout, using_mid_sa)
        stages, cin = [], top
        for lev in range(self.num_resolutions - 1, -1, -1):                 # built from the lowest resolution up
            cout = ch * ch_mult[lev]
            stage = _res_stack(cin, cout, num_res_blocks + 1, dropout, lev == self.num_resolutions - 1 and using_sa)
            if lev > 0:
                stage.upsample = Upsample2x(cout)
            stages.append(stage)
            cin = cout
        self.up = nn.ModuleList(stages[::-1])               # index = resolution level, 0 = full resolution (as in the reference)
        self.norm_out = Normalize(cin)
        self.conv_out = _conv3(cin, in_channels)

    def forward(self, z):
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(self.conv_in(z))))
        for lev in range(self.num_resolutions - 1, -1, -1):
            h = _run_stage(self.up[lev], h)
            if lev > 0:
                h = self.up[lev].upsample(h)
        return self.conv_out(_act_norm(self.norm_out, h))
