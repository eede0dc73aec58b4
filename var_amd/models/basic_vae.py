"""Conv encoder/decoder of the VQVAE as parameter containers with the reference's state-dict layout
(models/basic_vae.py:99-226: LDM-style, ch_mult (1,1,2,2,4), 2 res blocks per level, attention at the lowest resolution).

On the GPU sampling path VQVAE.fhat_to_img bypasses these `forward`s and runs var_amd.engine.DecoderEngine on the same
parameters; the PyTorch `forward`s serve the encode side and autograd users (trainer.py, fork scripts)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['Encoder', 'Decoder']


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample2x(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='nearest'))


class Downsample2x(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, dropout):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout) if dropout > 1e-6 else nn.Identity()
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        return self.nin_shortcut(x) + h


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.C = in_channels
        self.norm = Normalize(in_channels)
        self.qkv = nn.Conv2d(in_channels, 3 * in_channels, 1)
        self.w_ratio = int(in_channels) ** (-0.5)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        B, C, H, W = x.shape
        q, k, v = self.qkv(self.norm(x)).reshape(B, 3, C, H * W).unbind(1)        # each B, C, HW
        w = torch.softmax(torch.bmm(q.transpose(1, 2), k) * self.w_ratio, dim=2)   # B, HW(q), HW(k)
        h = torch.bmm(v, w.transpose(1, 2)).view(B, C, H, W)
        return x + self.proj_out(h)


def make_attn(in_channels, using_sa=True):
    return AttnBlock(in_channels) if using_sa else nn.Identity()


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    def __init__(self, *, ch=128, ch_mult=(1, 2, 4, 8), num_res_blocks=2, dropout=0.0, in_channels=3, z_channels, double_z=False,
                 using_sa=True, using_mid_sa=True):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks, self.in_channels = ch, len(ch_mult), num_res_blocks, in_channels
        self.downsample_ratio = 2 ** (self.num_resolutions - 1)
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        widths = [ch] + [ch * m for m in ch_mult]
        self.down = nn.ModuleList()
        for lev in range(self.num_resolutions):
            cin, cout, last = widths[lev], widths[lev + 1], lev == self.num_resolutions - 1
            stage = _Level()
            stage.block = nn.ModuleList(ResnetBlock(in_channels=cin if i == 0 else cout, out_channels=cout, dropout=dropout) for i in range(num_res_blocks))
            stage.attn = nn.ModuleList(make_attn(cout) for _ in range(num_res_blocks)) if (last and using_sa) else nn.ModuleList()
            if not last:
                stage.downsample = Downsample2x(cout)
            self.down.append(stage)
        top = widths[-1]
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=top, out_channels=top, dropout=dropout)
        self.mid.attn_1 = make_attn(top, using_sa=using_mid_sa)
        self.mid.block_2 = ResnetBlock(in_channels=top, out_channels=top, dropout=dropout)
        self.norm_out = Normalize(top)
        self.conv_out = nn.Conv2d(top, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for lev, stage in enumerate(self.down):
            for i, blk in enumerate(stage.block):
                h = blk(h)
                if len(stage.attn): h = stage.attn[i](h)
            if lev != self.num_resolutions - 1:
                h = stage.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(F.silu(self.norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, *, ch=128, ch_mult=(1, 2, 4, 8), num_res_blocks=2, dropout=0.0, in_channels=3, z_channels, using_sa=True, using_mid_sa=True):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks, self.in_channels = ch, len(ch_mult), num_res_blocks, in_channels
        top = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, top, 3, 1, 1)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=top, out_channels=top, dropout=dropout)
        self.mid.attn_1 = make_attn(top, using_sa=using_mid_sa)
        self.mid.block_2 = ResnetBlock(in_channels=top, out_channels=top, dropout=dropout)
        stages, cin = [], top
        for lev in reversed(range(self.num_resolutions)):
            cout, lowest = ch * ch_mult[lev], lev == self.num_resolutions - 1
            stage = _Level()
            stage.block = nn.ModuleList(ResnetBlock(in_channels=cin if i == 0 else cout, out_channels=cout, dropout=dropout) for i in range(num_res_blocks + 1))
            stage.attn = nn.ModuleList(make_attn(cout) for _ in range(num_res_blocks + 1)) if (lowest and using_sa) else nn.ModuleList()
            if lev != 0:
                stage.upsample = Upsample2x(cout)
            stages.append(stage); cin = cout
        self.up = nn.ModuleList(reversed(stages))           # index = resolution level, 0 = full resolution (as in the reference)
        self.norm_out = Normalize(cin)
        self.conv_out = nn.Conv2d(cin, in_channels, 3, 1, 1)

    def forward(self, z):
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(self.conv_in(z))))
        for lev in reversed(range(self.num_resolutions)):
            stage = self.up[lev]
            for i, blk in enumerate(stage.block):
                h = blk(h)
                if len(stage.attn): h = stage.attn[i](h)
            if lev != 0:
                h = stage.upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))
