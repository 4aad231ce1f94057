"""Counterpart of the reference's vit_models/transformer_block.py: plain pre-norm Block used by T2T-ViT (qkv_bias False by
default, nn.LayerNorm eps 1e-5) and the frozen sinusoid position table.  Block runs as one fused Function (d2s BlockFn)."""
import numpy as np
import torch
import torch.nn as nn

from d2s import functional as DF


class Mlp(nn.Module):
    """transformer_block.py:14-30."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        assert drop == 0.
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        shape = x.shape
        h = DF.LinearFn.apply(x.reshape(-1, shape[-1]), self.fc1.weight, self.fc1.bias, "gelu")
        return DF.LinearFn.apply(h, self.fc2.weight, self.fc2.bias, None).reshape(*shape[:-1], -1)


class Attention(nn.Module):
    """transformer_block.py:32-57."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert dim // num_heads == 64 and attn_drop == 0. and proj_drop == 0.
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = DF.LinearFn.apply(x.reshape(B * N, C), self.qkv.weight, self.qkv.bias, None)
        o, _ = DF.AttnCoreFn.apply(qkv, B, N, self.num_heads, self.scale, False)
        return DF.LinearFn.apply(o, self.proj.weight, self.proj.bias, None).reshape(B, N, C)


class Block(nn.Module):
    """transformer_block.py:59-75."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop_path == 0. and drop == 0. and qk_scale is None
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, return_cls_attn=False):
        a, m = self.attn, self.mlp
        y, cls_row = DF.run(DF.BlockFn, x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                      self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                      a.num_heads, self.norm1.eps, bool(return_cls_attn), a.scale)
        return (y, cls_row) if return_cls_attn else y


def get_sinusoid_encoding(n_position, d_hid):
    """transformer_block.py:78-88 (host side, float64 table -> FloatTensor, frozen parameter)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0)
