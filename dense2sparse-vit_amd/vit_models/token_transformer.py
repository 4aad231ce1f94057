"""Counterpart of the reference's vit_models/token_transformer.py (softmax-attention token encoder of T2T-ViT_t)."""
import torch.nn as nn

from d2s import functional_t2t as TF
from .transformer_block import Mlp


class Attention(nn.Module):
    """token_transformer.py:12-43 - parameter container (qkv to 3*in_dim, skip through v); runs inside TokenTransformerFn."""

    def __init__(self, dim, num_heads=8, in_dim=None, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert num_heads == 1 and in_dim == 64 and qk_scale is None and attn_drop == 0. and proj_drop == 0.
        self.num_heads, self.in_dim = num_heads, in_dim
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, in_dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(in_dim, in_dim)


class Token_transformer(nn.Module):
    def __init__(self, dim, in_dim, num_heads, mlp_ratio=1., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop == 0. and drop_path == 0.
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, in_dim=in_dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(in_dim)
        self.mlp = Mlp(in_features=in_dim, hidden_features=int(in_dim * mlp_ratio), out_features=in_dim, act_layer=act_layer, drop=drop)

    def forward(self, x):
        a, m = self.attn, self.mlp
        return TF.TokenTransformerFn.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                           self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
