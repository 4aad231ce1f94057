"""Counterpart of the reference's vit_models/token_performer.py (FAVOR+ token encoder of T2T).  The two dropouts the
reference hard-wires at p = 0.1 (token_performer.py:9,13,24) are the identity here: the accelerated path defines parity for
p = 0 / eval mode."""
import math

import torch
import torch.nn as nn

from d2s import functional_t2t as TF


class Token_performer(nn.Module):
    def __init__(self, dim, in_dim, head_cnt=1, kernel_ratio=0.5, dp1=0.1, dp2=0.1):
        super().__init__()
        assert head_cnt == 1 and in_dim == 64 and kernel_ratio == 0.5, "HIP performer kernels: emb 64, m 32, one head"
        self.emb = in_dim * head_cnt
        self.kqv = nn.Linear(dim, 3 * self.emb)
        self.dp = nn.Identity()
        self.proj = nn.Linear(self.emb, self.emb)
        self.head_cnt = head_cnt
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(self.emb)
        self.epsilon = 1e-8
        self.mlp = nn.Sequential(nn.Linear(self.emb, 1 * self.emb), nn.GELU(), nn.Linear(1 * self.emb, self.emb), nn.Identity())
        self.m = int(self.emb * kernel_ratio)
        w = torch.randn(self.m, self.emb)
        self.w = nn.Parameter(nn.init.orthogonal_(w) * math.sqrt(self.m), requires_grad=False)   # token_performer.py:28-29

    def forward(self, x):
        return TF.TokenPerformerFn.apply(x, self.norm1.weight, self.norm1.bias, self.kqv.weight, self.kqv.bias, self.proj.weight,
                                         self.proj.bias, self.norm2.weight, self.norm2.bias, self.mlp[0].weight, self.mlp[0].bias,
                                         self.mlp[2].weight, self.mlp[2].bias, self.w)
