"""Counterpart of the reference's vit_models/t2t_vit.py (T2T_module, T2T_ViT, T2t_vit_14 / T2t_vit_t_14 factories) plus the
BUILD-DEFINED pruned composition of BASELINE config 4: the reference's T2T-ViT has no pruning hooks at all (SURVEY section 2
row 6), so `T2T_ViT_DiffPruning` / `T2T_ViT_Teacher` put the reference's own predictor + top-k + gather (dynamic_vit.py:852-912)
in front of chosen backbone blocks, exactly as VisionTransformerDiffPruning does, and return the same tuples.

Line references are to /root/reference/vit_models/t2t_vit.py unless stated otherwise."""
import math

import torch
import torch.nn as nn

from d2s import functional as DF
from d2s import functional_t2t as TF
from .dynamic_vit import PredictorLG, trunc_normal_
from .token_performer import Token_performer
from .token_transformer import Token_transformer
from .transformer_block import Block, get_sinusoid_encoding


from .dynamic_vit import _scores

class _Unfold(nn.Module):
    """nn.Unfold(kernel_size, stride, padding) followed by .transpose(1, 2) (:85,92,99), reading image or token layout."""

    def __init__(self, kernel_size, stride, padding):
        super().__init__()
        self.k, self.s, self.p = kernel_size[0], stride[0], padding[0]

    def forward(self, x, hw=None):
        return TF.UnfoldFn.apply(x, self.k, self.s, self.p, hw)


class T2T_module(nn.Module):
    """:45-104."""

    def __init__(self, img_size=224, tokens_type='performer', in_chans=3, embed_dim=768, token_dim=64):
        super().__init__()
        if tokens_type not in ('performer', 'transformer'):
            raise NotImplementedError("tokens_type 'convolution' is a comparison baseline of the reference, not on the path")
        self.soft_split0 = _Unfold((7, 7), (4, 4), (2, 2))
        self.soft_split1 = _Unfold((3, 3), (2, 2), (1, 1))
        self.soft_split2 = _Unfold((3, 3), (2, 2), (1, 1))
        if tokens_type == 'transformer':
            self.attention1 = Token_transformer(dim=in_chans * 7 * 7, in_dim=token_dim, num_heads=1, mlp_ratio=1.0)
            self.attention2 = Token_transformer(dim=token_dim * 3 * 3, in_dim=token_dim, num_heads=1, mlp_ratio=1.0)
        else:
            self.attention1 = Token_performer(dim=in_chans * 7 * 7, in_dim=token_dim, kernel_ratio=0.5)
            self.attention2 = Token_performer(dim=token_dim * 3 * 3, in_dim=token_dim, kernel_ratio=0.5)
        self.project = nn.Linear(token_dim * 3 * 3, embed_dim)
        self.num_patches = (img_size // (4 * 2 * 2)) * (img_size // (4 * 2 * 2))

    def forward(self, x):
        x = self.soft_split0(x)                                   # [B, (H/4)^2, 147]
        x = self.attention1(x)                                    # [B, (H/4)^2, 64]
        x = self.soft_split1(x, hw=int(math.isqrt(x.shape[1])))   # tokens read as an image in place (:90-92)
        x = self.attention2(x)
        x = self.soft_split2(x, hw=int(math.isqrt(x.shape[1])))
        B, L, F_ = x.shape
        return DF.LinearFn.apply(x.reshape(B * L, F_), self.project.weight, self.project.bias, None).reshape(B, L, -1)


class T2T_ViT(nn.Module):
    """:106-179."""

    def __init__(self, img_size=224, tokens_type='performer', in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, token_dim=64):
        super().__init__()
        assert drop_rate == 0. and attn_drop_rate == 0. and drop_path_rate == 0.
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.tokens_to_token = T2T_module(img_size=img_size, tokens_type=tokens_type, in_chans=in_chans, embed_dim=embed_dim,
                                          token_dim=token_dim)
        num_patches = self.tokens_to_token.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(data=get_sinusoid_encoding(n_position=num_patches + 1, d_hid=embed_dim), requires_grad=False)
        self.pos_drop = nn.Identity()
        self.blocks = nn.ModuleList([Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                           norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'cls_token'}

    def get_classifier(self):
        return self.head

    def _tokens(self, x):
        """tokens_to_token, prepend CLS, add the frozen sinusoid table (:158-163)."""
        t = self.tokens_to_token(x)
        return DF.AddClsPosFn.apply(t, self.cls_token, self.pos_embed)

    def forward_features(self, x):
        x = self._tokens(x)
        block_heads = []
        for blk in self.blocks:
            x = blk(x)
            block_heads.append(TF.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps))
        return block_heads[-1][:, 0], block_heads

    def forward(self, x, get_average=False):
        x, block_heads = self.forward_features(x)
        if get_average:
            outs = [DF.LinearFn.apply(h[:, 0].contiguous(), self.head.weight, self.head.bias, None) for h in block_heads]
            return torch.mean(torch.stack(outs, 0), dim=0)
        return DF.LinearFn.apply(x.contiguous(), self.head.weight, self.head.bias, None)


class T2T_ViT_Teacher(T2T_ViT):
    """Dense T2T-ViT that also returns what the distillation losses need: (logits, tokens[B,N,D], cls_attn[B,depth,H,N+1])."""

    def forward(self, x):
        x = self._tokens(x)
        rows = []
        for blk in self.blocks:
            x, cls_row = blk(x, return_cls_attn=True)
            rows.append(cls_row.detach())
        logits, tokens = DF.run(DF.HeadFn, x, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, self.norm.eps)
        return logits, tokens, torch.stack(rows, dim=1)


class T2T_ViT_DiffPruning(T2T_ViT):
    """T2T-ViT backbone with the dense-to-sparse pruning stages of dynamic_vit.py:847-924 in front of the blocks listed in
    pruning_loc.  Same outputs / attributes as VisionTransformerDiffPruning."""

    def __init__(self, pruning_loc=None, token_ratio=None, topk_selection=True, predictor_loss_type="kl_div", init_n=14 * 14, **kwargs):
        super().__init__(**kwargs)
        self.pruning_loc, self.token_ratio, self.init_n = list(pruning_loc or []), list(token_ratio or []), init_n
        self.score_predictor = nn.ModuleList([PredictorLG(self.embed_dim, topk_selection=topk_selection, k=int(r * init_n),
                                                          loss_type=predictor_loss_type) for r in self.token_ratio])
        self.score_predictor.apply(self._init_weights)
        self.kept_token_indices, self.dropped_token_indices, self.pred_logits, self.cls_attns = None, None, [], []
        self.grad_ready_hook = None

    def forward(self, x):
        x = self._tokens(x)
        self.cls_attns, self.pred_logits, self.kept_token_indices, self.dropped_token_indices = [], [], [], []
        p = 0
        for i, blk in enumerate(self.blocks):
            if self.grad_ready_hook is not None and x.requires_grad:
                x.register_hook(lambda g, i=i, cb=self.grad_ready_hook: (cb(i), None)[1])
            if i in self.pruning_loc:
                pred_logits, pred_score = _scores(self.score_predictor[p], x)
                kept, dropped = DF.select_topk(pred_score, int(self.init_n * self.token_ratio[p]))
                self.kept_token_indices.append(kept)
                self.dropped_token_indices.append(dropped)
                self.pred_logits.append(pred_logits)
                x = DF.GatherFn.apply(x, kept)
                p += 1
            x, cls_row = blk(x, return_cls_attn=True)
            self.cls_attns.append(cls_row[:, :, 1:])
        logits, features = DF.run(DF.HeadFn, x, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, self.norm.eps)
        if self.training:
            return logits, features, self.pred_logits, self.kept_token_indices
        return logits, self.cls_attns, self.pred_logits, self.kept_token_indices


def _load_local(model, checkpoint_path):
    if checkpoint_path is not None:     # the reference reads pretrained_models/*.pth.tar["state_dict_ema"] (:186-275); local files only
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        model.load_state_dict(sd.get("state_dict_ema", sd))
    return model


def T2t_vit_14(pretrained=False, checkpoint_path=None, **kwargs):
    """:211-221 - performer token encoder, D 384, depth 14, 6 heads, mlp_ratio 3."""
    return _load_local(T2T_ViT(tokens_type='performer', embed_dim=384, depth=14, num_heads=6, mlp_ratio=3., **kwargs), checkpoint_path)


def T2t_vit_t_14(pretrained=False, checkpoint_path=None, **kwargs):
    """:249-258 - transformer token encoder."""
    return _load_local(T2T_ViT(tokens_type='transformer', embed_dim=384, depth=14, num_heads=6, mlp_ratio=3., **kwargs), checkpoint_path)


def t2t_vit_14_student(pruning_locs, keep_ratios, **kwargs):
    """BASELINE config 4 (build-defined): T2T-ViT-14 with the predictor + gather in front of the listed backbone blocks."""
    return T2T_ViT_DiffPruning(pruning_loc=pruning_locs, token_ratio=keep_ratios, tokens_type='performer', embed_dim=384, depth=14,
                               num_heads=6, mlp_ratio=3., **kwargs)


def t2t_vit_14_teacher(**kwargs):
    return T2T_ViT_Teacher(tokens_type='performer', embed_dim=384, depth=14, num_heads=6, mlp_ratio=3., **kwargs)
