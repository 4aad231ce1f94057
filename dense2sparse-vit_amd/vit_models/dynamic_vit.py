"""MI355X-native counterpart of the reference's vit_models/dynamic_vit.py.

Same module-level names, constructor signatures, state-dict keys, forward return tuples and post-forward attributes as
the reference, so train.py / evaluate.py style callers run unchanged - but every arithmetic
operation is a hand-written HIP kernel reached through the C ABI (d2s.functional / d2s.ops).  nn.Linear / nn.LayerNorm /
nn.Conv2d objects are kept purely as parameter containers (they give the reference's state-dict key names); their own
forward is never called.  There is no CPU path: tensors must live on an MI355X and the library must be built.

Line references are to /root/reference/vit_models/dynamic_vit.py.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from d2s import functional as DF
from .peturbed_topk import PerturbedTopK


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


def batch_index_select(x, idx):
    """:39-60 - gather rows per batch element (kept for API parity; the model itself uses GatherFn)."""
    if x.dim() == 3:
        B, N, C = x.shape
        off = torch.arange(B, dtype=torch.long, device=x.device).view(B, 1) * N
        return x.reshape(B * N, C)[(idx + off).reshape(-1)].reshape(B, idx.size(1), C)
    if x.dim() == 2:
        B, N = x.shape
        off = torch.arange(B, dtype=torch.long, device=x.device).view(B, 1) * N
        return x.reshape(B * N)[(idx + off).reshape(-1)].reshape(B, idx.size(1))
    raise NotImplementedError


def patch_drop_mask(kept_token_indices, num_patches):
    """The kept / dropped mask reconstruction of the reference's visual outputs (visualizations.py:18-26: scatter ones for the kept
    ids, zeros for the dropped ones) for every stage, in the coordinates of the ORIGINAL patch grid: stage i's ids are relative to the
    tokens that survived stage i-1, so they are first composed with the previous stages' ids.
    kept_token_indices: list of int64 [B, k_i] (model.kept_token_indices) -> list of int64 [B, num_patches] 0/1 masks (1 = kept)."""
    from d2s import ops
    masks, absolute = [], None
    for ids in kept_token_indices:
        ids = ids.contiguous()
        absolute = ids if absolute is None else ops.compose_ids(absolute, ids)
        masks.append(ops.patch_keep_mask(absolute, num_patches))
    return masks



def _scores(predictor, x):
    """predictor.forward_tokens(x); the reference's predictor falls through and returns None without topk_selection (:537) and the
    caller then fails on the unpacking - same outcome here, with a message that says what to pass."""
    out = predictor.forward_tokens(x)
    if out is None:
        raise RuntimeError("the score predictor was built with topk_selection=False: the reference's PredictorLG.forward returns None on that "
                           "path (vit_models/dynamic_vit.py:537), so the pruning stages cannot run - pass --topk-selection / topk_selection=True")
    return out

class Mlp(nn.Module):
    """:159-175."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        assert drop == 0., "dropout is identity at the rates the reference uses (p=0)"
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        shape = x.shape
        h = DF.LinearFn.apply(x.reshape(-1, shape[-1]), self.fc1.weight, self.fc1.bias, "gelu")
        return DF.LinearFn.apply(h, self.fc2.weight, self.fc2.bias, None).reshape(*shape[:-1], -1)


class Attention(nn.Module):
    """:179-236.  policy / softmax_with_policy (:195-214) belongs to the patch_score_threshold path."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == 64, "the HIP attention kernels are specialised for head_dim 64"
        assert attn_drop == 0. and proj_drop == 0.
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def softmax_with_policy(self, attn, policy, eps=1e-6):
        """:195-214 on a materialised score tensor (stand-alone operator; Block / Attention.forward use the fused form)."""
        return DF.PolicySoftmaxFn.apply(attn, policy, eps)

    def forward(self, x, policy=None, return_cls_attn=False):
        B, N, C = x.shape
        qkv = DF.LinearFn.apply(x.reshape(B * N, C), self.qkv.weight, self.qkv.bias, None)
        if policy is not None:      # :229 softmax_with_policy, fused into the attention pass
            o, cls_row = DF.AttnCoreFn.apply(qkv, B, N, self.num_heads, self.scale, bool(return_cls_attn), DF.as_policy(policy, B, N))
        else:
            o, cls_row = DF.AttnCoreFn.apply(qkv, B, N, self.num_heads, self.scale, bool(return_cls_attn))
        o = DF.LinearFn.apply(o, self.proj.weight, self.proj.bias, None).reshape(B, N, C)
        return (o, cls_row) if return_cls_attn else o


class Block(nn.Module):
    """:240-283.  The whole block (LN, qkv, fused attention, proj+residual, LN, fc1+GELU, fc2+residual) is one Function."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop_path == 0., "DropPath is identity at rate 0 (:249); stochastic depth is not on the hot path"
        assert qkv_bias, "qkv_bias=True everywhere on the path"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _params(self):
        a, m = self.attn, self.mlp
        return (self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)

    def forward(self, x, policy=None, return_cls_attn=False):
        a = self.attn
        extra = () if policy is None else (DF.as_policy(policy, x.shape[0], x.shape[1]),)     # :263-283 with policy -> fused policy softmax
        y, cls_row = DF.run(DF.BlockFn, x, *self._params(), a.num_heads, self.norm1.eps, bool(return_cls_attn), a.scale, *extra)
        return (y, cls_row) if return_cls_attn else y

    def forward_ragged(self, xp, cu_seqlens, B, max_n, return_cls_attn=False):
        """The block on a ragged packed batch [total, D] (inference with a dynamic keep ratio, :935-949).  Forward only."""
        a = self.attn
        y, cls_rows = DF.ragged_block_forward(xp, cu_seqlens, B, max_n, self._params(), a.num_heads, self.norm1.eps, a.scale,
                                              want_cls=bool(return_cls_attn))
        return (y, cls_rows) if return_cls_attn else y


class PatchEmbed(nn.Module):
    """:286-306."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def check(self, x):
        H, W = x.shape[2], x.shape[3]
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."

    def forward(self, x):
        """Patch tokens without CLS / pos (API parity); the models call EmbedFn which fuses those adds."""
        self.check(x)
        D = self.proj.weight.shape[0]
        T = self.num_patches
        zeros_pos = torch.zeros((1, T + 1, D), dtype=torch.float32, device=x.device)
        zeros_cls = torch.zeros((1, 1, D), dtype=torch.float32, device=x.device)
        return DF.run(DF.EmbedFn, x, self.proj.weight, self.proj.bias, zeros_cls, zeros_pos, self.patch_size[0])[:, 1:]


class BatchNormLayer(nn.Module):
    """:350-367: BatchNorm1d over all B * N token rows of a [B, N, D] tensor (the reference transposes around nn.BatchNorm1d).  Holds
    the parameters / running estimates under the reference's keys (`bn.weight`, `bn.running_mean`, ...); inside the predictor the
    arithmetic runs in d2s.functional_bn.PredictorBNFn, standalone use goes through the same kernels."""

    def __init__(self, input_dim=384):
        super().__init__()
        self.bn = nn.BatchNorm1d(input_dim)

    def forward(self, x):
        from d2s import ops
        B, N, D = x.shape
        if torch.is_grad_enabled() and (x.requires_grad or self.bn.weight.requires_grad):
            raise NotImplementedError("differentiable standalone BatchNormLayer: use it through PredictorLG (predictor_bn=True)")
        y, _, _ = ops.batchnorm_fwd(x.contiguous().view(B * N, D), self.bn.weight, self.bn.bias, self.bn.running_mean,
                                    self.bn.running_var, self.training, eps=self.bn.eps, momentum=self.bn.momentum)
        if self.training:
            self.bn.num_batches_tracked += 1
        return y.view(B, N, D)


class PredictorLG(nn.Module):
    """:370-560.  Large LayerNorm variant (:491-531) runs as one Function; keys follow the nn.Sequential positions."""

    def __init__(self, embed_dim=384, topk_selection=False, k=None, small_predictor=False, loss_type="kl_div", use_bn=False):
        super().__init__()
        self.small_predictor, self.k, self.topk_selection, self.loss_type, self.use_bn = small_predictor, k, topk_selection, loss_type, use_bn
        D = embed_dim
        relu = nn.ReLU()
        if use_bn and small_predictor:      # :383-400: small BatchNorm predictor (ReLU, unlike the small LayerNorm one which uses GELU)
            self.in_conv = nn.Sequential(BatchNormLayer(D), nn.Linear(D, D), relu)
            self.out_conv = nn.Sequential(BatchNormLayer(D), nn.Linear(D, D // 2), relu, BatchNormLayer(D // 2),
                                          nn.Linear(D // 2, D // 4), relu, BatchNormLayer(D // 4), nn.Linear(D // 4, 1),
                                          nn.Flatten(start_dim=-2, end_dim=-1))
            self.topk = PerturbedTopK(k)
            return
        if use_bn:               # :438-476: the large predictor with BatchNormLayer in place of every LayerNorm
            self.in_conv = nn.Sequential(BatchNormLayer(D), nn.Linear(D, D * 4), relu)
            self.out_conv = nn.Sequential(
                BatchNormLayer(D * 4), nn.Linear(D * 4, D * 2), relu,
                BatchNormLayer(D * 2), nn.Linear(D * 2, D), relu,
                BatchNormLayer(D), nn.Linear(D, D // 2), relu,
                BatchNormLayer(D // 2), nn.Linear(D // 2, D // 4), relu,
                BatchNormLayer(D // 4), nn.Linear(D // 4, 1), nn.Flatten(start_dim=-2, end_dim=-1))
            self.topk = PerturbedTopK(k)
            return
        if small_predictor:      # :409-426 (LayerNorm + GELU variant)
            self.in_conv = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, D), nn.GELU())
            self.out_conv = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, D // 2), nn.GELU(), nn.LayerNorm(D // 2),
                                          nn.Linear(D // 2, D // 4), nn.GELU(), nn.LayerNorm(D // 4), nn.Linear(D // 4, 1),
                                          nn.Flatten(start_dim=-2, end_dim=-1))
            self.topk = PerturbedTopK(k)
            return
        self.in_conv = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, D * 4), relu)
        self.out_conv = nn.Sequential(
            nn.LayerNorm(D * 4), nn.Linear(D * 4, D * 2), relu,
            nn.LayerNorm(D * 2), nn.Linear(D * 2, D), relu,
            nn.LayerNorm(D), nn.Linear(D, D // 2), relu,
            nn.LayerNorm(D // 2), nn.Linear(D // 2, D // 4), relu,
            nn.LayerNorm(D // 4), nn.Linear(D // 4, 1), nn.Flatten(start_dim=-2, end_dim=-1))
        self.topk = PerturbedTopK(k)

    def _params(self):
        norm = (lambda m: m.bn) if self.use_bn else (lambda m: m)
        ps = [norm(self.in_conv[0]).weight, norm(self.in_conv[0]).bias, self.in_conv[1].weight, self.in_conv[1].bias]
        for i in ((0, 3, 6) if self.small_predictor else (0, 3, 6, 9, 12)):
            ps += [norm(self.out_conv[i]).weight, norm(self.out_conv[i]).bias, self.out_conv[i + 1].weight, self.out_conv[i + 1].bias]
        return ps

    def _bn_layers(self):
        return [self.in_conv[0].bn] + [self.out_conv[i].bn for i in ((0, 3, 6) if self.small_predictor else (0, 3, 6, 9, 12))]

    def forward_tokens(self, x_with_cls):
        """Scores for x[:, 1:] of a [B, n, D] tensor, read in place (no slice copy).  -> (scores, keep_probs)."""
        if not self.topk_selection:
            return None  # the reference's forward falls through and returns None (:537)
        if self.loss_type not in ("kl_div", "mse"):
            raise NotImplementedError("sigmoid scores (bce loss type) are not on the accelerated hot path")
        if self.use_bn:
            from d2s.functional_bn import PredictorBNFn
            bns = self._bn_layers()
            running = [t for bn in bns for t in (bn.running_mean, bn.running_var)]
            out = DF.run(PredictorBNFn, x_with_cls, self.training, running, *self._params())
            if self.training:
                for bn in bns:
                    bn.num_batches_tracked += 1
            return out
        if self.small_predictor:
            from d2s.functional_small import SmallPredictorFn
            return DF.run(SmallPredictorFn, x_with_cls, *self._params())
        return DF.run(DF.PredictorFn, x_with_cls, *self._params())

    def forward(self, x, policy=None, current_sigma=0.0005, cls_attn=None):
        """Reference signature: x is the CLS-free token tensor [B, N, D] (:855 passes x[:, 1:])."""
        B, N, D = x.shape
        pad = torch.zeros((B, 1, D), dtype=x.dtype, device=x.device)
        return self.forward_tokens(torch.cat([pad, x], dim=1))


class _ViTBase(nn.Module):
    def _build_trunk(self, img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale,
                     representation_size, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone, norm_layer):
        assert hybrid_backbone is None, "HybridEmbed is not on the hot path"
        assert representation_size is None, "pre_logits representation layer is unused by the reference's factories"
        assert drop_rate == 0. and attn_drop_rate == 0. and drop_path_rate == 0., "dropout / drop-path are 0 on the path"
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=0., norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()

    def _init_weights(self, m):
        """:794-801."""
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def _embed(self, x):
        self.patch_embed.check(x)
        return DF.run(DF.EmbedFn, x, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token, self.pos_embed,
                                self.patch_embed.patch_size[0])

    def _head(self, x):
        return DF.run(DF.HeadFn, x, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, self.norm.eps)


class VisionTransformerDiffPruning(_ViTBase):
    """:642-1015 - the student.  Hard top-k by score (argsort path, :858-862), kept-token gather (:907-912)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 pruning_loc=None, token_ratio=None, distill=False, attn_selection=False, attn_selection_threshold=0.0,
                 topk_selection=False, early_exit=False, mean_heads=False, random_drop=False, small_predictor=False,
                 predictor_loss_type=False, predictor_bn=False, patch_score_threshold=None, init_n=14 * 14):
        super().__init__()
        self._build_trunk(img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale,
                          representation_size, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone, norm_layer)
        if early_exit:      # :752-758: the head is created (state-dict keys, 'early_exit' parameter group) but no forward path of the
            # reference ever calls it, so its parameters never receive a gradient and the optimiser never moves them
            self.early_exit_head = nn.Sequential((norm_layer or partial(nn.LayerNorm, eps=1e-6))(embed_dim),
                                                 nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity())
        pruning_loc = list(pruning_loc or [])
        token_ratio = list(token_ratio or [])
        isz = img_size if isinstance(img_size, int) else img_size[0]
        psz = patch_size if isinstance(patch_size, int) else patch_size[0]
        self.score_predictor = nn.ModuleList([
            PredictorLG(embed_dim, topk_selection=topk_selection, k=int(token_ratio[i] * (isz / psz) ** 2),
                        small_predictor=small_predictor, loss_type=predictor_loss_type, use_bn=predictor_bn)
            for i in range(len(pruning_loc))])
        self.num_kept_tokens = []
        self.attn_selection, self.attn_selection_threshold = attn_selection, attn_selection_threshold
        self.topk_selection = topk_selection
        if self.topk_selection:
            self.current_sigma = 0.05
        self.mean_heads, self.random_drop, self.current_score, self.early_exit = mean_heads, random_drop, None, early_exit
        self.cls_attns = []
        self.kept_token_indices = None
        self.dropped_token_indices = None
        self.pred_logits = []
        self.patch_score_threshold = patch_score_threshold
        self.keep_ratios = None          # :886 / :941 - per-image kept fraction of the last thresholded stage (device tensor)
        self.cu_seqlens = self.ragged_row_src = None     # ragged inference: packed-row offsets per image / source token of each row
        # replay of a recorded selection (analysis / tests): a list of int64 [B, k_i] kept-id tensors, one per stage, used INSTEAD of the
        # top-k of this forward's scores (which are still computed and returned); None = normal operation
        self.kept_token_override = None
        self.unpruned = False
        self.distill = distill
        self.pruning_loc, self.token_ratio = pruning_loc, token_ratio
        # the reference hard-codes init_n = 14*14 (:828,852); exposed so that 384x384 inputs can use int(N * ratio) instead
        self.init_n = init_n
        # optional callable(block_index): invoked from autograd when every parameter gradient of block i and of all later
        # layers is final (d2s.engine uses it to start the bucketed gradient all-reduce while backward continues)
        self.grad_ready_hook = None
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    # :887-889 / :942-944 read the minimum / mean / maximum kept fraction with three .item() calls (three device syncs per stage per
    # step); here they are derived from `keep_ratios` only when a caller looks at them
    @property
    def min_keep_ratio(self):
        return None if self.keep_ratios is None else float(self.keep_ratios.min())

    @property
    def avg_keep_ratio(self):
        return None if self.keep_ratios is None else float(self.keep_ratios.mean())

    @property
    def max_keep_ratio(self):
        return None if self.keep_ratios is None else float(self.keep_ratios.max())

    def forward(self, x, stacked_cls_attn_weights=None):
        if self.patch_score_threshold is not None:
            return self._forward_threshold(x)
        x = self._embed(x)                                                  # :816-824
        self.num_kept_tokens, self.cls_attns, self.pred_logits = [], [], []
        self.kept_token_indices, self.dropped_token_indices = [], []
        p_count = 0
        for i, blk in enumerate(self.blocks):
            if self.grad_ready_hook is not None and x.requires_grad:
                x.register_hook(lambda g, i=i, cb=self.grad_ready_hook: (cb(i), None)[1])
            if i in self.pruning_loc:
                num_keep_node = int(self.init_n * self.token_ratio[p_count])   # :852
                pred_logits, pred_score = _scores(self.score_predictor[p_count], x)   # :855
                kept, dropped = DF.select_topk(pred_score, num_keep_node)     # :858-862
                if self.kept_token_override is not None:
                    kept = self.kept_token_override[p_count].to(device=x.device, dtype=torch.int64).contiguous()
                    assert kept.shape == (x.shape[0], num_keep_node), "override ids must be [B, int(init_n * ratio)]"
                    keep_mask = DF.ops.patch_keep_mask(kept, x.shape[1] - 1)
                    dropped = torch.nonzero(keep_mask == 0)[:, 1].reshape(x.shape[0], -1)
                self.kept_token_indices.append(kept)
                self.dropped_token_indices.append(dropped)
                self.pred_logits.append(pred_logits)
                x = DF.GatherFn.apply(x, kept)                               # :907-912 / :954-960
                p_count += 1
            x, cls_attn = blk(x, return_cls_attn=True)                       # :924 / :985
            self.cls_attns.append(cls_attn[:, :, 1:])
        logits, features = self._head(x)                                      # :993-1006
        if self.training:
            return logits, features, self.pred_logits, self.kept_token_indices   # :1013
        return logits, self.cls_attns, self.pred_logits, self.kept_token_indices  # :1015

    def _forward_threshold(self, x):
        """Dynamic keep ratio (patch_score_threshold is set).

        Training (:880-894, :981-983, :1010-1011): no token is removed.  Each pruning stage turns its keep probabilities into a 0/1 mask
        (tokens whose ascending cumulative probability exceeds the threshold are kept), and that stage's block and every later block
        attend through Attention.softmax_with_policy with the mask as key policy; the blocks before the first stage use the all-ones
        policy, exactly as the reference does.  A later stage REPLACES the mask (the reference does not intersect them).
        Returns (logits, features [B,N,D], [pred_logits per stage], [keep mask [B,N] per stage]) - lists, where the reference returns the
        last stage's tensors only (:1011): the per-stage mask loss needs every stage.

        Inference (:935-949, with the reference's undefined `score` read as `pred_score`): the kept tokens of every image are packed
        into one ragged batch [total, D] (`cu_seqlens` [B+1] gives each image's rows) and the remaining blocks run on it - LayerNorm
        and GEMMs over all packed rows, attention per image.  One pruning stage (the reference's second stage scatters an n_kept-long
        mask into an N-long buffer and cannot run).  Returns (logits, [cls rows: dense [B,H,N] before the stage, packed [H,total]
        after it], [pred_logits], [keep mask [B,N]])."""
        from d2s import ops
        thr = float(self.patch_score_threshold)
        x = self._embed(x)
        B, n, D = x.shape
        N = n - 1
        self.num_kept_tokens, self.cls_attns, self.pred_logits = [], [], []
        self.kept_token_indices, self.dropped_token_indices = [], []       # here: per-stage keep masks / their complements
        self.cu_seqlens = self.ragged_row_src = None
        p_count = 0
        if self.training:
            policy = torch.ones((B, n), dtype=torch.float32, device=x.device)       # :830,839
            for i, blk in enumerate(self.blocks):
                if self.grad_ready_hook is not None and x.requires_grad:
                    x.register_hook(lambda g, i=i, cb=self.grad_ready_hook: (cb(i), None)[1])
                if i in self.pruning_loc:
                    pred_logits, pred_score = _scores(self.score_predictor[p_count], x)      # :855
                    policy, counts = ops.select_threshold(pred_score.detach().contiguous(), thr, lead=1)   # :881-893 -> [1, mask]
                    self.keep_ratios = counts.float() / N                                          # :886
                    self.pred_logits.append(pred_logits)
                    self.kept_token_indices.append(policy[:, 1:])
                    self.dropped_token_indices.append(1.0 - policy[:, 1:])
                    p_count += 1
                x = blk(x, policy=policy)                                                          # :894 / :983
            logits, features = self._head(x)
            return logits, features, self.pred_logits, self.kept_token_indices
        if len(self.pruning_loc) > 1:
            raise NotImplementedError("ragged inference with a dynamic keep ratio supports one pruning stage: the reference's second "
                                      "stage scatters an n_kept-long mask into an N-long buffer (dynamic_vit.py:945-946) and cannot run")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("ragged inference is forward only: call it under torch.no_grad()")
        cu = None
        for i, blk in enumerate(self.blocks):
            if i in self.pruning_loc:
                pred_logits, pred_score = _scores(self.score_predictor[p_count], x)
                mask, counts = ops.select_threshold(pred_score.contiguous(), thr)                  # :936-938 (score := pred_score)
                self.keep_ratios = counts.float() / N                                              # :941
                self.pred_logits.append(pred_logits)
                self.kept_token_indices.append(mask)
                self.dropped_token_indices.append(1.0 - mask)
                cu = ops.ragged_offsets(counts, extra=1)
                total = int(cu[-1])          # the one device sync of the ragged path: the packed row count sizes every later launch
                x, self.ragged_row_src = ops.ragged_pack(x, mask, cu, total)                       # :947-948
                self.cu_seqlens = cu
                x = blk.forward_ragged(x, cu, B, n)                                                # :949 blk(x)
                p_count += 1
            elif cu is None:
                x, cls_attn = blk(x, return_cls_attn=True)                                         # :987
                self.cls_attns.append(cls_attn[:, :, 1:])
            else:
                x, cls_rows = blk.forward_ragged(x, cu, B, n, return_cls_attn=True)
                self.cls_attns.append(cls_rows)             # packed [H, total]; image b's CLS row is columns cu[b] .. cu[b+1]
        if cu is None:
            logits, features = self._head(x)
            return logits, self.cls_attns, self.pred_logits, self.kept_token_indices
        total = x.shape[0]
        xn, _, _ = ops.layernorm_fwd(x, ops.contiguous_map(total, D), self.norm.weight, self.norm.bias, total, D, self.norm.eps, stats=False)
        cls_rows = ops.gather_rows_i32(xn, cu, B)                                                  # :996 x[:, 0] of every image
        logits = ops.linear_fwd(cls_rows, self.head.weight, self.head.bias)
        self.ragged_features = xn
        return logits, self.cls_attns, self.pred_logits, self.kept_token_indices

    def forward_cls_attn(self, x):
        """:1018-1033."""
        x = self._embed(x)
        final = None
        for i, blk in enumerate(self.blocks):
            if i == len(self.blocks) - 1:
                _, final = blk(x, return_cls_attn=True)
            else:
                x = blk(x)
        return final


class VisionTransformerTeacher(_ViTBase):
    """:1036-1176 - dense frozen teacher; every block returns its (detached) CLS row."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None):
        super().__init__()
        self._build_trunk(img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale,
                          representation_size, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone, norm_layer)
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def forward_cls_attention(self, x):
        """:1134-1148."""
        x = self._embed(x)
        rows = []
        for blk in self.blocks:
            x, cls_attns = blk(x, return_cls_attn=True)
            rows.append(cls_attns.detach())
        return torch.stack(rows, dim=1)

    def forward(self, x):
        """:1150-1176 -> (logits, tokens[B,N,D], cls_attn[B,depth,H,N+1])."""
        x = self._embed(x)
        rows = []
        for blk in self.blocks:
            x, cls_attns = blk(x, return_cls_attn=True)
            rows.append(cls_attns.detach())
        logits, tokens = self._head(x)
        return logits, tokens, torch.stack(rows, dim=1)


def resize_pos_embed(posemb, posemb_new):
    """Position-embedding table of a checkpoint trained at another resolution -> this model's grid (reference :1178-1195): the CLS
    row is kept, the square patch grid is resampled bilinearly (align_corners=False).  Host side, runs once at load time."""
    cls_row, grid = posemb[:, :1], posemb[0, 1:]
    side_old = int(math.sqrt(grid.shape[0]))
    side_new = int(math.sqrt(posemb_new.shape[1] - 1))
    dim = grid.shape[-1]
    grid = grid.t().reshape(1, dim, side_old, side_old)                               # [1, D, h, w]
    grid = F.interpolate(grid, size=(side_new, side_new), mode='bilinear')
    grid = grid.reshape(dim, side_new * side_new).t().unsqueeze(0)                    # [1, h'w', D]
    return torch.cat([cls_row, grid], dim=1)


def checkpoint_filter_fn(state_dict, model):
    """Make a DeiT-style checkpoint loadable (reference :1198-1213): unwrap {'model': ...}, give a patch-projection weight stored as a
    matrix its conv shape, and resample a position table of a different resolution."""
    source = state_dict.get('model', state_dict)
    conv_shape = model.patch_embed.proj.weight.shape

    def adapt(key, value):
        if 'patch_embed.proj.weight' in key and value.dim() < 4:
            return value.reshape(conv_shape[0], -1, conv_shape[2], conv_shape[3])
        if key == 'pos_embed' and value.shape != model.pos_embed.shape:
            return resize_pos_embed(value, model.pos_embed)
        return value

    return {key: adapt(key, value) for key, value in source.items()}


def _load_local(model, checkpoint_path, strict):
    """The reference's factories download DeiT weights (:1221 ff.); there is no network here, so a local file is
    the only source.  weights_only=True: nothing from the file is executed."""
    if checkpoint_path is None:
        return model
    sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    missing, unexpected = model.load_state_dict(checkpoint_filter_fn(sd, model), strict=strict)
    from d2s import ops
    ops.invalidate_bf16_weights()       # cached bf16 forms of frozen weights follow the version counter; a fresh load starts clean anyway
    print('# missing keys=', missing)
    print('# unexpected keys=', unexpected)
    return model


_GEOM = {"tiny": dict(embed_dim=192, num_heads=3), "small": dict(embed_dim=384, num_heads=6), "base": dict(embed_dim=768, num_heads=12)}


def _student(size, pruning_locs, keep_ratios, checkpoint_path=None, **kwargs):
    model = VisionTransformerDiffPruning(patch_size=16, depth=12, mlp_ratio=4, qkv_bias=True, pruning_loc=pruning_locs,
                                         token_ratio=keep_ratios, distill=True, **_GEOM[size], **kwargs)
    return _load_local(model, checkpoint_path, strict=False)


def _teacher(size, checkpoint_path=None):
    model = VisionTransformerTeacher(patch_size=16, depth=12, mlp_ratio=4, qkv_bias=True, **_GEOM[size])
    return _load_local(model, checkpoint_path, strict=True)


def dynamic_vit_tiny_patch16_224_student(pruning_locs, keep_ratios, **kwargs):
    return _student("tiny", pruning_locs, keep_ratios, **kwargs)


def dynamic_vit_small_patch16_224_student(pruning_locs, keep_ratios, **kwargs):
    return _student("small", pruning_locs, keep_ratios, **kwargs)


def dynamic_vit_base_patch16_224_student(pruning_locs, keep_ratios, **kwargs):
    return _student("base", pruning_locs, keep_ratios, **kwargs)


def dynamic_vit_tiny_patch16_224_teacher(checkpoint_path=None):
    return _teacher("tiny", checkpoint_path)


def dynamic_vit_small_patch16_224_teacher(checkpoint_path=None):
    return _teacher("small", checkpoint_path)


def dynamic_vit_base_patch16_224_teacher(checkpoint_path=None):
    return _teacher("base", checkpoint_path)
