"""CPU ORACLE - test infrastructure, not product code.

A plain-PyTorch (CPU, fp32) restatement of the reference's dense-to-sparse ViT training path, written
as pure functions over a state dict whose keys are the reference's own state-dict keys.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; nothing under
dense2sparse-vit_amd/ does (the product path has no CPU fallback and fails loudly without the HIP
library).

Parity status: PINNED.  tools/gen_golden.py imports the reference's unmodified
vit_models/dynamic_vit.py, vit_models/peturbed_topk.py and losses.py in the survey container, loads
the deterministic weights of dense2sparse-vit_amd/d2s/synth.py into the reference's own classes and
stores inputs/outputs under tests/golden/; tests/test_oracle_golden.py checks every function below
against those fixtures.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# configuration helper
# --------------------------------------------------------------------------------------------------
def make_cfg(img_size=224, patch=16, dim=384, depth=12, heads=6, mlp_ratio=4.0, num_classes=1000,
             pruning_loc=(), token_ratio=(), small_predictor=False, loss_type="kl_div", init_n=None,
             ln_eps=1e-6, predictor_bn=False):
    """Geometry of one student/teacher pair.  `init_n` reproduces the reference's hard-coded
    `init_n = 14 * 14` (vit_models/dynamic_vit.py:828,852) when left at None for 224x224 inputs; the
    keep count is always int(init_n * ratio) like the reference."""
    n_patches = (img_size // patch) ** 2
    return dict(img_size=img_size, patch=patch, dim=dim, depth=depth, heads=heads, mlp_ratio=mlp_ratio,
                num_classes=num_classes, pruning_loc=tuple(pruning_loc), token_ratio=tuple(token_ratio),
                small_predictor=small_predictor, loss_type=loss_type, n_patches=n_patches,
                init_n=(14 * 14 if init_n is None else init_n), ln_eps=ln_eps, predictor_bn=predictor_bn)


def keep_counts(cfg):
    """vit_models/dynamic_vit.py:852 - num_keep_node = int(init_n * ratio)."""
    return [int(cfg["init_n"] * r) for r in cfg["token_ratio"]]


def student_param_shapes(cfg):
    """(name, shape) for every parameter of VisionTransformerDiffPruning with the large LayerNorm
    predictor (vit_models/dynamic_vit.py:684-722, 491-531) or the small LN predictor (:409-426)."""
    D, C = cfg["dim"], cfg["num_classes"]
    hid = int(D * cfg["mlp_ratio"])
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, cfg["n_patches"] + 1, D)),
           ("patch_embed.proj.weight", (D, 3, cfg["patch"], cfg["patch"])), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg["depth"]):
        p = f"blocks.{i}."
        out += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
                (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
                (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
                (p + "mlp.fc1.weight", (hid, D)), (p + "mlp.fc1.bias", (hid,)),
                (p + "mlp.fc2.weight", (D, hid)), (p + "mlp.fc2.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,)), ("head.weight", (C, D)), ("head.bias", (C,))]
    bn = "bn." if cfg.get("predictor_bn") else ""      # BatchNormLayer wraps nn.BatchNorm1d as `.bn` (:350-367)
    for s in range(len(cfg["pruning_loc"])):
        p = f"score_predictor.{s}."
        if cfg["small_predictor"]:
            out += [(p + f"in_conv.0.{bn}weight", (D,)), (p + f"in_conv.0.{bn}bias", (D,)),
                    (p + "in_conv.1.weight", (D, D)), (p + "in_conv.1.bias", (D,))]
            widths = [D, D // 2, D // 4, 1]
            idx = [0, 1, 3, 4, 6, 7]
        else:
            out += [(p + f"in_conv.0.{bn}weight", (D,)), (p + f"in_conv.0.{bn}bias", (D,)),
                    (p + "in_conv.1.weight", (4 * D, D)), (p + "in_conv.1.bias", (4 * D,))]
            widths = [4 * D, 2 * D, D, D // 2, D // 4, 1]
            idx = [0, 1, 3, 4, 6, 7, 9, 10, 12, 13]
        for j in range(len(widths) - 1):
            ln_i, fc_i = idx[2 * j], idx[2 * j + 1]
            out += [(p + f"out_conv.{ln_i}.{bn}weight", (widths[j],)), (p + f"out_conv.{ln_i}.{bn}bias", (widths[j],)),
                    (p + f"out_conv.{fc_i}.weight", (widths[j + 1], widths[j])),
                    (p + f"out_conv.{fc_i}.bias", (widths[j + 1],))]
    return out


def teacher_param_shapes(cfg):
    """VisionTransformerTeacher (vit_models/dynamic_vit.py:1070-1101): the student minus predictors."""
    c = dict(cfg)
    c["pruning_loc"] = ()
    return student_param_shapes(c)


# --------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------
def patch_embed(sd, x, cfg):
    """PatchEmbed.forward, vit_models/dynamic_vit.py:300-306: stride-16 conv, flatten(2), transpose."""
    assert x.shape[2] == cfg["img_size"] and x.shape[3] == cfg["img_size"], "image size mismatch (:303-304)"
    y = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg["patch"])
    return y.flatten(2).transpose(1, 2)


def embed_tokens(sd, x, cfg):
    """vit_models/dynamic_vit.py:816-824 - patch embed, prepend CLS, add pos_embed (dropout p=0)."""
    t = patch_embed(sd, x, cfg)
    cls = sd["cls_token"].expand(t.shape[0], -1, -1)
    return torch.cat((cls, t), dim=1) + sd["pos_embed"]


def softmax_with_policy(attn, policy, eps=1e-6):
    """Attention.softmax_with_policy, vit_models/dynamic_vit.py:195-214."""
    B, N, _ = policy.shape
    work = torch.float64 if attn.dtype == torch.float64 else torch.float32     # the reference casts to float32 (:210); an fp64 run of
    pol = policy.reshape(B, 1, 1, N).to(work)                                    # the oracle (gradient-accuracy checks) stays in fp64
    eye = torch.eye(N, dtype=work).view(1, 1, N, N)
    pol = pol + (1.0 - pol) * eye
    attn = attn - attn.max(dim=-1, keepdim=True)[0]
    attn = attn.to(work).exp() * pol
    return (attn + eps / N) / (attn.sum(dim=-1, keepdim=True) + eps)


def attention(sd, pre, x, heads, policy=None):
    """Attention.forward, vit_models/dynamic_vit.py:216-236.  Returns (out, cls_row[B,H,n])."""
    B, n, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, n, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    a = a.softmax(dim=-1) if policy is None else softmax_with_policy(a, policy)
    o = (a @ v).transpose(1, 2).reshape(B, n, C)
    o = F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    return o, a[:, :, 0, :]


def mlp(sd, pre, x):
    """Mlp.forward, vit_models/dynamic_vit.py:169-175 (exact erf GELU, dropout p=0)."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def block(sd, i, x, cfg, policy=None):
    """Block.forward, vit_models/dynamic_vit.py:263-283 (DropPath = identity at rate 0, :249)."""
    p = f"blocks.{i}."
    D, eps = cfg["dim"], cfg["ln_eps"]
    y, cls_row = attention(sd, p + "attn.", F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps),
                           cfg["heads"], policy)
    x = x + y
    x = x + mlp(sd, p + "mlp.", F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    return x, cls_row


def predictor(sd, s, x, cfg, margins=None, training=True, bn_state=None):
    """PredictorLG.forward with topk_selection=True, vit_models/dynamic_vit.py:536-560.
    Large LN variant :491-531 (ReLU), small LN variant :409-426 (GELU), large BatchNorm variant :438-476 (predictor_bn: every
    LayerNorm replaced by BatchNormLayer :350-367 = BatchNorm1d over all B * N rows; `bn_state` maps "<key>running_mean/var" to the
    running estimates, updated in place in training mode; missing entries start at 0 / 1 like a fresh module).
    nn.LayerNorm / nn.BatchNorm1d default eps 1e-5.  Returns (scores, keep_probs) each [B, n-1]."""
    p = f"score_predictor.{s}."
    small = cfg["small_predictor"]
    use_bn = bool(cfg.get("predictor_bn"))
    act = F.gelu if (small and not use_bn) else F.relu      # the small BatchNorm variant keeps self.act = ReLU (:383-400)
    D = cfg["dim"]

    def norm(h, key):
        if not use_bn:
            w = sd[key + "weight"]
            return F.layer_norm(h, (w.shape[0],), w, sd[key + "bias"], 1e-5)
        w = sd[key + "bn.weight"]
        st = bn_state if bn_state is not None else {}
        rm = st.setdefault(key + "bn.running_mean", torch.zeros_like(w.detach()))
        rv = st.setdefault(key + "bn.running_var", torch.ones_like(w.detach()))
        Bh, Nh, Ch = h.shape
        return F.batch_norm(h.reshape(Bh * Nh, Ch), rm, rv, w, sd[key + "bn.bias"], training, 0.1, 1e-5).reshape(Bh, Nh, Ch)

    h = norm(x, p + "in_conv.0.")
    z = F.linear(h, sd[p + "in_conv.1.weight"], sd[p + "in_conv.1.bias"])
    if margins is not None:
        margins.append(float(z.detach().abs().min()))
    h = act(z)
    B, N, C = h.shape
    local_x = h[:, :, :C // 2]
    global_x = torch.mean(h[:, :, C // 2:], dim=1, keepdim=True)
    h = torch.cat([local_x, global_x.expand(B, N, C // 2)], dim=-1)
    idx = [0, 1, 3, 4, 6, 7] if small else [0, 1, 3, 4, 6, 7, 9, 10, 12, 13]
    nl = len(idx) // 2
    for j in range(nl):
        ln_i, fc_i = idx[2 * j], idx[2 * j + 1]
        h = norm(h, p + f"out_conv.{ln_i}.")
        h = F.linear(h, sd[p + f"out_conv.{fc_i}.weight"], sd[p + f"out_conv.{fc_i}.bias"])
        if j < nl - 1:
            if margins is not None:
                margins.append(float(h.detach().abs().min()))
            h = act(h)
    scores = h.flatten(-2, -1)
    if cfg["loss_type"] in ("kl_div", "mse"):
        keep_probs = F.softmax(scores, dim=-1)
    else:
        keep_probs = torch.sigmoid(scores)
    return scores, keep_probs


def select_topk(keep_probs, k):
    """vit_models/dynamic_vit.py:858-862 - full descending argsort, split at k, sort each ascending."""
    order = torch.argsort(keep_probs, dim=1, descending=True)
    kept = torch.sort(order[:, :k], dim=1)[0]
    dropped = torch.sort(order[:, k:], dim=1)[0]
    return kept, dropped


def select_topk_stable(keep_probs, k):
    """Same selection with the tie rule spelled out (larger value first, equal values lowest index first),
    independent of the sort implementation.  Used to pin the tie behaviour the HIP kernel implements."""
    order = torch.sort(keep_probs, dim=1, descending=True, stable=True)[1]
    kept = torch.sort(order[:, :k], dim=1)[0]
    dropped = torch.sort(order[:, k:], dim=1)[0]
    return kept, dropped


def gather_pack(x, kept):
    """vit_models/dynamic_vit.py:907-912 - keep CLS (row 0) and rows kept+1."""
    B = x.shape[0]
    pol = torch.cat([torch.zeros(B, 1, dtype=kept.dtype), kept + 1], dim=1)
    return torch.gather(x, 1, pol.unsqueeze(-1).expand(-1, -1, x.shape[-1]))


def batch_index_select(x, idx):
    """vit_models/dynamic_vit.py:39-60."""
    if x.dim() == 3:
        B, N, C = x.shape
        off = torch.arange(B, dtype=torch.long).view(B, 1) * N
        return x.reshape(B * N, C)[(idx + off).reshape(-1)].reshape(B, idx.shape[1], C)
    if x.dim() == 2:
        B, N = x.shape
        off = torch.arange(B, dtype=torch.long).view(B, 1) * N
        return x.reshape(B * N)[(idx + off).reshape(-1)].reshape(B, idx.shape[1])
    raise NotImplementedError


# --------------------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------------------
def student_forward(sd, x, cfg, training=True, bn_state=None):
    """VisionTransformerDiffPruning.forward, vit_models/dynamic_vit.py:814-1015 (patch_score_threshold
    None).  training: (logits, features, [pred_logits], [kept]); eval: (logits, [cls_attn], [pred_logits],
    [kept]).  Also returns aux = dict(dropped=[...], keep_probs=[...], cls_attns=[...]).  bn_state: running estimates of a
    predictor_bn student (see predictor)."""
    x = embed_tokens(sd, x, cfg)
    counts = keep_counts(cfg)
    pred_logits, kept_all, dropped_all, probs_all, cls_attns = [], [], [], [], []
    relu_margins = []   # min |pre-activation| of every predictor ReLU: a value at fp32-noise level means the gate (and every
    stage = 0           # gradient below it) is decided by rounding, which parity tests have to know about
    for i in range(cfg["depth"]):
        if i in cfg["pruning_loc"]:
            scores, probs = predictor(sd, stage, x[:, 1:], cfg, relu_margins, training=training, bn_state=bn_state)
            kept, dropped = select_topk(probs, counts[stage])
            pred_logits.append(scores)
            kept_all.append(kept)
            dropped_all.append(dropped)
            probs_all.append(probs)
            x = gather_pack(x, kept)
            stage += 1
        x, cls_row = block(sd, i, x, cfg)
        cls_attns.append(cls_row[:, :, 1:])
    x = F.layer_norm(x, (cfg["dim"],), sd["norm.weight"], sd["norm.bias"], cfg["ln_eps"])
    features = x[:, 1:]
    logits = F.linear(x[:, 0], sd["head.weight"], sd["head.bias"])
    aux = dict(dropped=dropped_all, keep_probs=probs_all, cls_attns=cls_attns, relu_margins=relu_margins)
    if training:
        return (logits, features, pred_logits, kept_all), aux
    return (logits, cls_attns, pred_logits, kept_all), aux


# --------------------------------------------------------------------------------------------------
# dynamic keep ratio (--patch-score-threshold), SURVEY 8f rank 3
# --------------------------------------------------------------------------------------------------
def select_threshold(keep_probs, threshold):
    """vit_models/dynamic_vit.py:881-891: ascending sort of the keep probabilities, cumulative sum, keep the tokens whose running sum
    exceeds the threshold, scattered back to token order.  Returns (mask [B,N] float, counts [B])."""
    val, idx = torch.sort(keep_probs.detach().clone())
    th = torch.cumsum(val, dim=-1) > threshold
    mask = torch.scatter(torch.zeros_like(th), 1, idx, th)     # the reference scatters into torch.empty: every slot is overwritten
    return mask.float(), th.sum(dim=1)


def select_threshold_stable(keep_probs, threshold):
    """The same selection with the tie rule spelled out: equal probabilities are taken lowest index first (a stable ascending sort),
    i.e. of two tied tokens that straddle the threshold the one with the lower index is the one dropped.  torch.sort without
    stable=True leaves that order to the sort implementation (on this build pairs of equal values come out in either order), so the
    reference itself only defines the NUMBER of kept tokens and the multiset of their probabilities on such rows; this is the rule the
    HIP kernel implements."""
    val, idx = torch.sort(keep_probs.detach().clone(), stable=True)
    th = torch.cumsum(val, dim=-1) > threshold
    mask = torch.scatter(torch.zeros_like(th), 1, idx, th)
    return mask.float(), th.sum(dim=1)


def student_forward_threshold_train(sd, x, cfg, threshold, relu_margins=None):
    """VisionTransformerDiffPruning.forward in training mode with patch_score_threshold set (:826-894, :981-983, :993-1011): no token
    is removed; a stage's mask becomes the key policy of that block and every later block (softmax_with_policy, :195-214), blocks
    before the first stage use the all-ones policy, a later stage replaces the mask.  Returns (logits, features [B,N,D],
    [pred_logits per stage], [mask [B,N] per stage]) - the reference returns the LAST stage's pred_logits / mask only (:1011).
    relu_margins (optional list): receives min |pre-activation| of every predictor ReLU (see student_forward)."""
    x = embed_tokens(sd, x, cfg)
    B, n, _ = x.shape
    policy = torch.ones(B, n, 1, dtype=x.dtype)
    pred_logits, masks = [], []
    stage = 0
    margins = relu_margins if relu_margins is not None else []
    for i in range(cfg["depth"]):
        if i in cfg["pruning_loc"]:
            scores, probs = predictor(sd, stage, x[:, 1:], cfg, margins)
            mask, _ = select_threshold(probs, threshold)
            mask = mask.to(x.dtype)
            policy = torch.cat((torch.ones(B, 1, dtype=x.dtype), mask), dim=1).unsqueeze(-1)
            pred_logits.append(scores)
            masks.append(mask)
            stage += 1
        x, _ = block(sd, i, x, cfg, policy=policy)
    x = F.layer_norm(x, (cfg["dim"],), sd["norm.weight"], sd["norm.bias"], cfg["ln_eps"])
    logits = F.linear(x[:, 0], sd["head.weight"], sd["head.bias"])
    return logits, x[:, 1:], pred_logits, masks


def student_forward_threshold_eval(sd, x, cfg, threshold):
    """Inference with a dynamic keep ratio (:935-949), one pruning stage, with the reference's undefined `score` read as the stage's
    keep probabilities and its flat boolean indexing done per image (the reference's reshape(B, -1, D) only exists for equal counts):
    every image goes on with its CLS token and its own kept tokens.  PARITY UNPINNED (the reference raises NameError here); this is
    the build's definition, checked per image against the same blocks run densely on the kept subset.
    Returns (logits [B,C], [normed tokens per image [n_b, D]], pred_logits, mask)."""
    assert len(cfg["pruning_loc"]) == 1
    x = embed_tokens(sd, x, cfg)
    loc = cfg["pruning_loc"][0]
    for i in range(loc):
        x, _ = block(sd, i, x, cfg)
    scores, probs = predictor(sd, 0, x[:, 1:], cfg, training=False)
    mask, _ = select_threshold(probs, threshold)
    logits, feats = [], []
    for b in range(x.shape[0]):
        keep = torch.cat((torch.ones(1, dtype=torch.bool), mask[b] > 0))
        xb = x[b:b + 1, keep]
        for i in range(loc, cfg["depth"]):
            xb, _ = block(sd, i, xb, cfg)
        xb = F.layer_norm(xb, (cfg["dim"],), sd["norm.weight"], sd["norm.bias"], cfg["ln_eps"])
        logits.append(F.linear(xb[:, 0], sd["head.weight"], sd["head.bias"]))
        feats.append(xb[0])
    return torch.cat(logits, dim=0), feats, scores, mask


def mask_loss_threshold(pred_logits, cls_attn, masks, threshold):
    """The build's fix of MaskLoss for the dynamic-keep-ratio path (the reference's loop runs over the batch dimension of the mask
    tensor, losses.py:81, and cannot run): per stage KL(log_softmax(scores) || log target) over ALL N tokens (nothing was gathered),
    accuracy = agreement of the stage's mask with the teacher target thresholded by the same rule.  PARITY UNPINNED."""
    target = teacher_target(cls_attn)
    gt, _ = select_threshold(target, threshold)
    loss, accs = 0, []
    for i in range(len(masks)):
        loss = loss + F.kl_div(F.log_softmax(pred_logits[i], dim=-1), torch.log(target), log_target=True, reduction="batchmean")
        accs.append(torch.sum(masks[i] == gt) / gt.numel())
    return loss, accs


def backbone_loss_threshold(logits_s, token_s, logits_t, token_t, masks, labels):
    """The build's fix of BackboneLoss' threshold branch (losses.py:216-218: undefined `C`, float-row indexing): the token KL over
    the tokens the LAST stage keeps, paired by position, averaged over the kept rows.  PARITY UNPINNED."""
    cls_loss = F.cross_entropy(logits_s, labels)
    cls_kl = F.kl_div(F.log_softmax(logits_s, dim=-1), F.log_softmax(logits_t, dim=-1), reduction="batchmean", log_target=True)
    keep = masks[-1].reshape(-1) > 0
    C = token_t.shape[-1]
    ts, tt = token_s.reshape(-1, C)[keep], token_t.reshape(-1, C)[keep]
    tok_kl = F.kl_div(F.log_softmax(ts, dim=-1), F.log_softmax(tt, dim=-1), reduction="batchmean", log_target=True)
    return cls_loss + cls_kl + tok_kl, cls_loss, cls_kl, tok_kl


def patch_drop_mask(kept, dropped):
    """visualizations.py:18-26: kept / dropped id lists of one stage -> 0/1 mask in token order."""
    token_idx = torch.cat((kept, dropped), dim=1)
    srt = torch.cat((torch.ones_like(kept), torch.zeros_like(dropped)), dim=1)
    out = torch.empty_like(srt)
    out.scatter_(dim=1, index=token_idx.long(), src=srt)
    return out


def teacher_forward(sd, x, cfg):
    """VisionTransformerTeacher.forward, vit_models/dynamic_vit.py:1150-1176:
    (logits, tokens[B,N,D], cls_attn[B,depth,H,N+1]) - CLS rows detached (:1165)."""
    x = embed_tokens(sd, x, cfg)
    rows = []
    for i in range(cfg["depth"]):
        x, cls_row = block(sd, i, x, cfg)
        rows.append(cls_row.detach())
    f = F.layer_norm(x, (cfg["dim"],), sd["norm.weight"], sd["norm.bias"], cfg["ln_eps"])
    logits = F.linear(f[:, 0], sd["head.weight"], sd["head.bias"])
    return logits, f[:, 1:], torch.stack(rows, dim=1)


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
def topk_mask(values, keep_ratio):
    """MaskLoss.get_mask_from_pred_logits / get_mask_from_cls_attns, losses.py:121-164."""
    order = torch.argsort(values, dim=-1, descending=True)
    nk = int(values.shape[-1] * keep_ratio)
    mask = torch.cat((torch.ones_like(order[:, :nk]), torch.zeros_like(order[:, nk:])), dim=-1).float()
    mask.scatter_(index=order, src=mask.clone(), dim=-1)
    return mask


def teacher_target(cls_attn):
    """losses.py:76-79 - mean over layers, max over heads, drop CLS column, renormalise."""
    w = torch.mean(cls_attn, dim=1)
    w, _ = torch.max(w, dim=1)
    return w[:, 1:] / torch.sum(w[:, 1:], dim=-1, keepdim=True)


def mask_loss_kl(pred_logits, cls_attn, kept, keep_ratios):
    """MaskLoss.forward, kl_div branch, losses.py:75-96.  Returns (loss, [mask_acc_i])."""
    target = teacher_target(cls_attn)
    loss = 0
    accs = []
    for i in range(len(kept)):
        if i > 0:
            ratio = keep_ratios[i] / keep_ratios[i - 1]
            gt = topk_mask(torch.gather(target, 1, kept[i - 1]), ratio)
            pm = topk_mask(F.softmax(pred_logits[i], dim=-1), ratio)
            target = torch.gather(target, 1, kept[i - 1])
            target = target / torch.sum(target, dim=1, keepdim=True)
        else:
            gt = topk_mask(target, keep_ratios[i])
            pm = topk_mask(F.softmax(pred_logits[i], dim=-1), keep_ratios[i])
        loss = loss + F.kl_div(F.log_softmax(pred_logits[i], dim=-1), torch.log(target),
                               log_target=True, reduction="batchmean")
        accs.append(torch.sum(pm == gt) / pm.numel())
    return loss, accs


def mask_loss_mse(pred_logits, cls_attn, kept):
    """MaskLoss.forward, mse branch, losses.py:61-73: 100 * mse(raw scores, renormalised teacher target) per stage; the target of
    stage i > 0 is re-gathered by kept_{i-1} and renormalised.  (The reference accumulates no mask accuracy in this branch.)"""
    target = teacher_target(cls_attn)
    loss = 0
    for i in range(len(kept)):
        if i > 0:
            target = torch.gather(target, 1, kept[i - 1])
            target = target / torch.sum(target, dim=1, keepdim=True)
        loss = loss + 100 * F.mse_loss(pred_logits[i], target, reduction="mean")
    return loss


def backbone_loss(logits_s, token_s, logits_t, token_t, kept, labels):
    """BackboneLoss.forward, losses.py:185-227 (mixup off -> CrossEntropyLoss, :174).  The teacher tokens
    are gathered with the LAST stage's (stage-relative) ids exactly as the reference does (:212).
    Returns (total, cls_loss, cls_kl, token_kl)."""
    if labels.dtype.is_floating_point:    # mixup: timm's SoftTargetCrossEntropy (losses.py:170-172) = mean_b sum_c -t log_softmax(x)
        cls_loss = torch.sum(-labels * F.log_softmax(logits_s, dim=-1), dim=-1).mean()
    else:
        cls_loss = F.cross_entropy(logits_s, labels)
    cls_kl = F.kl_div(F.log_softmax(logits_s, dim=-1), F.log_softmax(logits_t, dim=-1),
                      reduction="batchmean", log_target=True)
    C = token_t.shape[-1]
    tt = torch.gather(token_t, 1, kept[-1].unsqueeze(-1).expand(-1, -1, C)).reshape(-1, C)
    ts = token_s.reshape(-1, C)
    tok_kl = F.kl_div(F.log_softmax(ts, dim=-1), F.log_softmax(tt, dim=-1), reduction="batchmean",
                      log_target=True)
    return cls_loss + cls_kl + tok_kl, cls_loss, cls_kl, tok_kl


# --------------------------------------------------------------------------------------------------
# perturbed top-k
# --------------------------------------------------------------------------------------------------
def perturbed_topk_fwd(x, noise, k, sigma):
    """PerturbedTopKFunction.forward, vit_models/peturbed_topk.py:27-51, with the noise tensor injected.
    x [b,d], noise [b,nS,d] -> indicators [b,k,d] and the sorted ids [b,nS,k]."""
    d = x.shape[1]
    pert = x[:, None, :] + noise * sigma
    ids = torch.topk(pert, k=k, dim=-1, sorted=False).indices
    ids = torch.sort(ids, dim=-1).values
    return F.one_hot(ids, num_classes=d).float().mean(dim=1), ids


def perturbed_topk_bwd(grad_out, noise, ids, sigma):
    """PerturbedTopKFunction.backward, vit_models/peturbed_topk.py:76-79."""
    d = noise.shape[-1]
    onehot = F.one_hot(ids, num_classes=d).float()
    e = torch.einsum("bnkd,bnd->bkd", onehot, noise) / noise.shape[1] / sigma
    return torch.einsum("bkd,bkd->bd", grad_out, e)


# --------------------------------------------------------------------------------------------------
# the training step (train.py:40-57) with the optimiser of mask_predictor.py / utils.py:67-90
# --------------------------------------------------------------------------------------------------
def param_groups(named_params, weight_decay):
    """utils.get_param_groups, utils.py:67-90.  named_params: iterable of (name, tensor)."""
    decay, no_decay, pred = [], [], []
    for name, p in named_params:
        if "predictor" in name or "dist" in name:
            pred.append(p)
        elif not p.requires_grad:
            continue
        elif "cls_token" in name or "pos_embed" in name:
            continue
        elif p.dim() == 1 or name.endswith(".bias"):
            no_decay.append(p)
        else:
            decay.append(p)
    return [dict(params=pred, weight_decay=weight_decay, name="predictor"),
            dict(params=no_decay, weight_decay=0.0, name="base_no_decay"),
            dict(params=decay, weight_decay=weight_decay, name="base_decay")]


def cosine_lrs(step, epochs, lr, min_lr, warmup_steps):
    """utils.adjust_learning_rate, utils.py:96-122: (predictor_lr, backbone_lr)."""
    cos_lr = (math.cos(step / epochs * math.pi) + 1) * 0.5
    cos_lr = min_lr + cos_lr * (lr - min_lr)
    if step < warmup_steps:
        return cos_lr, 0.0
    return cos_lr, min(lr * 0.01, cos_lr)


def train_step_losses(sd_s, sd_t, cfg, x, labels, warmup=False):
    """train.py:40-53: teacher fwd (no grad), student fwd, MaskLoss + BackboneLoss, warm-up switch."""
    tcfg = dict(cfg)
    tcfg["pruning_loc"] = ()
    with torch.no_grad():
        logits_t, token_t, cls_attn = teacher_forward(sd_t, x.clone(), tcfg)
    (logits_s, token_s, pred_logits, kept), aux = student_forward(sd_s, x.clone(), cfg, training=True)
    m_loss, accs = mask_loss_kl(pred_logits, cls_attn, kept, cfg["token_ratio"])
    b_loss, cls_loss, cls_kl, tok_kl = backbone_loss(logits_s, token_s, logits_t, token_t, kept, labels)
    total = m_loss if warmup else b_loss + m_loss
    return total, dict(mask_loss=m_loss, backbone_loss=b_loss, cls_loss=cls_loss, cls_kl=cls_kl,
                       token_kl=tok_kl, mask_accs=accs, logits_s=logits_s, token_s=token_s,
                       pred_logits=pred_logits, kept=kept, logits_t=logits_t, token_t=token_t,
                       cls_attn=cls_attn, aux=aux)


class TrainState:
    """Holds leaf parameter tensors + AdamW exactly as mask_predictor.py:224-232 builds them
    (AdamW over utils.get_param_groups; cls_token / pos_embed are never optimised, utils.py:79-80)."""

    def __init__(self, sd_student, sd_teacher, cfg, lr=1e-3, min_lr=1e-5, weight_decay=0.05, epochs=30,
                 warmup_steps=0):
        self.cfg = cfg
        self.sd_s = {k: v.clone().requires_grad_(True) for k, v in sd_student.items()}
        self.sd_t = {k: v.clone() for k, v in sd_teacher.items()}
        self.lr, self.min_lr, self.epochs, self.warmup_steps = lr, min_lr, epochs, warmup_steps
        self.groups = param_groups(self.sd_s.items(), weight_decay)
        self.opt = torch.optim.AdamW([g for g in self.groups if len(g["params"])], lr=lr)
        self.set_epoch(0)

    def set_epoch(self, step):
        """utils.adjust_learning_rate: lr per group and requires_grad toggling (utils.py:110-147)."""
        self.epoch = step
        pred_lr, bb_lr = cosine_lrs(step, self.epochs, self.lr, self.min_lr, self.warmup_steps)
        for name, p in self.sd_s.items():
            is_pred = "predictor" in name or "dist" in name
            p.requires_grad_(True if is_pred else step >= self.warmup_steps)
        for g in self.opt.param_groups:
            g["lr"] = pred_lr if g["name"] == "predictor" else bb_lr
            for p in g["params"]:
                p.requires_grad_(g["lr"] != 0)

    def step(self, x, labels):
        """train.py:40-57."""
        total, info = train_step_losses(self.sd_s, self.sd_t, self.cfg, x, labels,
                                        warmup=self.epoch < self.warmup_steps)
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        info["loss"] = total.detach()
        return info


# --------------------------------------------------------------------------------------------------
# T2T path: vit_models/t2t_vit.py, token_performer.py, token_transformer.py,
# transformer_block.py.  Dropout inside Token_performer (p = 0.1, token_performer.py:13,24) is the
# identity here: parity is defined for p = 0 / eval mode.
# --------------------------------------------------------------------------------------------------
def sinusoid_encoding(n_position, d_hid):
    """transformer_block.get_sinusoid_encoding, transformer_block.py:78-88 (float64 table -> FloatTensor)."""
    import numpy as np
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0)


def unfold_tokens(x_bchw, k, s, p):
    """nn.Unfold(k, s, p)(x).transpose(1, 2), t2t_vit.py:85,92,99: [B,C,H,W] -> [B, L, C*k*k]."""
    return F.unfold(x_bchw, kernel_size=k, stride=s, padding=p).transpose(1, 2)


def tokens_to_image(x):
    """t2t_vit.py:90,97: [B, HW, C] -> [B, C, sqrt(HW), sqrt(HW)]."""
    B, HW, C = x.shape
    h = int(math.isqrt(HW))
    return x.transpose(1, 2).reshape(B, C, h, h)


def prm_exp(x, w):
    """Token_performer.prm_exp, token_performer.py:31-43."""
    m = w.shape[0]
    xd = (x * x).sum(dim=-1, keepdim=True).repeat(1, 1, m) / 2
    wtx = torch.einsum("bti,mi->btm", x, w)
    return torch.exp(wtx - xd) / math.sqrt(m)


def token_performer(sd, pre, x):
    """Token_performer.forward, token_performer.py:45-59 (emb = in_dim, 1 head, epsilon 1e-8, dropout off)."""
    emb = sd[pre + "proj.weight"].shape[0]
    h = F.layer_norm(x, (x.shape[-1],), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    k, q, v = torch.split(F.linear(h, sd[pre + "kqv.weight"], sd[pre + "kqv.bias"]), emb, dim=-1)
    kp, qp = prm_exp(k, sd[pre + "w"]), prm_exp(q, sd[pre + "w"])
    D = torch.einsum("bti,bi->bt", qp, kp.sum(dim=1)).unsqueeze(dim=2)
    kptv = torch.einsum("bin,bim->bnm", v, kp)
    y = torch.einsum("bti,bni->btn", qp, kptv) / (D.repeat(1, 1, emb) + 1e-8)
    y = v + F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    h2 = F.layer_norm(y, (emb,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    m = F.linear(F.gelu(F.linear(h2, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])), sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"])
    return y + m


def token_transformer(sd, pre, x):
    """Token_transformer.forward, token_transformer.py:26-60: 1 head of width in_dim, scale dim**-0.5, skip through v."""
    dim = x.shape[-1]
    in_dim = sd[pre + "attn.proj.weight"].shape[0]
    B, N, _ = x.shape
    h = F.layer_norm(x, (dim,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd.get(pre + "attn.qkv.bias")).reshape(B, N, 3, 1, in_dim).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * (dim ** -0.5)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, in_dim)
    y = v.squeeze(1) + F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    h2 = F.layer_norm(y, (in_dim,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    m = F.linear(F.gelu(F.linear(h2, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return y + m


def t2t_module(sd, x, tokens_type="performer"):
    """T2T_module.forward, t2t_vit.py:83-104."""
    stage = token_performer if tokens_type == "performer" else token_transformer
    pre = "tokens_to_token."
    t = unfold_tokens(x, 7, 4, 2)
    t = stage(sd, pre + "attention1.", t)
    t = unfold_tokens(tokens_to_image(t), 3, 2, 1)
    t = stage(sd, pre + "attention2.", t)
    t = unfold_tokens(tokens_to_image(t), 3, 2, 1)
    return F.linear(t, sd[pre + "project.weight"], sd[pre + "project.bias"])


def plain_block(sd, i, x, heads, eps=1e-5, want_cls=False):
    """transformer_block.Block.forward, transformer_block.py:72-75 (qkv_bias False by default, LayerNorm eps 1e-5)."""
    p = f"blocks.{i}."
    D = x.shape[-1]
    B, n, _ = x.shape
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd.get(p + "attn.qkv.bias")).reshape(B, n, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * ((D // heads) ** -0.5)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, n, D)
    x = x + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h2 = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    x = x + F.linear(F.gelu(F.linear(h2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return (x, a[:, :, 0, :]) if want_cls else x


def t2t_forward(sd, x, depth, heads, tokens_type="performer"):
    """T2T_ViT.forward_features + forward, t2t_vit.py:156-179: (logits, [normed output of every block])."""
    B = x.shape[0]
    t = t2t_module(sd, x, tokens_type)
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    D = t.shape[-1]
    heads_out = []
    for i in range(depth):
        t = plain_block(sd, i, t, heads)
        heads_out.append(F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], 1e-5))
    f = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return F.linear(f[:, 0], sd["head.weight"], sd["head.bias"]), heads_out


def t2t_param_shapes(img_size, dim, depth, heads, mlp_ratio, num_classes, tokens_type="performer", token_dim=64,
                     pruning_loc=(), in_chans=3):
    """State-dict entries of T2T_ViT (t2t_vit.py:106-141) in registration order (+ score predictors for the
    build-defined pruned student).  pos_embed and the performers' `w` are frozen parameters."""
    out = [("cls_token", (1, 1, dim)), ("pos_embed", (1, (img_size // 16) ** 2 + 1, dim))]
    for a, d_in in (("attention1", in_chans * 49), ("attention2", token_dim * 9)):
        p = f"tokens_to_token.{a}."
        if tokens_type == "performer":
            out += [(p + "w", (token_dim // 2, token_dim)), (p + "kqv.weight", (3 * token_dim, d_in)), (p + "kqv.bias", (3 * token_dim,)),
                    (p + "proj.weight", (token_dim, token_dim)), (p + "proj.bias", (token_dim,)),
                    (p + "norm1.weight", (d_in,)), (p + "norm1.bias", (d_in,)), (p + "norm2.weight", (token_dim,)), (p + "norm2.bias", (token_dim,)),
                    (p + "mlp.0.weight", (token_dim, token_dim)), (p + "mlp.0.bias", (token_dim,)),
                    (p + "mlp.2.weight", (token_dim, token_dim)), (p + "mlp.2.bias", (token_dim,))]
        else:
            out += [(p + "norm1.weight", (d_in,)), (p + "norm1.bias", (d_in,)), (p + "attn.qkv.weight", (3 * token_dim, d_in)),
                    (p + "attn.proj.weight", (token_dim, token_dim)), (p + "attn.proj.bias", (token_dim,)),
                    (p + "norm2.weight", (token_dim,)), (p + "norm2.bias", (token_dim,)),
                    (p + "mlp.fc1.weight", (token_dim, token_dim)), (p + "mlp.fc1.bias", (token_dim,)),
                    (p + "mlp.fc2.weight", (token_dim, token_dim)), (p + "mlp.fc2.bias", (token_dim,))]
    out += [("tokens_to_token.project.weight", (dim, token_dim * 9)), ("tokens_to_token.project.bias", (dim,))]
    hid = int(dim * mlp_ratio)
    for i in range(depth):
        p = f"blocks.{i}."
        out += [(p + "norm1.weight", (dim,)), (p + "norm1.bias", (dim,)), (p + "attn.qkv.weight", (3 * dim, dim)),
                (p + "attn.proj.weight", (dim, dim)), (p + "attn.proj.bias", (dim,)), (p + "norm2.weight", (dim,)), (p + "norm2.bias", (dim,)),
                (p + "mlp.fc1.weight", (hid, dim)), (p + "mlp.fc1.bias", (hid,)), (p + "mlp.fc2.weight", (dim, hid)), (p + "mlp.fc2.bias", (dim,))]
    out += [("norm.weight", (dim,)), ("norm.bias", (dim,)), ("head.weight", (num_classes, dim)), ("head.bias", (num_classes,))]
    if pruning_loc:
        c = make_cfg(dim=dim, pruning_loc=pruning_loc, token_ratio=(0.5,) * len(pruning_loc))
        out += [e for e in student_param_shapes(c) if e[0].startswith("score_predictor.")]
    return out
