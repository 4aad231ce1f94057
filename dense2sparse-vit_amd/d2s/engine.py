"""Training engine for the accelerated step: flat parameter / gradient arenas in HBM, fused AdamW, the step of
train.py:40-57, and the data-parallel gradient exchange (ddp_training.py:93 intent) as bucketed RCCL all-reduces on a
side stream that overlap the rest of backward.

Layout: every student parameter lives in ONE fp32 arena (each tensor padded to 1024 elements); gradients, exp_avg and
exp_avg_sq have arenas of the same layout.  The Functions in d2s.functional write parameter gradients straight into
the gradient arena (ops.grad_buffer), so the optimiser is a single kernel launch and a DDP bucket is a contiguous
slice - there is no flatten/unflatten copy anywhere.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import lib, ops


def _group_of(name, p):
    """utils.get_param_groups, utils.py:67-90 -> 'predictor' | 'base_no_decay' | 'base_decay' | None (never optimised)."""
    if "predictor" in name or "dist" in name:
        return "predictor"
    if "early_exit" in name:
        return "early_exit"
    if "cls_token" in name or "pos_embed" in name:
        return None
    if p.dim() == 1 or name.endswith(".bias"):
        return "base_no_decay"
    return "base_decay"


class ParamArena:
    """Moves the parameters of `model` into one flat fp32 buffer (in registration order) and owns the matching
    gradient buffer.  param.data becomes a view; state_dict()/load_state_dict() keep working."""

    def __init__(self, model, order=None):
        self.chunk = lib.query("d2s_adamw_chunk_elems")
        named = [(n, p) for n, p in model.named_parameters()]
        if order is not None:
            byname = dict(named)
            assert sorted(order) == sorted(byname), "order must list every parameter exactly once"
            named = [(n, byname[n]) for n in order]
        assert named and all(p.is_cuda and p.dtype == torch.float32 for _, p in named), "move the model to the GPU first"
        dev = named[0][1].device
        self.names, self.offsets, self.sizes, self.params_list = [], [], [], []
        off = 0
        for n, p in named:
            self.names.append(n)
            self.offsets.append(off)
            self.sizes.append(p.numel())
            self.params_list.append(p)
            off += ((p.numel() + self.chunk - 1) // self.chunk) * self.chunk
        self.total = off
        self.n_chunks = off // self.chunk
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views = {}
        for n, p, o, s in zip(self.names, self.params_list, self.offsets, self.sizes):
            v = self.params[o:o + s].view(p.shape)
            v.copy_(p.data)
            p.data = v
            gv = self.grads[o:o + s].view(p.shape)
            self.grad_views[n] = gv
            ops.register_grad_buffer(p, gv)
        ops.register_weight_arena(self.params)       # weights at stable addresses: their W^T copies may be cached (ops.linear_dgrad)

    def chunk_range(self, i):
        o = self.offsets[i]
        return o // self.chunk, (o + ((self.sizes[i] + self.chunk - 1) // self.chunk) * self.chunk) // self.chunk

    def check_alias(self):
        """The optimiser updates the arena; a parameter whose .data was re-assigned afterwards (.to(), .half(), load into a new
        tensor) would silently stop following it."""
        base = self.params.data_ptr()
        for n, p, o in zip(self.names, self.params_list, self.offsets):
            if p.data_ptr() != base + 4 * o:
                raise lib.D2SError(f"parameter {n} no longer aliases the parameter arena (re-assigned after TrainStep was built)")

    def collect_grads(self):
        """Make sure every live gradient sits in the arena (it does when the Functions produced it; a gradient that
        autograd materialised elsewhere is copied in) and detach .grad from autograd's bookkeeping."""
        for n, p in zip(self.names, self.params_list):
            if p.grad is not None and p.grad.data_ptr() != self.grad_views[n].data_ptr():
                self.grad_views[n].copy_(p.grad)


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) over a ParamArena in one kernel launch.
    Groups and learning rates follow utils.get_param_groups / utils.adjust_learning_rate (utils.py:67-147)."""

    def __init__(self, arena, lr=5e-4, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8):
        self.arena, self.betas, self.eps = arena, betas, eps
        self.exp_avg = torch.zeros_like(arena.params)
        self.exp_avg_sq = torch.zeros_like(arena.params)
        self.groups = [_group_of(n, p) for n, p in zip(arena.names, arena.params_list)]
        self.group_lr = {"predictor": lr, "base_no_decay": lr, "base_decay": lr, "early_exit": lr}
        self.group_wd = {"predictor": weight_decay, "base_no_decay": 0.0, "base_decay": weight_decay, "early_exit": weight_decay}
        self.steps = 0
        # torch.optim.AdamW counts updates per parameter (state['step'] advances only when the parameter has a gradient), so a tensor
        # frozen during the warm-up epochs starts its bias correction at t = 1 afterwards: one counter per arena chunk, on the device
        self.chunk_steps = torch.zeros(arena.n_chunks, dtype=torch.int32, device=arena.params.device)
        self._desc = None
        self._dirty = True
        self._wt = None

    def set_lrs(self, predictor_lr, backbone_lr):
        self.group_lr.update(predictor=predictor_lr, base_no_decay=backbone_lr, base_decay=backbone_lr)
        self._dirty = True

    def _build_desc(self):
        a = self.arena
        desc = np.zeros(a.n_chunks, dtype=[("lr", "<f4"), ("wd", "<f4"), ("active", "<i4"), ("pad", "<i4")])
        for i, (g, p) in enumerate(zip(self.groups, a.params_list)):
            c0, c1 = a.chunk_range(i)
            if g is None or g == "early_exit" or not p.requires_grad:
                continue      # early_exit: the head is never called (dynamic_vit.py:752-758), its gradients stay None and torch's AdamW
                              # skips such parameters entirely - so does this one
            desc["lr"][c0:c1] = self.group_lr[g]
            desc["wd"][c0:c1] = self.group_wd[g]
            desc["active"][c0:c1] = 1
        new = torch.from_numpy(desc.view(np.uint8).copy())
        if self._desc is None:
            self._desc = new.to(a.params.device)
        else:                           # in place: a captured step (TrainStep graph mode) keeps reading this buffer
            self._desc.copy_(new, non_blocking=False)
        self._dirty = False

    def mark_dirty(self):
        self._dirty = True

    def step(self, grad_scale=1.0):
        if self._dirty:
            self._build_desc()
        self.steps += 1
        a = self.arena
        ops.adamw_step(a.params, a.grads, self.exp_avg, self.exp_avg_sq, self._desc, a.n_chunks, self.betas[0], self.betas[1],
                       self.eps, self.steps, grad_scale, chunk_steps=self.chunk_steps)
        ops.bump_weights_epoch()        # the kernel wrote the parameters through raw pointers: cached W^T copies are stale

    def refresh_transposed_weights(self):
        """Rebuild the W^T copies of every weight of the arena in one launch (the input-gradient GEMMs read them, ops.linear_dgrad).
        TrainStep calls this at the START of every step, so whatever wrote the parameters since the last step - the fused AdamW, a
        broadcast into the arena, load_state_dict, a raw `.data` edit that no version counter sees - is picked up."""
        a = self.arena
        ops.bump_weights_epoch()
        if self._wt is None:
            self._wt = ops.TransposedArena(a.params, list(zip(a.params_list, a.offsets)))
        self._wt.refresh()
        if ops.bf16_io() and ops._BF16_WEIGHTS:      # bf16 mode: the GEMMs read bf16 mirrors of W and W^T made here, once per step
            if getattr(self, "_w16", None) is None:
                self._w16 = ops.Bf16Weights(a.params, list(zip(a.params_list, a.offsets)), self._wt)
            self._w16.refresh()

    def zero_grad(self):
        for p in self.arena.params_list:
            p.grad = None


def adjust_learning_rate(optimizer, model, step, epochs, lr, min_lr, warmup_steps, frozen=()):
    """utils.adjust_learning_rate (utils.py:93-147): cosine schedule, backbone frozen for the first `warmup_steps`
    epochs (requires_grad toggled exactly like the reference), backbone lr = min(0.01 lr, cos) afterwards."""
    cos_lr = (math.cos(step / epochs * math.pi) + 1) * 0.5
    cos_lr = min_lr + cos_lr * (lr - min_lr)
    predictor_lr = cos_lr
    backbone_lr = 0.0 if step < warmup_steps else min(lr * 0.01, cos_lr)
    for n, p in model.named_parameters():
        if n in frozen:     # frozen by construction (T2T's sinusoid pos_embed, the performer's random features): never trained
            continue
        is_pred = "dist" in n or "predictor" in n
        g = _group_of(n, p)
        if g is None:                       # cls_token / pos_embed: in no param group, follow the first loop of the reference
            p.requires_grad_(True if step >= warmup_steps else is_pred)
        elif g == "predictor":
            p.requires_grad_(predictor_lr != 0)
        else:
            p.requires_grad_(backbone_lr != 0)
    optimizer.set_lrs(predictor_lr, backbone_lr)
    return predictor_lr, backbone_lr


def execution_order(student):
    """Parameter names in forward-execution order (embed, then per block: the predictor that runs before it, the block;
    then norm / head).  Autograd runs backward in exactly the reverse order, so gradients become final from the END of
    an arena laid out this way towards its start - which is what makes a DDP bucket a contiguous tail slice."""
    names = [n for n, _ in student.named_parameters()]
    pre = [n for n in names if not (n.startswith("blocks.") or n.startswith("score_predictor.") or n.startswith("norm.")
                                    or n.startswith("head."))]
    order, starts = list(pre), {}
    locs = list(getattr(student, "pruning_loc", []))
    for i in range(len(student.blocks)):
        starts[i] = len(order)
        if i in locs:
            s_ = locs.index(i)
            order += [n for n in names if n.startswith(f"score_predictor.{s_}.")]
        order += [n for n in names if n.startswith(f"blocks.{i}.")]
    order += [n for n in names if n.startswith("norm.") or n.startswith("head.")]
    assert sorted(order) == sorted(names)
    return order, starts


class GradReducer:
    """Data-parallel gradient averaging (the DDP of ddp_training.py:93): tail slices of the gradient arena are
    all-reduced with RCCL on a side stream as soon as autograd has finished the layers that own them (reverse layer
    order), so the exchange overlaps the remaining backward; joined before the optimiser step."""

    def __init__(self, arena, group=None, bucket_mb=None, collective=None):
        """bucket_mb (default: D2S_DDP_BUCKET_MB, else 16): smallest slice worth a collective of its own.
        collective (default: D2S_DDP_COLLECTIVE, else "allreduce"): "allreduce" = one all_reduce per bucket; "rs_ag" = reduce_scatter
        into this rank's 1/world shard of the bucket followed by an all_gather of the shards - the two halves of a direct
        all-reduce over the fully connected xGMI mesh (SURVEY 8d sizes both forms); same sums, same result layout."""
        import os
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # its waits on the weight-gradient stream sit in its hardware queue: on the main stream's queue they would stall the backward itself
        self.stream = ops.concurrent_stream([torch.cuda.current_stream()], "gradient exchange") if arena.params.is_cuda else None
        if bucket_mb is None:
            bucket_mb = float(os.environ.get("D2S_DDP_BUCKET_MB", "16"))
        self.collective = collective or os.environ.get("D2S_DDP_COLLECTIVE", "allreduce")
        if self.collective not in ("allreduce", "rs_ag"):
            raise ValueError(f"collective {self.collective!r}: expected 'allreduce' or 'rs_ag'")
        self.bucket_mb = bucket_mb
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._hi = arena.total
        # arena ranges [lo, hi) whose gradients exist this epoch (ascending, merged); None = the whole arena.  requires_grad flips
        # between epochs (utils.py:112-147): in the warm-up epochs only the predictors train, and cls_token / pos_embed are never
        # optimised (utils.py:79-80), so their slices are not exchanged either (set_live_ranges, called from TrainStep.set_epoch)
        self.live = None
        self.force = False      # rehearsal: issue the collectives even with a single rank
        self.paused = False
        self.timing = False     # bench: bracket every collective with events on the side stream
        self._events, self._exposed, self._bytes, self._launched = [], [], 0, False
        self._n_collectives = 0

    def set_live_ranges(self, ranges):
        """ranges: iterable of (lo, hi) arena offsets holding live gradients; merged when they touch."""
        merged = []
        for lo, hi in sorted(ranges):
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            elif hi > lo:
                merged.append([lo, hi])
        self.live = [tuple(r) for r in merged]

    def live_elems(self):
        return self.arena.total if self.live is None else sum(hi - lo for lo, hi in self.live)

    def _launch(self, lo):
        hi, self._hi = self._hi, lo
        if (self.world == 1 and not self.force) or lo >= hi:
            return
        if self.live is None:
            self._reduce(lo, hi)
        else:       # only the parts of [lo, hi) that hold live gradients, from the arena's end towards its start
            for a, b in reversed(self.live):
                a, b = max(a, lo), min(b, hi)
                if b > a:
                    self._reduce(a, b)

    def _collective(self, buf, async_op):
        """SUM of `buf` over the ranks, result in `buf` on every rank.  -> list of work handles (async_op) or []"""
        n = buf.numel()
        if self.collective == "rs_ag" and self.world > 1 and n % self.world == 0:
            per = n // self.world
            shard = torch.empty(per, dtype=buf.dtype, device=buf.device)    # a buffer of its own: gloo on device tensors returns wrong sums
            w1 = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)   # for an aliased output
            if async_op:
                w1.wait()            # RCCL: a stream-level dependency, no host block; gloo's worker threads would otherwise start the
                                     # gather before the scatter has written the shard
            w2 = dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=async_op)
            self._n_collectives += 2
            return [w2] if async_op else []
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)     # also rs_ag's path for a slice the world does not divide
        self._n_collectives += 1
        return [w] if async_op else []

    def _reduce(self, lo, hi):
        buf = self.arena.grads[lo:hi]
        self._launched = True
        if self.stream is None:          # CPU tensors (gloo rehearsal): blocking
            self._collective(buf, False)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        wg = ops.weight_grad_stream()
        if wg is not None:                          # the weight gradients of this bucket were written on their own stream
            self.stream.wait_stream(wg)
        with torch.cuda.stream(self.stream):
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in self._collective(buf, True):
                w.wait()                 # stream-level wait: the side stream follows RCCL's stream
            if self.timing:
                e1.record()
                self._events.append((e0, e1))
                self._bytes += buf.numel() * 4

    def ready_from(self, lo):
        """Every gradient at arena offset >= lo is final."""
        if self.paused:          # a step that is being captured into a hipGraph: no collective from inside; finish() sends everything
            return
        if self._pending(lo) >= self.bucket_elems:
            self._launch(lo)

    def _pending(self, lo):
        if self.live is None:
            return self._hi - lo
        return sum(max(0, min(b, self._hi) - max(a, lo)) for a, b in self.live)

    def finish(self):
        """Flush the remainder, join, and return 1/world (folded into the optimiser's gradient scale)."""
        self._launch(0)
        if self._launched and self.stream is not None:
            cur = torch.cuda.current_stream()
            if self.timing:
                arrived, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                arrived.record(cur)
                done.record(self.stream)
                self._exposed.append((arrived, done))
            cur.wait_stream(self.stream)
        self._launched = False
        self._hi = self.arena.total
        return 1.0 / self.world

    def comm_summary(self, steps):
        """Per-step communication figures from the recorded events (call after a device synchronize): bytes all-reduced, time
        the collectives occupied the side stream, bus bandwidth 2(N-1)/N * bytes / time, and the exposed part = how long the
        main stream had to wait in finish() after the backward was already done."""
        if not self._events:
            return None
        t_ms = sum(a.elapsed_time(b) for a, b in self._events)
        exposed = sum(max(0.0, a.elapsed_time(b)) for a, b in self._exposed)
        n = max(self.world, 1)
        out = {"allreduce_bytes_per_step": self._bytes / steps, "buckets_per_step": len(self._events) / steps,
               "collectives_per_step": round(self._n_collectives / steps, 2), "collective": self.collective, "bucket_mb": self.bucket_mb,
               "allreduce_ms_per_step": round(t_ms / steps, 3), "exposed_ms_per_step": round(exposed / steps, 3),
               "bus_GBps": round(2.0 * (n - 1) / n * self._bytes / (t_ms * 1e-3) / 1e9, 1) if t_ms > 0 else None,
               "backend": "nccl (RCCL)", "world": n}
        self._events, self._exposed, self._bytes, self._n_collectives = [], [], 0, 0
        return out


class TrainStep:
    """One optimiser step as train.py:40-57 defines it: teacher forward (no grad), student forward, MaskLoss +
    BackboneLoss, warm-up switch (train.py:50-53), zero_grad / backward / step."""

    GRAPH_WARM_STEPS = 2        # eager steps of a given shape before it is captured (lazy one-time work happens there)

    def __init__(self, student, teacher, args, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0,
                 distributed=False, bucket_mb=None, collective=None, graph=None):
        """graph: capture the step into a hipGraph and replay it (None: D2S_STEP_GRAPH = 0 | 1 | auto, default 0; see _use_graph).
        Off by default: on ROCm 7.2 the replay of the ~1400-node graph is SLOWER than issuing the kernels (config 3, 32 images per GPU:
        2064 vs 2310 images/s; the replay call itself keeps the host busy for 11.4 ms against 12.5 ms of eager enqueue, and the GPU side
        gains nothing: profiles/r03_a_graph_vs_eager.txt) - the runtime walks the graph node by node on the host."""
        from losses import MaskLoss, BackboneLoss
        self.student, self.teacher, self.args = student, teacher, args
        self.teacher.eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)
        order, starts = execution_order(student)
        self.arena = ParamArena(student, order)
        self.block_offset = {i: self.arena.offsets[starts[i]] for i in starts}
        self.opt = FusedAdamW(self.arena, lr=lr, weight_decay=weight_decay)
        self.mask_loss_fn = MaskLoss(args, "train")
        self.backbone_loss_fn = BackboneLoss(args)
        self.metrics = {}
        self.lr, self.min_lr, self.epochs, self.warmup_steps = lr, min_lr, epochs, warmup_steps
        self.reducer = GradReducer(self.arena, bucket_mb=bucket_mb, collective=collective) if distributed else None
        if self.reducer is not None:
            student.grad_ready_hook = lambda i: self.reducer.ready_from(self.block_offset[i])
        self.frozen = frozenset(n for n, p in student.named_parameters() if not p.requires_grad)
        import os
        # the frozen teacher's forward runs on a second HIP stream beside the student's forward (independent until the losses):
        # +2.6 % images/s, identical losses (profiles/r02_f_teacher_stream_ab.txt).  D2S_TEACHER_STREAM=0 serialises them.
        two = os.environ.get("D2S_TEACHER_STREAM", "1") == "1" and self.arena.params.is_cuda
        # chosen, not just created: it must sit on another hardware queue than the stream the step is issued on (ops.concurrent_stream)
        self._teacher_stream = ops.concurrent_stream([torch.cuda.current_stream()], "teacher forward") if two else None
        if graph is None:
            graph = {"0": False, "1": True, "auto": "auto"}[os.environ.get("D2S_STEP_GRAPH", "0")]
        self.graph = None if graph == "auto" else bool(graph)      # True | False | None (auto: small per-rank batches only)
        self.graph_auto_max_rows = int(os.environ.get("D2S_STEP_GRAPH_AUTO_ROWS", "16384"))
        self._graphs = {}                   # key -> entry (see _graph_step)
        self._capture_stream = None
        self._ahead = None                  # teacher outputs of the next batch, issued one step early (_teacher_ahead)
        self._steps_seen = 0
        self._step_on_side = False          # the step is issued on a stream of its own choice (see _place_beside_process_group)
        self.pg_probe = None
        self.last_step_captured = False     # diagnostic: did the last call replay a graph
        self.set_epoch(0)

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.args.step = epoch
        out = adjust_learning_rate(self.opt, self.student, epoch, self.epochs, self.lr, self.min_lr, self.warmup_steps, self.frozen)
        if self.reducer is not None:       # the live gradient set changed: rebuild the bucket plan (SURVEY 8e caveat ii)
            a = self.arena
            chunk = a.chunk
            self.reducer.set_live_ranges(
                (o, o + ((sz + chunk - 1) // chunk) * chunk)
                for o, sz, p, g in zip(a.offsets, a.sizes, a.params_list, self.opt.groups)
                if p.requires_grad and g is not None and g != "early_exit")
        return out

    @staticmethod
    def _batch_key(images):
        return (images.data_ptr(), images._version, tuple(images.shape))

    def _teacher_ahead(self, images, ready):
        """Issue the frozen teacher's forward for the NEXT batch on the teacher stream now (it depends on nothing this step computes),
        so that it shares the GPU with this step's backward instead of with the next step's student forward.  `ready`: event on the
        caller's stream after which `images` is valid."""
        side = self._teacher_stream
        side.wait_event(ready)
        with torch.cuda.stream(side), torch.no_grad():
            out = self.teacher(images)
            done = side.record_event()
        images.record_stream(side)
        self._ahead = dict(key=self._batch_key(images), images=images, out=out, done=done)

    def forward_losses(self, images, labels, accumulate=True, next_images=None):
        ahead, self._ahead = self._ahead, None
        if ahead is not None and (self._teacher_stream is None or ahead["key"] != self._batch_key(images)):
            ahead = None                                # a different batch arrived than the one announced: its teacher pass is dropped
        if self._teacher_stream is not None:
            # The frozen teacher's forward and the student's forward are independent until the losses: the teacher runs on a second
            # HIP stream so that its kernels fill the CUs the student's kernels leave idle in their ramp-up / last residency round
            # (and vice versa).  Scratch buffers are per stream (ops.workspace), so the two forwards never share one.
            from .functional import shared_patch_columns, prime_patch_columns
            side, main = self._teacher_stream, torch.cuda.current_stream()
            ready = main.record_event() if next_images is not None else None
            if ahead is not None:                       # issued during the previous step (_teacher_ahead)
                logits_t, token_t, cls_attn = ahead["out"]
                logits_s, token_s, pred_logits, kept = self.student(images)
                main.wait_event(ahead["done"])
            else:
                with shared_patch_columns():
                    pe = getattr(self.student, "patch_embed", None)
                    if pe is not None and images.is_contiguous() and hasattr(pe, "patch_size") and getattr(self.teacher, "patch_embed", None) is not None:
                        prime_patch_columns(images, pe.patch_size[0])       # one im2col for both, built before the fork
                    side.wait_stream(main)
                    with torch.cuda.stream(side), torch.no_grad():
                        logits_t, token_t, cls_attn = self.teacher(images)
                    logits_s, token_s, pred_logits, kept = self.student(images)
                    main.wait_stream(side)
            for t in (logits_t, token_t, cls_attn):      # produced on the side stream, consumed (and later freed) on the main one
                t.record_stream(main)
            if next_images is not None:
                self._teacher_ahead(next_images, ready)
        else:
            from .functional import shared_patch_columns
            with shared_patch_columns():       # teacher and student embed the same images: one im2col pass for both
                with torch.no_grad():
                    logits_t, token_t, cls_attn = self.teacher(images)
                logits_s, token_s, pred_logits, kept = self.student(images)
        mask_loss = self.mask_loss_fn(pred_logits, cls_attn, kept, self.metrics, accumulate=accumulate)
        backbone_loss = self.backbone_loss_fn(logits_s, token_s, logits_t, token_t, kept, labels, self.metrics, accumulate=accumulate)
        loss = mask_loss if self.epoch < self.warmup_steps else backbone_loss + mask_loss     # train.py:50-53
        return loss, dict(mask_loss=mask_loss, backbone_loss=backbone_loss, kept=kept, logits_s=logits_s, token_s=token_s,
                          pred_logits=pred_logits, logits_t=logits_t, token_t=token_t, cls_attn=cls_attn)

    def __call__(self, images, labels, next_images=None):
        """next_images: the batch of the FOLLOWING call, if the caller already has it (a prefetching loader does).  The frozen teacher's
        forward for it is then issued during this step (see _teacher_ahead) - every step still runs exactly one teacher forward, the
        results are bit-identical, and a following call with any other batch simply recomputes.  Eager steps only."""
        self.arena.check_alias()
        if images.is_cuda and self._steps_seen == 0 and self.reducer is not None:
            self._place_beside_process_group()
        if images.is_cuda and self.graph is False and self._steps_seen in (1, 2):
            # the side streams were picked before anything else had run; an RCCL communicator (and whatever else creates streams late)
            # can have moved onto their hardware queues since: verify once the first step(s) are behind us
            main = self._capture_stream if self._step_on_side else torch.cuda.current_stream()
            self._teacher_stream = ops.recheck_stream(self._teacher_stream, [main], "teacher forward")
            ops.recheck_weight_grad_stream([main])
        self._steps_seen += 1
        if (self.graph is False and not self._step_on_side) or not images.is_cuda:
            return self._eager_step(images, labels, next_images)
        # Graph-capable mode: EVERY step - the eager warm-up steps, the capture, the replays, and steps of shapes that stay eager - is
        # issued on one side stream of this TrainStep.  Autograd pins a parameter's AccumulateGrad node to the stream of the forward that
        # created it, and such a node outlives its step whenever anything still references that step's graph (the model's own
        # `pred_logits` list does); a node pinned to the legacy default stream pulls that stream into a later capture, which then cannot
        # end.  The caller's stream waits for the side stream before this returns, so the caller sees an ordinary in-order step.
        cur = torch.cuda.current_stream()
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream()
        side = self._capture_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            info = self._graph_step(images, labels) if self._use_graph(images) else self._eager_step(images, labels, next_images)
        cur.wait_stream(side)
        if not self.last_step_captured:
            for v in info.values():            # allocated on the side stream, consumed (and freed) by the caller on its own stream
                for t in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
        return info

    def _place_beside_process_group(self):
        """First call of a data-parallel TrainStep: make sure the stream the step is issued on does not share a hardware queue with the
        process group's RCCL stream (ops.pg_stream_shares_queue).  If it does on ANY rank, every rank probes the same three candidate
        streams (the probes are collectives), and a rank whose stream collided moves its steps to the first candidate that does not -
        issued there the way the graph-capable mode issues them, the caller's stream waiting at both ends."""
        if not (dist.is_initialized() and dist.get_backend(self.reducer.group) == "nccl") or self.graph is not False:
            return
        cur = torch.cuda.current_stream()
        bad = ops.pg_stream_shares_queue(cur, self.reducer.group)
        flag = torch.tensor([1.0 if bad else 0.0], device=self.arena.params.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.reducer.group)
        self.pg_probe = {"caller_stream_shares_rccl_queue": bool(bad), "moved": False}
        if float(flag.item()) != 0.0:
            chosen = None
            for _ in range(3):
                cand = torch.cuda.Stream()
                if not ops.pg_stream_shares_queue(cand, self.reducer.group) and chosen is None:
                    chosen = cand
            if bad and chosen is not None:
                self._capture_stream, self._step_on_side = chosen, True
                self._teacher_stream = ops.recheck_stream(self._teacher_stream, [chosen], "teacher forward")
                self.reducer.stream = ops.recheck_stream(self.reducer.stream, [chosen], "gradient exchange")
                ops.recheck_weight_grad_stream([chosen])
                self.pg_probe["moved"] = True
        # the weight-gradient stream must not sit behind RCCL either: every bucket's collective would hold up the weight gradients issued
        # after it for as long as the exchange takes (the same fixed number of probes on every rank)
        if not ops._WGRAD_ENABLED:
            return
        step_stream = self._capture_stream if self._step_on_side else cur
        wg = ops.ensure_weight_grad_stream([step_stream])
        bad_wg = ops.pg_stream_shares_queue(wg, self.reducer.group)
        flag.fill_(1.0 if bad_wg else 0.0)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.reducer.group)
        self.pg_probe["weight_grad_stream_shares_rccl_queue"] = bool(bad_wg)
        if float(flag.item()) != 0.0:
            chosen = None
            for _ in range(3):
                cand = ops.concurrent_stream([step_stream], "weight gradients (off the RCCL queue)")
                if not ops.pg_stream_shares_queue(cand, self.reducer.group) and chosen is None:
                    chosen = cand
            if bad_wg and chosen is not None:
                ops.set_weight_grad_stream(chosen)
                self.pg_probe["weight_grad_stream_moved"] = True

    def _eager_step(self, images, labels, next_images=None):
        self.last_step_captured = False
        loss, info = self._forward_backward(images, labels, accumulate=True, next_images=next_images)
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.opt.step(grad_scale=scale)
        info["loss"] = loss.detach()
        return info

    def _forward_backward(self, images, labels, accumulate, next_images=None):
        self.opt.refresh_transposed_weights()
        self.student.train()
        loss, info = self.forward_losses(images, labels, accumulate=accumulate, next_images=next_images)
        self.opt.zero_grad()
        with ops.async_weight_grads():              # wgrad GEMMs trail the dgrad chain on a second stream; joined on exit
            loss.backward()
        self.arena.collect_grads()
        return loss, info

    # ---- the step as a hipGraph ----
    # With a fixed keep count every tensor of the step has a static shape and nothing in it synchronises with the host, so the ~450
    # C-ABI launches + autograd bookkeeping of a step (9.5 ms of Python at 32 images per GPU) can be recorded once
    # and replayed with one call.  What is captured: W^T / bf16 weight refresh, teacher and student forward (their two streams become
    # two branches of the graph), losses, backward (the weight-gradient stream is a third branch).  What stays outside: the gradient
    # exchange (RCCL collectives are issued eagerly after the replay - one flush of the live gradient set instead of buckets that
    # overlap backward), the fused AdamW launch (its per-chunk learning rates and step counters live in device buffers that are
    # updated in place), and the host-side running means of the two loss modules.
    def _use_graph(self, images):
        if self.graph is False or not images.is_cuda:
            return False
        if getattr(self.student, "patch_score_threshold", None) is not None:
            return False                         # dynamic keep ratio: not needed for the fixed-k path this serves; stays eager
        if self.graph is True:
            return True
        # auto: small per-rank batches, where the host cannot issue launches as fast as the GPU retires them (C3 / C4 regime).  With
        # a reducer the eager path's overlap of the exchange with backward is worth more than the launch time at large batches.
        n_tok = getattr(getattr(self.student, "patch_embed", None), "num_patches", 196) + 1
        return images.shape[0] * n_tok <= self.graph_auto_max_rows

    def _graph_key(self, images, labels):
        live = tuple(p.requires_grad for p in self.arena.params_list)
        return (tuple(images.shape), images.dtype, tuple(labels.shape), labels.dtype, self.epoch < self.warmup_steps, hash(live),
                ops.get_gemm_mode(), self._teacher_stream is not None, ops._WGRAD_ENABLED)

    def drop_graphs(self):
        self._graphs.clear()

    def _graph_step(self, images, labels):
        key = self._graph_key(images, labels)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 4:           # shapes come and go (a shorter last batch): keep the pool of captured steps small
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = {"seen": 0, "graph": None}
        if ent["graph"] is None and ent["seen"] < self.GRAPH_WARM_STEPS:
            ent["seen"] += 1
            return self._eager_step(images, labels)
        if ent["graph"] is None:
            self._capture(ent, images, labels)
        ent["images"].copy_(images, non_blocking=True)
        ent["labels"].copy_(labels, non_blocking=True)
        ent["graph"].replay()
        self.last_step_captured = True
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.opt.step(grad_scale=scale)
        self.mask_loss_fn.accumulate(self.metrics, ent["mask_last"])
        self.backbone_loss_fn.accumulate(self.metrics, ent["backbone_last"])
        return ent["info"]          # static tensors: overwritten by the next replay of this graph

    def _capture(self, ent, images, labels):
        if self.opt._dirty:
            self.opt._build_desc()               # host -> device copy: before the capture, not inside it
        ent["images"], ent["labels"] = images.clone(), labels.clone()
        # references into the previous step's autograd graph that the model itself holds (they would keep its AccumulateGrad nodes alive)
        for attr in ("pred_logits", "cls_attns", "kept_token_indices", "dropped_token_indices"):
            if isinstance(getattr(self.student, attr, None), list):
                setattr(self.student, attr, [])
        g = torch.cuda.CUDAGraph()
        if self.reducer is not None:
            self.reducer.paused = True
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g, stream=self._capture_stream):
                loss, info = self._forward_backward(ent["images"], ent["labels"], accumulate=False)
                info["loss"] = loss.detach()
        finally:
            if self.reducer is not None:
                self.reducer.paused = False
        ent["graph"], ent["info"] = g, info
        ent["ws_refs"] = list(ops._ws.values())       # scratch buffers the recorded kernels point at: must outlive the graph even if an
                                                      # eager call with a larger shape replaces them in ops._ws later
        ent["mask_last"], ent["backbone_last"] = self.mask_loss_fn.last, self.backbone_loss_fn.last
