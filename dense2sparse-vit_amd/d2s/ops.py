"""Tensor-level wrappers over the C ABI (one Python function per entry point, no arithmetic here).

Conventions: fp32 contiguous device tensors unless a row map says otherwise; 2-D views [rows, features].
"""
import os
import threading

import torch

from . import lib

EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_GELU, EPI_BIAS_RESID = 0, 1, 2, 3, 4
EPI_MUL_GELU_GRAD, EPI_MUL_RELU_MASK, EPI_BIAS_ROWADD, EPI_ACCUM = 5, 6, 7, 8
EPI_BIAS_GELU_Z16, EPI_MUL_GELU_GRAD_Z16 = 9, 10      # d2s_gemm_f32_bf16io only: the saved GELU pre-activation in bf16 (chosen by gemm() from the tensor's dtype)
NT, NN, TN = 0, 1, 2

_ws = {}
_WS_NEED = {}       # (layout, M, N, K, mode) -> workspace bytes of d2s_gemm_f32 (a pure function of its arguments)
_BF16_ATTENTION = os.environ.get("D2S_BF16_ATTENTION", "1") != "0"
_BF16_PREACT = os.environ.get("D2S_BF16_PREACT", "1") != "0"      # bf16 data path: fc1's pre-activation is saved in bf16 (0: fp32, A/B)


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream): kernels of one stream reuse it in order, kernels of different streams (the
    optional teacher stream, d2s.engine) never share one."""
    key = (device.type, device.index, lib.stream() if device.type == "cuda" else 0)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def workspace_on(stream_obj, nbytes, device):
    """The scratch buffer of another stream than the current one (the weight-gradient stream of a composite backward call) - the same
    buffer workspace() hands out inside `with torch.cuda.stream(stream_obj)`.  A (re)allocation happens UNDER that stream: the caching
    allocator ties a block to the stream that was current when it was allocated, and a block tied to the caller's stream would, once a
    larger request replaces it, be handed to the caller's next tensor while launches queued on the other stream still use it (seen as a
    wrong predictor weight gradient in the first step of one run in three, tests/test_graph_gpu.py)."""
    key = (device.type, device.index, stream_obj.cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        with torch.cuda.stream(stream_obj):
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


# bf16 data path of the bf16 arithmetic mode: producers (LayerNorm, the attention forward, GEMM epilogues) emit a bf16 copy of what the
# next GEMM multiplies, and that GEMM reads it directly instead of converting its fp32 operand first (d2s_gemm_f32_bf16io).
_BF16_IO = os.environ.get("D2S_BF16_IO", "1") != "0"


def bf16_io():
    return _BF16_IO and get_gemm_mode() == GEMM_BF16


def bf16_buffer(rows, cols, device):
    return torch.empty((rows, cols), dtype=torch.bfloat16, device=device)


# A gradient that leaves one autograd Function and enters the next (a block's dx = the previous block's dy) cannot carry its bf16 copy
# through autograd; the producer parks it here under the fp32 tensor's address and the consumer takes it out again.  An entry is used at
# most once and only if address, element count and the version counter of the fp32 tensor still match (anything that rewrote or replaced
# the gradient in between - an accumulation, a hook, the token scatter at a pruning stage - makes the lookup miss, and the consumer
# converts the fp32 gradient itself).
_SHADOW = {}
shadow_hits = 0                           # diagnostic: how many gradients found their bf16 copy (tests assert the path is taken)


def shadow_put(t, t16):
    _SHADOW.clear()                       # at most one gradient is in flight between two blocks
    # the entry holds the fp32 tensor itself: while it is parked here its address cannot be recycled for another tensor with the same
    # element count and version 0 (a hook that replaces the gradient out of place would otherwise be able to alias it)
    _SHADOW[t.data_ptr()] = (t.numel(), t._version, t16, t)


def shadow_take(t):
    e = _SHADOW.pop(t.data_ptr(), None)
    _SHADOW.clear()
    if e is None or e[0] != t.numel() or e[1] != t._version or e[3].untyped_storage().data_ptr() != t.untyped_storage().data_ptr():
        return None
    global shadow_hits
    shadow_hits += 1
    return e[2]


# bf16 weights.  A weight's bf16 form is made once and reused: for the weights of a parameter arena (the trained model) d2s.engine converts the
# whole arena - and the arena of W^T copies - in one launch each at the start of every step (Bf16Weights.refresh); a weight outside any
# arena that does not require gradients (the frozen teacher) is converted on first use and again only when its version counter moves.
# Anything else (a free-standing trainable tensor) has no safe invalidation signal and is converted inside each GEMM call as before.
_W16 = {}                       # (data_ptr, transposed) -> (weakref to the weight or None for arena entries, epoch, version, shape, bf16 tensor)
_BF16_WEIGHTS = os.environ.get("D2S_BF16_WEIGHT_CACHE", "1") != "0"


def bf16_weight(W, transposed=False):
    """bf16 [N][K] form of the B operand of y = x W^T (transposed=False: W itself) or of dx = dy W (transposed=True: W^T), or None."""
    if not (_BF16_WEIGHTS and _BF16_IO and W.is_cuda and W.dim() == 2 and W.shape[0 if transposed else 1] % 32 == 0):
        return None
    ent = _W16.get((W.data_ptr(), transposed))
    shape = tuple(W.shape)
    if ent is not None and ent[2] == W._version and ent[3] == shape:
        if ent[0] is None:
            if ent[1] == weights_epoch:
                return ent[4]                       # arena weight, converted at the start of this step
        elif ent[0]() is W:
            return ent[4]                           # frozen weight, unchanged since its conversion
    if transposed or W.requires_grad or _in_weight_arena(W.data_ptr()):
        return None
    import weakref
    t16 = torch.empty(shape, dtype=torch.bfloat16, device=W.device)
    lib.call("d2s_convert_bf16", lib.ptr(W), lib.ptr(t16), W.numel())
    torch.cuda.current_stream().synchronize()       # once per frozen weight: later calls may read the copy from any stream
    if len(_W16) > 4096:
        _W16.clear()
    _W16[(W.data_ptr(), False)] = (weakref.ref(W), -1, W._version, shape, t16)
    return t16


def invalidate_bf16_weights():
    """Drop every cached bf16 form of a frozen weight.  The cache follows a weight's version counter; writes that bypass it
    (`teacher.weight.data.copy_()`, `dist.broadcast(p.data)`, raw-pointer kernels - `.data` has a version counter of its own) must be
    followed by this call.  vit_models' checkpoint loaders call it; arena weights are refreshed every step and need nothing."""
    _W16.clear()


class Bf16Weights:
    """bf16 mirrors of a parameter arena and of its W^T arena (TransposedArena), each refreshed by ONE launch per step."""

    def __init__(self, arena_params, weights, transposed_arena):
        self.src, self.tsrc = arena_params, transposed_arena.dst
        self.dst = torch.empty(arena_params.numel(), dtype=torch.bfloat16, device=arena_params.device)
        self.tdst = torch.empty(self.tsrc.numel(), dtype=torch.bfloat16, device=arena_params.device)
        self.entries = []
        toff = 0
        for w, off in weights:
            if w.dim() != 2:
                continue
            R, C = w.shape
            self.entries.append((w, self.dst[off:off + R * C].view(R, C), self.tdst[toff:toff + R * C].view(C, R)))
            toff += R * C

    def refresh(self):
        """call after TransposedArena.refresh() of the same step"""
        lib.call("d2s_convert_bf16", lib.ptr(self.src), lib.ptr(self.dst), self.src.numel())
        lib.call("d2s_convert_bf16", lib.ptr(self.tsrc), lib.ptr(self.tdst), self.tsrc.numel())
        for w, w16, wt16 in self.entries:
            shape = tuple(w.shape)
            if w16.data_ptr() % 16 == 0 and wt16.data_ptr() % 16 == 0:
                _W16[(w.data_ptr(), False)] = (None, weights_epoch, w._version, shape, w16)
                _W16[(w.data_ptr(), True)] = (None, weights_epoch, w._version, shape, wt16)


def gemm(layout, A, lda, B, ldb, C, ldc, M, N, K, epi=EPI_NONE, bias=None, aux=None, ldaux=0, aux_out=None, aux_rows=0,
         remap_rows=0, remap_skip=0, accumulate=False, a16=None, c16=None, b16=None):
    mode = get_gemm_mode()
    qk = (layout, M, N, K, mode)
    need = _WS_NEED.get(qk)
    if need is None:
        need = _WS_NEED[qk] = lib.query("d2s_gemm_f32_workspace_bytes", layout, M, N, K, mode)
    dev = C.device if C is not None else c16.device
    ws = workspace(need, dev) if need else None
    z16 = (aux_out if epi == EPI_BIAS_GELU else aux if epi == EPI_MUL_GELU_GRAD else None)
    if z16 is not None and z16.dtype == torch.bfloat16:      # bf16 data path: the GELU pre-activation is kept in bf16 (as under autocast)
        assert a16 is not None or c16 is not None or b16 is not None, "a bf16 pre-activation exists on the bf16 data path only"
        epi = EPI_BIAS_GELU_Z16 if epi == EPI_BIAS_GELU else EPI_MUL_GELU_GRAD_Z16
    if a16 is not None or c16 is not None or b16 is not None:
        assert mode == GEMM_BF16 and not accumulate and remap_rows == 0 and aux_rows == 0
        assert b16 is None or (b16.dtype == torch.bfloat16 and b16.is_contiguous() and tuple(b16.shape) == (N, K)), "b16 must be dense bf16 [N, K]"
        assert a16 is None or (a16.dtype == torch.bfloat16 and a16.is_contiguous() and tuple(a16.shape) == (M, K)), "a16 must be dense bf16 [M, K]"
        assert c16 is None or (c16.dtype == torch.bfloat16 and c16.is_contiguous() and tuple(c16.shape) == (M, N)), "c16 must be dense bf16 [M, N]"
        lib.call("d2s_gemm_f32_bf16io", layout, lib.ptr(A), lda, lib.ptr(B), ldb, lib.ptr(C), ldc, M, N, K, epi, lib.ptr(bias),
                 lib.ptr(aux), ldaux, lib.ptr(aux_out), lib.ptr(a16), lib.ptr(b16), lib.ptr(c16), lib.ptr(ws), ws.numel() if ws is not None else 0)
        return C
    lib.call("d2s_gemm_f32", layout, lib.ptr(A), lda, lib.ptr(B), ldb, lib.ptr(C), ldc, M, N, K, epi, lib.ptr(bias),
             lib.ptr(aux), ldaux, lib.ptr(aux_out), aux_rows, remap_rows, remap_skip, int(accumulate), mode, lib.ptr(ws),
             ws.numel() if ws is not None else 0)
    return C


def linear_fwd(x, W, bias=None, epi=None, aux=None, aux_out=None, out=None, a16=None, c16=None, want_f32=True):
    """y[M,N] = epi(x[M,K] @ W[N,K]^T + bias).  nn.Linear forward (F.linear).
    bf16 mode only: a16 = bf16 copy of x (x itself may then be None), c16 = bf16 [M, N] buffer that receives a copy of y;
    want_f32=False (needs c16) skips the fp32 result and returns None."""
    _f32(W)
    if x is not None:
        _f32(x)
    M, K = (x if x is not None else a16).shape
    N = W.shape[0]
    if out is None and want_f32:
        out = torch.empty((M, N), dtype=torch.float32, device=W.device)
    if epi is None:
        epi = EPI_BIAS if bias is not None else EPI_NONE
    b16 = bf16_weight(W) if get_gemm_mode() == GEMM_BF16 else None
    return gemm(NT, x, K, W, K, out, N, M, N, K, epi, bias, aux, N if aux is not None else 0, aux_out, a16=a16, c16=c16, b16=b16)


# ---- k-contiguous copies W^T of Linear weights for the input-gradient GEMM ----
# dx = dy W is an NN product (W's reduction index is its row index); with W^T [n_in, n_out] at hand it is an NT product, both operands
# k-contiguous, which the fp32 kernel runs 10-15 % faster on the student's shapes (profiles/r02_e_*).  A weight changes once per
# optimiser step, so its transpose is rebuilt lazily on the first input-gradient call after a change and reused until the next one:
# `weights_epoch` is bumped by everything that writes parameters behind torch's back (the fused AdamW kernel); in-place torch writes
# (load_state_dict, torch optimisers) show up in the tensor's version counter.
# Only weights that live in a registered parameter arena (d2s.engine.ParamArena) take this path: their addresses are stable and unique for
# the arena's lifetime, whereas the address of a free-standing tensor can be recycled for another tensor with the same shape and version.
weights_epoch = 0
_WT = {}                        # data_ptr -> (epoch, version, shape, W^T tensor)
_WT_ARENAS = []                 # (weakref to the arena tensor, first byte, one past the last byte)


def register_weight_arena(arena_tensor):
    import weakref
    _WT_ARENAS[:] = [a for a in _WT_ARENAS if a[0]() is not None]
    lo = arena_tensor.data_ptr()
    hi = lo + arena_tensor.numel() * 4
    for key in [k for k in _WT if lo <= k < hi]:       # an earlier arena's entries at recycled addresses
        del _WT[key]
    _WT_ARENAS.append((weakref.ref(arena_tensor), lo, hi))


def _in_weight_arena(ptr):
    for ref, lo, hi in _WT_ARENAS:
        if lo <= ptr < hi and ref() is not None:
            return True
    return False
_DGRAD_NT_MIN_ROWS = int(os.environ.get("D2S_DGRAD_NT_MIN_ROWS", "1024"))      # arena weights get their W^T copies once per step anyway (TransposedArena); at 3168 rows the NT form is still 5-10 % faster (profiles/r03_d_config3_tile_sweep.txt)


def bump_weights_epoch():
    global weights_epoch
    weights_epoch += 1


def transposed_weight(W):
    key = W.data_ptr()
    ent = _WT.get(key)
    if ent is not None and ent[0] == weights_epoch and ent[1] == W._version and ent[2] == tuple(W.shape):
        return ent[3]
    Wt = ent[3] if (ent is not None and ent[2] == tuple(W.shape)) else torch.empty((W.shape[1], W.shape[0]), dtype=torch.float32, device=W.device)
    lib.call("d2s_transpose_f32", lib.ptr(W), lib.ptr(Wt), W.shape[0], W.shape[1])
    if len(_WT) > 4096:         # models come and go in a test process: do not let stale entries pile up
        _WT.clear()
    _WT[key] = (weights_epoch, W._version, tuple(W.shape), Wt)
    return Wt


class TransposedArena:
    """W^T copies of every 2-D weight of a parameter arena, rebuilt by ONE kernel launch per optimiser step (d2s.engine.FusedAdamW calls
    refresh() right after the update) instead of one small launch per weight on its first input-gradient use."""

    def __init__(self, arena_params, weights):
        """weights: list of (Parameter living in arena_params, its arena offset in floats); only 2-D ones are kept.  The Parameter objects
        (not .data aliases) are held so that refresh() records the version counter the autograd Functions will see."""
        import numpy as np
        self.src = arena_params
        mats = [(w, off) for w, off in weights if w.dim() == 2]
        total = sum(w.numel() for w, _ in mats)
        self.dst = torch.empty(max(total, 1), dtype=torch.float32, device=arena_params.device)
        rows, self.entries, doff = [], [], 0
        for w, off in mats:
            R, C = w.shape
            self.entries.append((w, self.dst[doff:doff + R * C].view(C, R)))
            for r0 in range(0, R, 64):
                for c0 in range(0, C, 64):
                    rows.append((off, doff, R, C, r0, c0))
            doff += R * C
        desc = np.zeros(len(rows), dtype=[("s", "<i8"), ("d", "<i8"), ("R", "<i4"), ("C", "<i4"), ("r0", "<i4"), ("c0", "<i4")])
        for i, r in enumerate(rows):
            desc[i] = r
        self.n_tiles = len(rows)
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(arena_params.device) if rows else None

    def refresh(self):
        if not self.n_tiles:
            return
        lib.call("d2s_transpose_batched_f32", lib.ptr(self.src), lib.ptr(self.dst), lib.ptr(self.desc), self.n_tiles)
        for w, wt in self.entries:
            _WT[w.data_ptr()] = (weights_epoch, w._version, tuple(w.shape), wt)


def linear_dgrad(dy, W, epi=EPI_NONE, aux=None, out=None, a16=None, c16=None, want_f32=True):
    """dx[M,K] = epi(dy[M,N] @ W[N,K]).  bf16 mode only: a16 = bf16 copy of dy (dy may then be None), c16 = bf16 [M, K] buffer receiving a
    copy of dx; want_f32=False (needs c16) skips the fp32 result and returns None."""
    _f32(W)
    if dy is not None:
        _f32(dy)
    M, N = (dy if dy is not None else a16).shape
    K = W.shape[1]
    if out is None and want_f32:
        out = torch.empty((M, K), dtype=torch.float32, device=W.device)
    if get_gemm_mode() == GEMM_BF16:
        wt16 = bf16_weight(W, transposed=True)      # [K][N] = W^T in bf16: the k-contiguous B operand of dx = dy W
        if a16 is not None or c16 is not None or wt16 is not None:
            return gemm(NN, dy, N, W, K, out, K, M, K, N, epi, None, aux, K if aux is not None else 0, a16=a16, c16=c16, b16=wt16)
    if get_gemm_mode() == GEMM_EXACT and M >= _DGRAD_NT_MIN_ROWS and (N % 16 == 0) and (K % 4 == 0) and _in_weight_arena(W.data_ptr()):
        return gemm(NT, dy, N, transposed_weight(W), N, out, K, M, K, N, epi, None, aux, K if aux is not None else 0)
    return gemm(NN, dy, N, W, K, out, K, M, K, N, epi, None, aux, K if aux is not None else 0)


def linear_wgrad(dy, x, dW, accumulate=False, db=None, x16=None, dy16=None):
    """dW[N,K] (+)= dy[M,N]^T @ x[M,K]; with db also db[N] (+)= dy.sum(0), folded into the same pass over dy.
    bf16 mode: x16 = the layer input in bf16 (x may then be None - the bf16 data path saves only that form); dy16 = the gradient in bf16
    where its producer wrote only that (dy may then be None; needs x16)."""
    _f32(dW)
    if dy is not None:
        _f32(dy)
    M, N = (dy if dy is not None else dy16).shape
    K = (x if x is not None else x16).shape[1]
    mode = get_gemm_mode()
    need = lib.query("d2s_linear_wgrad_workspace_bytes", M, N, K, mode)
    ws = workspace(need, dW.device) if need else None
    if x16 is not None:
        assert mode == GEMM_BF16 and x16.dtype == torch.bfloat16 and x16.is_contiguous() and tuple(x16.shape) == (M, K)
        assert dy16 is None or (dy16.dtype == torch.bfloat16 and dy16.is_contiguous() and tuple(dy16.shape) == (M, N))
        lib.call("d2s_linear_wgrad_f32_bf16x", lib.ptr(dy), lib.ptr(dy16), N, lib.ptr(x16), K, lib.ptr(dW), K, lib.ptr(db), M, N, K,
                 int(accumulate), lib.ptr(ws), ws.numel() if ws is not None else 0)
        return dW
    assert dy16 is None, "a bf16 gradient needs the bf16 layer input as well"
    _f32(x)
    lib.call("d2s_linear_wgrad_f32", lib.ptr(dy), N, lib.ptr(x), K, lib.ptr(dW), K, lib.ptr(db), M, N, K, int(accumulate), mode,
             lib.ptr(ws), ws.numel() if ws is not None else 0)
    return dW


# ---- weight gradients on their own stream ----
# Within a backward pass nothing waits for a weight gradient: dW / db of a Linear are only read by the optimiser (and the gradient
# all-reduce), while the chain that the next layer waits for is dy -> dx.  Inside `async_weight_grads()` (d2s.engine.TrainStep wraps its
# backward in it) the wgrad GEMMs, their slab combines and the bias column sums are issued on a second HIP stream that trails the main
# one, so they fill the CUs that the dgrad / attention kernels of the following layers leave idle in their tails; the caller joins the
# stream before anything reads the gradients (join_weight_grads).  Outside that block - any caller that runs loss.backward() itself and
# then reads .grad on the current stream - everything stays on the current stream.
# HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4), and two streams that share a queue run
# their kernels strictly one after the other.  Which queue a stream gets depends on what else created streams before it: with an RCCL
# process group alive the teacher stream of d2s.engine landed on the main stream's queue and the teacher / student overlap was gone
# (-5 % on the step, profiles/r03_o_stream_queue_collision.txt).  So a side stream is CHOSEN: candidates from torch's pool are kept only if a
# pair of spin kernels shows that they really run beside every stream they are meant to overlap with.
_STREAM_CHECK = os.environ.get("D2S_STREAM_CHECK", "1") != "0"
stream_picks = []          # diagnostic: (purpose, candidates tried, verified concurrent)


def _streams_overlap(a, b, cycles=3000000):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for st in (a, b):                         # first use of a stream sets things up lazily: keep that out of the measurement
        with torch.cuda.stream(st):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        ev[0].record()
        torch.cuda._sleep(cycles)
        ev[1].record()
    with torch.cuda.stream(b):
        ev[2].record()
        torch.cuda._sleep(cycles)
        ev[3].record()
    torch.cuda.synchronize()
    single = ev[0].elapsed_time(ev[1])
    return ev[0].elapsed_time(ev[3]) < 1.6 * single      # one queue: the second spin starts when the first has ended (2x)


def concurrent_stream(beside, purpose="", tries=12):
    """A torch.cuda.Stream whose kernels run concurrently with those of every stream in `beside` (verified on the device), or - if no
    candidate passes - the last candidate (the step is then correct but loses that overlap; stream_picks records it)."""
    cand = torch.cuda.Stream()
    if not _STREAM_CHECK:
        return cand
    kept = []                  # hold the rejected candidates while searching: the pool hands out its 32 streams round-robin
    for n in range(1, tries + 1):
        if all(_streams_overlap(o, cand) for o in beside):
            stream_picks.append((purpose, n, True))
            return cand
        kept.append(cand)
        cand = torch.cuda.Stream()
    stream_picks.append((purpose, tries, False))
    return cand


def recheck_stream(stream, beside, purpose):
    """`stream` if it still runs beside every stream in `beside`, else a replacement that does.  The queue a stream sits on can end up
    shared AFTER it was picked (an RCCL communicator sets itself up at its first collective): d2s.engine re-checks its side streams once
    the first steps have run."""
    if stream is None or not _STREAM_CHECK or all(_streams_overlap(o, stream) for o in beside):
        return stream
    return concurrent_stream(beside, purpose + " (re-picked)")


def pg_stream_shares_queue(stream, group=None, cycles=20000000):
    """Does the process group's RCCL stream sit on the hardware queue of `stream`?  (torch picks that stream from its pool at the
    group's first collective; nothing lets a caller choose it.)  If it does, every collective - which first waits for the gradient
    streams - holds up whatever `stream` issues after it: measured -17 % on the step.  Probe: a spin on a helper stream, a tiny
    all_reduce issued behind it (RCCL's stream now waits for the spin), then a spin on `stream` - which ends late only if it had to
    queue behind that wait.  The spins are long (~10 ms) so that the host time of the all_reduce call between the two launches cannot be
    mistaken for queueing.  Two collectives per call: every rank of the group must make the same calls."""
    import torch.distributed as dist
    helper = concurrent_stream([stream], "probe helper")
    t = torch.zeros(64, device=torch.device("cuda", torch.cuda.current_device()))
    with torch.cuda.stream(helper):                   # the group's stream exists (and is set up) after its first collective
        dist.all_reduce(t, group=group, async_op=True).wait()
        torch.cuda._sleep(1000)
    with torch.cuda.stream(stream):
        torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.cuda.stream(helper):
        ev[0].record()
        torch.cuda._sleep(cycles)
        ev[1].record()
        w = dist.all_reduce(t, group=group, async_op=True)
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cycles)
        ev[2].record()
    with torch.cuda.stream(helper):
        w.wait()
    torch.cuda.synchronize()
    if os.environ.get("D2S_STREAM_DEBUG") == "1":
        import sys
        print(f"[d2s] rccl-queue probe: helper spin {ev[0].elapsed_time(ev[1]):.2f} ms, start -> end of the probed stream's spin {ev[0].elapsed_time(ev[2]):.2f} ms",
              file=sys.stderr, flush=True)
    return ev[0].elapsed_time(ev[2]) > 1.6 * ev[0].elapsed_time(ev[1])


def ensure_weight_grad_stream(beside):
    """The weight-gradient stream, created now (instead of at the first backward) if it does not exist yet."""
    if _WGRAD["stream"] is None:
        _WGRAD["stream"] = concurrent_stream(list(beside), "weight gradients")
    return _WGRAD["stream"]


def set_weight_grad_stream(stream):
    assert not _WGRAD["on"]
    _WGRAD["stream"] = stream


def recheck_weight_grad_stream(beside):
    if _WGRAD["stream"] is not None and not _WGRAD["on"]:
        _WGRAD["stream"] = recheck_stream(_WGRAD["stream"], beside, "weight gradients")


_WGRAD = {"on": False, "stream": None, "used": False}
_WGRAD_ENABLED = os.environ.get("D2S_WGRAD_STREAM", "1") != "0"
_ATTN_BWD_STREAMS = os.environ.get("D2S_ATTN_BWD_STREAMS", "0") == "1"      # dQ and dK/dV kernels of the attention backward side by side:
# measured +-0 on top of the weight-gradient stream (3503 / 3502 images/s), so off by default


class async_weight_grads:
    def __enter__(self):
        if _WGRAD_ENABLED and torch.cuda.is_available():
            if _WGRAD["stream"] is None:
                _WGRAD["stream"] = concurrent_stream([torch.cuda.current_stream()] + list(_WGRAD.get("beside", [])), "weight gradients")
            _WGRAD["on"], _WGRAD["used"] = True, False
        return self

    def __exit__(self, *exc):
        _WGRAD["on"] = False
        join_weight_grads()
        return False


def weight_grad_stream():
    """The side stream if weight gradients were issued on it since the last join, else None (the reducer of the data-parallel path
    makes its own stream wait on it before it all-reduces a bucket)."""
    return _WGRAD["stream"] if _WGRAD["used"] else None


def join_weight_grads():
    if _WGRAD["used"] and _WGRAD["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_WGRAD["stream"])
    _WGRAD["used"] = False


def linear_param_grads(dy, x, W, b, want_w=True, want_b=True, accumulate=False, x16=None, dy16=None):
    """(dW, db) of a Linear into fresh arena-backed buffers: one fused pass when both are wanted.  x16 / dy16: see linear_wgrad."""
    dW = grad_buffer(W) if want_w else None
    db = grad_buffer(b) if (want_b and b is not None) else None
    if dW is None and db is None:
        return dW, db
    src = dy if dy is not None else dy16
    if dW is None and dy is None:
        dy = dy16.float()            # bias gradient alone from a bf16-only gradient (not on the model's path: weights and biases train together)
    side = _WGRAD["stream"] if (_WGRAD["on"] and src.is_cuda) else None
    if side is None:
        if dW is not None:
            linear_wgrad(dy, x, dW, accumulate=accumulate, db=db, x16=x16, dy16=dy16)
        else:
            colsum(dy, db, accumulate=accumulate)
        return dW, db
    main = torch.cuda.current_stream()
    side.wait_stream(main)                       # dy (and x) were produced on the main stream
    with torch.cuda.stream(side):
        if dW is not None:
            linear_wgrad(dy, x, dW, accumulate=accumulate, db=db, x16=x16, dy16=dy16)
        else:
            colsum(dy, db, accumulate=accumulate)
    for t in (dy, dy16, x, x16, dW, db):                    # autograd may free these on the main stream while the side stream still reads / writes them
        if t is not None:
            t.record_stream(side)
    _WGRAD["used"] = True
    return dW, db


def colsum(x, out, accumulate=False):
    _f32(x)
    M, N = x.shape
    need = lib.query("d2s_colsum_workspace_bytes", M, N)
    ws = workspace(need, x.device)
    lib.call("d2s_colsum_f32", lib.ptr(x), N, M, N, lib.ptr(out), int(accumulate), lib.ptr(ws), ws.numel())
    return out


def contiguous_map(rows, D):
    return (rows, 0, D, 0)


def skip_cls_map(n, D):
    """rows of x[:, 1:] inside a contiguous [B, n, D] buffer."""
    return (n - 1, n * D, D, D)


def layernorm_fwd(x, rowmap, w, b, rows, D, eps, stats=True):
    """stats=False (forward-only callers: the frozen teacher, eval): the per-row mean / rstd are not written."""
    y = torch.empty((rows, D), dtype=torch.float32, device=x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if stats else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if stats else None
    lib.call("d2s_layernorm_fwd", lib.ptr(x), *rowmap, lib.ptr(w), lib.ptr(b), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd),
             rows, D, float(eps))
    return y, mean, rstd


def layernorm_fwd_bf16(x, rowmap, w, b, rows, D, eps, stats=True, want_f32=True):
    """LayerNorm forward that also (or only: want_f32=False) writes the bf16 copy a bf16-mode GEMM reads.  -> (y or None, mean, rstd, y16)"""
    y = torch.empty((rows, D), dtype=torch.float32, device=x.device) if want_f32 else None
    y16 = bf16_buffer(rows, D, x.device)
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if stats else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if stats else None
    lib.call("d2s_layernorm_fwd_bf16out", lib.ptr(x), *rowmap, lib.ptr(w), lib.ptr(b), lib.ptr(y), lib.ptr(y16), lib.ptr(mean),
             lib.ptr(rstd), rows, D, float(eps))
    return y, mean, rstd, y16


def layernorm_bwd(x, rowmap, dy, w, mean, rstd, dx, add_src, dw, db, rows, D, accumulate_wb=False, relu_mask=False, dx16=None):
    """dx16 (bf16 mode): dense bf16 [rows, D] buffer that receives a copy of dx."""
    need = lib.query("d2s_layernorm_bwd_workspace_bytes", rows, D)
    ws = workspace(need, dy.device)
    if dx16 is not None:
        assert dx16.dtype == torch.bfloat16 and dx16.is_contiguous() and tuple(dx16.shape) == (rows, D)
        lib.call("d2s_layernorm_bwd_bf16out", lib.ptr(x), *rowmap, lib.ptr(dy), lib.ptr(w), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dx),
                 lib.ptr(dx16), lib.ptr(add_src), lib.ptr(dw), lib.ptr(db), int(accumulate_wb), int(relu_mask), rows, D, lib.ptr(ws), ws.numel())
        return dx
    lib.call("d2s_layernorm_bwd", lib.ptr(x), *rowmap, lib.ptr(dy), lib.ptr(w), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dx),
             lib.ptr(add_src), lib.ptr(dw), lib.ptr(db), int(accumulate_wb), int(relu_mask), rows, D, lib.ptr(ws), ws.numel())
    return dx


def batchnorm_fwd(x, w, b, running_mean, running_var, training, eps=1e-5, momentum=0.1):
    """BatchNorm1d over the rows of x [R, C] (dynamic_vit.py:350-367).  Returns (y, mean, rstd); updates the running estimates in
    place when training."""
    _f32(x)
    R, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = workspace(lib.query("d2s_batchnorm_workspace_bytes", R, C), x.device)
    lib.call("d2s_batchnorm_fwd", lib.ptr(x), lib.ptr(w), lib.ptr(b), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd), lib.ptr(running_mean),
             lib.ptr(running_var), R, C, float(eps), float(momentum), int(bool(training)), lib.ptr(ws), ws.numel())
    return y, mean, rstd


def batchnorm_bwd(x, dy, w, mean, rstd, dw, db, training, relu_mask=False, accumulate=False):
    _f32(x), _f32(dy)
    R, C = x.shape
    dx = torch.empty_like(x)
    ws = workspace(lib.query("d2s_batchnorm_workspace_bytes", R, C), x.device)
    lib.call("d2s_batchnorm_bwd", lib.ptr(x), lib.ptr(dy), lib.ptr(w), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dx), lib.ptr(dw), lib.ptr(db),
             int(relu_mask), int(accumulate), int(bool(training)), R, C, lib.ptr(ws), ws.numel())
    return dx


def softmax_rows(scores):
    _f32(scores)
    R, T = scores.shape
    probs = torch.empty_like(scores)
    lib.call("d2s_softmax_rows", lib.ptr(scores), lib.ptr(probs), R, T)
    return probs


def select_topk(probs, k):
    _f32(probs)
    B, T = probs.shape
    k = min(int(k), T)
    kept = torch.empty((B, k), dtype=torch.int64, device=probs.device)
    dropped = torch.empty((B, T - k), dtype=torch.int64, device=probs.device)
    lib.call("d2s_select_topk", lib.ptr(probs), B, T, k, lib.ptr(kept), lib.ptr(dropped) if T - k > 0 else None)
    return kept, dropped


def gather_pack(x, ids):
    _f32(x)
    B, n, D = x.shape
    k = ids.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    out = torch.empty((B, k + 1, D), dtype=torch.float32, device=x.device)
    lib.call("d2s_gather_pack_fwd", lib.ptr(x), lib.ptr(ids), lib.ptr(out), B, n, k, D)
    return out


def scatter_unpack(g, ids, n):
    _f32(g)
    B, k1, D = g.shape
    dx = torch.empty((B, n, D), dtype=torch.float32, device=g.device)
    lib.call("d2s_scatter_unpack_bwd", lib.ptr(g), lib.ptr(ids), lib.ptr(dx), B, n, k1 - 1, D)
    return dx


def half_mean_concat(x, B, T, C, relu_mask_src=None):
    out = torch.empty((B * T, C), dtype=torch.float32, device=x.device)
    lib.call("d2s_half_mean_concat", lib.ptr(x), lib.ptr(relu_mask_src), lib.ptr(out), B, T, C)
    return out


def im2col_patch(img, P):
    _f32(img)
    B, Cin, H, W = img.shape
    T = (H // P) * (W // P)
    col = torch.empty((B * T, Cin * P * P), dtype=torch.float32, device=img.device)
    lib.call("d2s_im2col_patch", lib.ptr(img), lib.ptr(col), B, Cin, H, W, P)
    return col


def fill_cls(cls, pos, tokens):
    B, n, D = tokens.shape
    lib.call("d2s_fill_cls", lib.ptr(cls), lib.ptr(pos), lib.ptr(tokens), B, n, D)


def batch_sum(g, out, B, count, image_stride, accumulate=False):
    lib.call("d2s_batch_sum", lib.ptr(g), lib.ptr(out), B, count, image_stride, int(accumulate))
    return out


def copy_rows(src, rowmap, rows, D, dst=None, dst_map=None, accumulate=False):
    if dst is None:
        dst = torch.empty((rows, D), dtype=torch.float32, device=src.device)
    if dst_map is None:
        dst_map = contiguous_map(rows, D)
    lib.call("d2s_copy_rows", lib.ptr(src), *rowmap, lib.ptr(dst), *dst_map, rows, D, int(accumulate))
    return dst


def unfold_fwd(src, strides, B, C, H, W, k, s, p):
    """strides = (sb, sc, sy, sx) of the source seen as [B, C, H, W]; -> [B, Ho*Wo, C*k*k]"""
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((B, Ho * Wo, C * k * k), dtype=torch.float32, device=src.device)
    lib.call("d2s_unfold_fwd", lib.ptr(src), *strides, lib.ptr(out), B, C, H, W, k, s, p)
    return out


def unfold_bwd(g, dsrc, strides, B, C, H, W, k, s, p):
    lib.call("d2s_unfold_bwd", lib.ptr(g), lib.ptr(dsrc), *strides, B, C, H, W, k, s, p)
    return dsrc


def performer_attn_fwd(kqv, w, B, T, eps=1e-8):
    dev = kqv.device
    M = B * T
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    y, kp, qp, A, ksum, D = f(M, 64), f(M, 32), f(M, 32), f(B, 64, 32), f(B, 32), f(M)
    ws = workspace(lib.query("d2s_performer_workspace_bytes", B, T), dev)
    lib.call("d2s_performer_attn_fwd", lib.ptr(kqv), lib.ptr(w), lib.ptr(y), lib.ptr(kp), lib.ptr(qp), lib.ptr(A), lib.ptr(ksum),
             lib.ptr(D), B, T, float(eps), lib.ptr(ws), ws.numel())
    return y, kp, qp, A, ksum, D


def performer_attn_bwd(kqv, w, y, kp, qp, A, ksum, D, gy, skip, B, T, eps=1e-8):
    dev = kqv.device
    M = B * T
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    dkqv, dnum, dD, dqp, dkp, dA, dksum = f(M, 192), f(M, 64), f(M), f(M, 32), f(M, 32), f(B, 64, 32), f(B, 32)
    ws = workspace(lib.query("d2s_performer_workspace_bytes", B, T), dev)
    lib.call("d2s_performer_attn_bwd", lib.ptr(kqv), lib.ptr(w), lib.ptr(y), lib.ptr(kp), lib.ptr(qp), lib.ptr(A), lib.ptr(ksum),
             lib.ptr(D), lib.ptr(gy), lib.ptr(skip), lib.ptr(dkqv), lib.ptr(dnum), lib.ptr(dD), lib.ptr(dqp), lib.ptr(dkp), lib.ptr(dA),
             lib.ptr(dksum), B, T, float(eps), lib.ptr(ws), ws.numel())
    return dkqv


def attn_fwd(qkv, B, n, H, scale, want_cls=True):
    out = torch.empty((B * n, H * 64), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    cls_row = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device) if want_cls else None
    # bf16 arithmetic mode: the two matrix products of the forward run on the bf16 matrix cores too (same outputs, fp32 backward)
    entry = "d2s_attn_fwd_bf16" if (get_gemm_mode() == GEMM_BF16 and _BF16_ATTENTION) else "d2s_attn_fwd_f32"
    lib.call(entry, lib.ptr(qkv), lib.ptr(out), lib.ptr(lse), lib.ptr(cls_row), B, n, H, float(scale))
    return out, lse, cls_row


def attn_fwd_bf16io(qkv, B, n, H, scale, want_cls=True, want_f32=True):
    """bf16-mode attention forward that also (or only) writes the bf16 copy of its output; qkv fp32 or bf16 (the qkv GEMM's c16).
    -> (out or None, lse, cls_row, out16)"""
    assert qkv.is_contiguous() and qkv.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty((B * n, H * 64), dtype=torch.float32, device=qkv.device) if want_f32 else None
    out16 = bf16_buffer(B * n, H * 64, qkv.device)
    lse = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    cls_row = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device) if want_cls else None
    lib.call("d2s_attn_fwd_bf16_bf16out", lib.ptr(qkv), int(qkv.dtype == torch.bfloat16), lib.ptr(out), lib.ptr(out16), lib.ptr(lse),
             lib.ptr(cls_row), B, n, H, float(scale))
    return out, lse, cls_row, out16


def attn_bwd(qkv, out, dout, lse, B, n, H, scale, dqkv16=None, want_f32=True):
    """dqkv16 (bf16 mode with the bf16 attention kernels): bf16 buffer shaped like qkv that receives a copy of dqkv; want_f32=False
    (needs dqkv16): only that form is written and None is returned."""
    dqkv = torch.empty(qkv.shape, dtype=torch.float32, device=qkv.device) if (want_f32 or dqkv16 is None) else None
    delta = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    bf16 = get_gemm_mode() == GEMM_BF16 and _BF16_ATTENTION
    if dqkv16 is not None:
        assert bf16 and dqkv16.dtype == torch.bfloat16 and dqkv16.is_contiguous() and dqkv16.numel() == qkv.numel()
        lib.call("d2s_attn_bwd_bf16_bf16out", lib.ptr(qkv), int(qkv.dtype == torch.bfloat16), lib.ptr(out), lib.ptr(dout), lib.ptr(lse),
                 lib.ptr(dqkv), lib.ptr(dqkv16), lib.ptr(delta), B, n, H, float(scale))
        return dqkv
    assert qkv.dtype == torch.float32
    if _WGRAD["on"] and _ATTN_BWD_STREAMS and not bf16 and qkv.is_cuda:
        # inside TrainStep's backward: the dK/dV kernel on a stream of its own beside the dQ kernel (independent, disjoint outputs);
        # both are joined again before anything reads dqkv
        if _WGRAD.get("attn_stream") is None:
            _WGRAD["attn_stream"] = torch.cuda.Stream()
        side, main = _WGRAD["attn_stream"], torch.cuda.current_stream()
        lib.call("d2s_attn_delta", lib.ptr(out), lib.ptr(dout), lib.ptr(delta), B, n, H)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            lib.call("d2s_attn_bwd_dkv_f32", lib.ptr(qkv), lib.ptr(dout), lib.ptr(lse), lib.ptr(delta), lib.ptr(dqkv), B, n, H, float(scale))
        lib.call("d2s_attn_bwd_dq_f32", lib.ptr(qkv), lib.ptr(dout), lib.ptr(lse), lib.ptr(delta), lib.ptr(dqkv), B, n, H, float(scale))
        main.wait_stream(side)
        return dqkv
    entry = "d2s_attn_bwd_bf16" if bf16 else "d2s_attn_bwd_f32"
    lib.call(entry, lib.ptr(qkv), lib.ptr(out), lib.ptr(dout), lib.ptr(lse), lib.ptr(dqkv), lib.ptr(delta), B, n, H, float(scale))
    return dqkv


# ---- one transformer block per C-ABI call (csrc/block.hip): the same launches as the per-op wrappers above, issued from C ----
_BLOCK_COMPOSITE = os.environ.get("D2S_BLOCK_COMPOSITE", "1") != "0"
_BLOCK_SIZES = {}       # (B, n, D, H, hidden, mode) -> (saved floats train, saved floats eval, bwd scratch floats, main ws bytes, wgrad ws bytes)
c_abi_calls_saved = 0   # diagnostic: per-op calls that the composite entries replaced


def block_composite_ok(x, heads, hidden):
    """fp32 data path (arithmetic modes 0 / 1) on dense [B, n, H * 64] tokens; the bf16 data path keeps its per-op sequence."""
    return _BLOCK_COMPOSITE and x.is_cuda and x.shape[2] == heads * _DH and get_gemm_mode() in (GEMM_EXACT, GEMM_SPLIT)


_DH = 64


def _block_sizes(B, n, D, H, hidden, mode):
    key = (B, n, D, H, hidden, mode)
    ent = _BLOCK_SIZES.get(key)
    if ent is None:
        ent = _BLOCK_SIZES[key] = (lib.query("d2s_block_saved_floats", B, n, D, H, hidden, 1), lib.query("d2s_block_saved_floats", B, n, D, H, hidden, 0),
                                   lib.query("d2s_block_bwd_scratch_floats", B, n, D, H, hidden), lib.query("d2s_block_workspace_bytes", B, n, D, hidden, mode),
                                   lib.query("d2s_block_wgrad_workspace_bytes", B, n, D, hidden, mode))
    return ent


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def block_fwd(x, params, B, n, D, H, hidden, eps, scale, want_cls, train):
    """-> (y [B,n,D], cls_row [B,H,n] or None, slab): slab holds what block_bwd needs when train, forward-only scratch otherwise."""
    mode = get_gemm_mode()
    sz = _block_sizes(B, n, D, H, hidden, mode)
    slab = torch.empty((sz[0] if train else sz[1],), dtype=torch.float32, device=x.device)
    y = torch.empty((B, n, D), dtype=torch.float32, device=x.device)
    cls_row = torch.empty((B, H, n), dtype=torch.float32, device=x.device) if want_cls else None
    ws = workspace(sz[3], x.device)
    lib.call("d2s_block_fwd_f32", lib.ptr(x), _ptr_array(params), B, n, D, H, hidden, float(eps), float(scale), lib.ptr(y), lib.ptr(cls_row),
             lib.ptr(slab), int(train), mode, lib.ptr(ws), ws.numel())
    return y, cls_row, slab


def block_bwd(gy, x, slab, params, B, n, D, H, hidden, scale, want_dx, dparams):
    """dparams: 12 gradient buffers (None = not wanted; LayerNorm weight / bias come as a pair).  -> dx [B,n,D] or None"""
    mode = get_gemm_mode()
    sz = _block_sizes(B, n, D, H, hidden, mode)
    dev = gy.device
    scratch = torch.empty((sz[2],), dtype=torch.float32, device=dev)
    dx = torch.empty((B, n, D), dtype=torch.float32, device=dev) if want_dx else None
    # input gradients through the cached k-contiguous W^T copies (exact mode, arena weights, enough rows: linear_dgrad's rule)
    wt = [None] * 4
    if mode == GEMM_EXACT and B * n >= _DGRAD_NT_MIN_ROWS and D % 16 == 0 and hidden % 16 == 0:
        for i, w in enumerate((params[2], params[4], params[8], params[10])):
            if _in_weight_arena(w.data_ptr()):
                wt[i] = transposed_weight(w)
    side = _WGRAD["stream"] if (_WGRAD["on"] and any(dparams[i] is not None for i in (2, 3, 4, 5, 8, 9, 10, 11))) else None
    ws_side = workspace_on(side, sz[4], dev) if side is not None else None
    ws = workspace(sz[3] if side is not None else max(sz[3], sz[4]), dev)      # in line, the weight gradients use the main scratch too
    lib.call("d2s_block_bwd_f32", lib.ptr(gy), lib.ptr(x), lib.ptr(slab), _ptr_array(params), _ptr_array(wt), B, n, D, H, hidden, float(scale),
             lib.ptr(dx), _ptr_array(dparams), lib.ptr(scratch), mode, lib.ptr(ws), ws.numel(), lib.ptr(ws_side),
             ws_side.numel() if ws_side is not None else 0, side.cuda_stream if side is not None else None)
    if side is not None:       # autograd may free these on the main stream while the side stream still reads / writes them
        for t in (gy, slab, scratch):
            t.record_stream(side)
        for t in dparams:
            if t is not None:
                t.record_stream(side)
        _WGRAD["used"] = True
    return dx


KL_LOGIT_TARGET, KL_PROB_TARGET, CE_LABEL, MSE_TARGET, SOFT_CE = 0, 1, 2, 3, 4


def teacher_target(cls_attn):
    B, L, H, n = cls_attn.shape
    out = torch.empty((B, n - 1), dtype=torch.float32, device=cls_attn.device)
    lib.call("d2s_teacher_target", lib.ptr(_f32(cls_attn)), lib.ptr(out), B, L, H, n)
    return out


def gather_renorm(target, ids, normalize=True):
    B, T = target.shape
    k = ids.shape[1]
    out = torch.empty((B, k), dtype=torch.float32, device=target.device)
    lib.call("d2s_gather_renorm", lib.ptr(_f32(target)), lib.ptr(ids), lib.ptr(out), B, T, k, int(normalize))
    return out


def kl_rows(s, s_map, rows, C, mode, t=None, t_map=(1, 0, 0, 0), t_ids=None, labels=None, want_grad=True, row_weight=None):
    """row_weight [rows]: every row's loss and gradient are scaled by it (mask / count restricts a mean to the kept tokens)."""
    loss_row = torch.empty((rows,), dtype=torch.float32, device=s.device)
    grad = torch.empty((rows, C), dtype=torch.float32, device=s.device) if want_grad else None
    lib.call("d2s_kl_rows", lib.ptr(s), *s_map, lib.ptr(t), *t_map, lib.ptr(t_ids), lib.ptr(labels), lib.ptr(loss_row),
             lib.ptr(grad), rows, C, mode, lib.ptr(row_weight))
    return loss_row, grad


# ---- dynamic keep ratio (--patch-score-threshold): vit_models/dynamic_vit.py:880-894, :935-949 ----
def select_threshold(probs, threshold, lead=0):
    """-> (mask [B, lead + T] float 0/1, counts [B] int32): keep the tokens whose ascending cumulative probability exceeds `threshold`;
    the `lead` first columns are 1 (lead=1: the attention policy row [CLS, tokens])."""
    _f32(probs)
    B, T = probs.shape
    mask = torch.empty((B, lead + T), dtype=torch.float32, device=probs.device)
    counts = torch.empty((B,), dtype=torch.int32, device=probs.device)
    lib.call("d2s_select_threshold", lib.ptr(probs), B, T, float(threshold), lib.ptr(mask), lead + T, int(lead), lib.ptr(counts))
    return mask, counts


def gather_rows_i32(src, idx, rows):
    D = src.shape[-1]
    dst = torch.empty((rows, D), dtype=torch.float32, device=src.device)
    lib.call("d2s_gather_rows_i32", lib.ptr(_f32(src)), lib.ptr(idx), lib.ptr(dst), rows, D)
    return dst


def ragged_offsets(counts, extra=1):
    cu = torch.empty((counts.numel() + 1,), dtype=torch.int32, device=counts.device)
    lib.call("d2s_ragged_offsets", lib.ptr(counts), counts.numel(), int(extra), lib.ptr(cu))
    return cu


def ragged_pack(x, mask, cu, total):
    """x [B,n,D], mask [B,n-1], cu [B+1] -> (packed [total,D], row_src [total] int32 = source token of every packed row)."""
    _f32(x), _f32(mask)
    B, n, D = x.shape
    out = torch.empty((total, D), dtype=torch.float32, device=x.device)
    src = torch.empty((total,), dtype=torch.int32, device=x.device)
    lib.call("d2s_ragged_pack", lib.ptr(x), lib.ptr(mask), lib.ptr(cu), lib.ptr(out), lib.ptr(src), B, n, D)
    return out, src


def mask_row_weights(mask):
    m = _f32(mask).reshape(-1)
    w = torch.empty_like(m)
    lib.call("d2s_mask_row_weights", lib.ptr(m), m.numel(), lib.ptr(w))
    return w


def dense_mask_agreement(a, b):
    B, T = a.shape
    out = torch.empty((B,), dtype=torch.float32, device=a.device)
    lib.call("d2s_dense_mask_agreement", lib.ptr(_f32(a)), lib.ptr(_f32(b)), B, T, lib.ptr(out))
    return out


def patch_keep_mask(kept, N):
    """visualizations.py:18-26: kept ids [B,k] -> int64 mask [B,N] in token order (1 = kept)."""
    assert kept.dtype == torch.int64 and kept.is_contiguous()
    B, k = kept.shape
    mask = torch.empty((B, N), dtype=torch.int64, device=kept.device)
    lib.call("d2s_patch_keep_mask", lib.ptr(kept) if k else None, B, k, N, lib.ptr(mask))
    return mask


def compose_ids(prev, rel):
    """stage-relative ids -> the coordinates `prev` is expressed in: out[b,j] = prev[b, rel[b,j]]."""
    B, k = rel.shape
    out = torch.empty_like(rel)
    lib.call("d2s_compose_ids", lib.ptr(prev.contiguous()), prev.shape[1], lib.ptr(rel.contiguous()), k, lib.ptr(out), B)
    return out


def attn_policy_fwd(qkv, policy, B, n, H, scale, eps=1e-6, want_cls=False):
    out = torch.empty((B * n, H * 64), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    cinv = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    cls_row = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device) if want_cls else None
    lib.call("d2s_attn_policy_fwd_f32", lib.ptr(qkv), lib.ptr(_f32(policy)), lib.ptr(out), lib.ptr(lse), lib.ptr(cinv), lib.ptr(cls_row),
             B, n, H, float(scale), float(eps))
    return out, lse, cinv, cls_row


def attn_policy_bwd(qkv, policy, out, dout, lse, cinv, B, n, H, scale):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
    lib.call("d2s_attn_policy_bwd_f32", lib.ptr(qkv), lib.ptr(policy), lib.ptr(out), lib.ptr(dout), lib.ptr(lse), lib.ptr(cinv),
             lib.ptr(dqkv), lib.ptr(delta), B, n, H, float(scale))
    return dqkv


def attn_varlen_fwd(qkv, cu, B, total, max_n, H, scale, want_cls=False):
    out = torch.empty((total, H * 64), dtype=torch.float32, device=qkv.device)
    cls_row = torch.empty((H, total), dtype=torch.float32, device=qkv.device) if want_cls else None
    lib.call("d2s_attn_varlen_fwd_f32", lib.ptr(qkv), lib.ptr(cu), lib.ptr(out), lib.ptr(cls_row), B, total, max_n, H, float(scale))
    return out, cls_row


def sum_scalar(v, scale=1.0):
    out = torch.empty((), dtype=torch.float32, device=v.device)
    lib.call("d2s_sum_scalar", lib.ptr(v), v.numel(), float(scale), lib.ptr(out))
    return out


def scale_by_scalar(x, gscalar, scale=1.0):
    y = torch.empty_like(x)
    lib.call("d2s_scale_by_scalar", lib.ptr(x), lib.ptr(gscalar), float(scale), lib.ptr(y), x.numel())
    return y


def mask_agreement(ids_a, ids_b, T):
    B, k = ids_a.shape
    out = torch.empty((B,), dtype=torch.float32, device=ids_a.device)
    lib.call("d2s_mask_agreement", lib.ptr(ids_a) if k else None, lib.ptr(ids_b) if k else None, B, T, k, lib.ptr(out))
    return out


def act_grad(g, z, kind):
    """g * act'(z) (kind 'gelu': z = pre-activation; 'relu': z = ReLU output)."""
    out = torch.empty_like(g)
    lib.call("d2s_act_grad", lib.ptr(g), lib.ptr(z), lib.ptr(out), g.numel(), 0 if kind == "gelu" else 1)
    return out


def normal_noise(shape, seed, device):
    """Standard-normal tensor from the library's counter-based stream `seed` (no torch RNG, no host copy)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    lib.call("d2s_normal_noise", lib.ptr(out), out.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF)
    return out


def perturbed_topk_fwd(x, noise, k, sigma):
    b, d = x.shape
    nS = noise.shape[1]
    ind = torch.empty((b, k, d), dtype=torch.float32, device=x.device)
    need = lib.query("d2s_perturbed_topk_workspace_bytes", b, k, d)
    ws = workspace(need, x.device)
    lib.call("d2s_perturbed_topk_fwd", lib.ptr(x), lib.ptr(noise), lib.ptr(ind), b, nS, d, k, float(sigma), lib.ptr(ws), ws.numel())
    return ind


def perturbed_topk_bwd(x, noise, g, k, sigma):
    b, d = x.shape
    nS = noise.shape[1]
    gx = torch.empty((b, d), dtype=torch.float32, device=x.device)
    lib.call("d2s_perturbed_topk_bwd", lib.ptr(x), lib.ptr(noise), lib.ptr(g), lib.ptr(gx), b, nS, d, k, float(sigma))
    return gx


def adamw_step(params, grads, exp_avg, exp_avg_sq, desc, n_chunks, beta1, beta2, eps, step, grad_scale=1.0, chunk_steps=None):
    """chunk_steps: int32 [n_chunks] per-tensor update counters (torch.optim.AdamW's state['step']), advanced in place."""
    assert chunk_steps is None or (chunk_steps.dtype == torch.int32 and chunk_steps.numel() == n_chunks)
    lib.call("d2s_adamw_step", lib.ptr(params), lib.ptr(grads), lib.ptr(exp_avg), lib.ptr(exp_avg_sq), lib.ptr(desc), n_chunks,
             float(beta1), float(beta2), float(eps), int(step), float(grad_scale), lib.ptr(chunk_steps))


# ---- gradient-arena routing: the slice of a flat gradient arena that the gradient of a parameter is written into.  The record lives
# ON the Parameter object (not in a table keyed by its address), so it dies with the parameter, an arena is freed as soon as its model
# and TrainStep are, and a recycled device address can never alias a stale entry. ----
_GRAD_ATTR = "_d2s_grad_slice"


def register_grad_buffer(param, grad_view):
    """Route the gradient of `param` into `grad_view` (a slice of a flat gradient arena, see d2s.engine.ParamArena)."""
    base = grad_view._base if grad_view._base is not None else grad_view
    setattr(param, _GRAD_ATTR, (base, grad_view.storage_offset(), grad_view.numel(), tuple(grad_view.shape)))


def unregister_grad_buffer(param):
    if hasattr(param, _GRAD_ATTR):
        delattr(param, _GRAD_ATTR)


def grad_buffer(param):
    """Tensor to write the gradient of `param` into: a FRESH view of its arena slice when registered (a fresh tensor
    object lets autograd's AccumulateGrad adopt it without a copy), otherwise a new tensor."""
    ent = getattr(param, _GRAD_ATTR, None)
    if ent is None:
        return torch.empty_like(param)
    base, off, numel, shape = ent
    if shape != tuple(param.shape) or base.device != param.device:
        raise lib.D2SError("parameter was reshaped / moved after its gradient arena was built (rebuild the TrainStep)")
    return base.view(-1)[off:off + numel].view(shape)


GEMM_EXACT, GEMM_SPLIT, GEMM_BF16 = 0, 1, 2

# Arithmetic mode of the GEMM / attention calls.  The C ABI takes it per call (no state in the library); this module keeps the
# process default plus a per-thread override stack so that one module (e.g. the frozen teacher) can run in another mode than the
# rest.  Autograd Functions record the mode of their forward and re-enter it in their backward (d2s.functional.mode_recorded),
# because backward runs on autograd's own thread, outside any `with gemm_mode(...)` block of the caller.
_default_mode = GEMM_EXACT
_tls = threading.local()


def set_gemm_mode(mode):
    """Process default: 0 exact fp32 MFMA (bit-for-bit an fp32 fma chain); 1 bf16x3 split on the bf16 matrix cores (fp32-class
    accuracy); 2 bf16 operands with fp32 accumulation (forward, dgrad and - in this mode only - wgrad GEMMs, bf16 attention)."""
    global _default_mode
    mode = int(mode)
    if mode not in (GEMM_EXACT, GEMM_SPLIT, GEMM_BF16):
        raise ValueError(f"gemm mode {mode}")
    _default_mode = mode


def get_gemm_mode():
    stack = getattr(_tls, "stack", None)
    return stack[-1] if stack else _default_mode


class gemm_mode:
    """with ops.gemm_mode(ops.GEMM_BF16): ...   - overrides the process default for the calls of this thread inside the block."""

    def __init__(self, mode):
        self.mode = int(mode)
        if self.mode not in (GEMM_EXACT, GEMM_SPLIT, GEMM_BF16):
            raise ValueError(f"gemm mode {mode}")

    def __enter__(self):
        if not hasattr(_tls, "stack"):
            _tls.stack = []
        _tls.stack.append(self.mode)
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        return False
