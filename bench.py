#!/usr/bin/env python3
"""Headline benchmark: training images/s of the dense-to-sparse ViT step on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or - when WORLD_SIZE is not set - this process starts them itself
(launch_ranks: N fresh children through torch.distributed.run on 127.0.0.1, before anything here touches the GPU; the reference spawns
its workers the same way, mask_predictor.py:160-162 `mp.spawn`, ddp_training.py:4-8), forwards rank 0's JSON line and exits non-zero if
any child did.  A line is only printed when the process group's world size equals --gpus.

A "step" is exactly train.py:40-57 of the reference: frozen teacher forward, student forward (token scoring, top-k
selection, kept-token gather, pruned blocks), MaskLoss + BackboneLoss, backward, AdamW update - on one synthetic batch
resident in HBM.  Workload: DeiT-Small 224x224, single-stage pruning keep_ratio 0.5 at block 3, per-GPU batch 128
(weak scaling: the global batch is 128 * N), fp32 (the reference computes in fp32).  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline      the dominant kernel (fp32 MFMA GEMM family): algorithmic FLOPs of every launch in the timed region /
                their HIP-event durations, against the 157.3 TFLOP/s fp32 matrix peak of gfx950.
  gather        the kept-token gather/pack kernel: algorithmic bytes / HIP-event time, against 8 TB/s.
  cpu_baseline  the CPU oracle (plain torch fp32 restatement of the reference step, pinned to the reference by
                tests/golden) timed on this host on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "dense2sparse-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity figure is never used)
# per arithmetic mode: (kernel the GEMM work runs in, matrix instruction, peak the ALGORITHMIC 2MNK FLOPs are priced against)
GEMM_MODE_ROOFLINE = {
    "exact": ("gemm_f32_kernel", "v_mfma_f32_32x32x2_f32", PEAK_F32_MFMA_TFLOPS),
    # bf16x3 split: 6 bf16 MFMAs per fp32-class product -> the algorithmic ceiling is a sixth of the bf16 peak
    "split": ("gemm_pieces_nt_kernel<3 pieces>", "v_mfma_f32_32x32x16_bf16 x6 per product", PEAK_BF16_MFMA_TFLOPS / 6.0),
    # large shapes run in the LDS-DMA kernel (256x256 / 256x128 tiles), small ones in the 128x128 pieces kernel; the figure is the whole C-ABI
    # call, i.e. it includes the weight's bf16 conversion pass and, where the caller has no bf16 copy of the activation, that conversion too
    "bf16": ("gemm_bf16_dma_kernel (+ gemm_pieces_nt_kernel<1 piece> on small shapes, conversion passes included)", "v_mfma_f32_32x32x16_bf16", PEAK_BF16_MFMA_TFLOPS),
}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
LARGE_LAUNCH_BYTES = 64e6      # HBM-bound kernels: launches that move at least this much are also reported on their own (`large_launches`)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--keep", type=float, default=0.5)
    ap.add_argument("--gemm-mode", choices=["exact", "split", "bf16"], default="exact",
                    help="exact: v_mfma_f32_32x32x2_f32 (default, fp32 fma chain); split: bf16x3 on the bf16 matrix cores, fp32-class "
                         "accuracy; bf16: bf16 operands, fp32 accumulate (all three GEMM layouts; bias gradients, LayerNorm, softmax, selection, losses, AdamW stay fp32)")
    ap.add_argument("--config", choices=["headline", "c2", "c3", "c4", "c5"], default="headline",
                    help="BASELINE.json config matrix: headline = DeiT-S keep 0.5 B=128 (the metric's config, default); c2 = DeiT-S keep 0.7 "
                         "B=128; c3 = DeiT-S 3-stage 0.7/0.5/0.3 B=32/GPU; c4 = T2T-ViT-14 keep 0.5 B=64/GPU; c5 = DeiT-B 384^2 keep 0.3 "
                         "bf16 GEMM operands B=64/GPU.  Non-headline presets skip the CPU baseline and set batch / keep / gemm mode.")
    ap.add_argument("--reference-quirk", action="store_true",
                    help="c5 only: keep int(196 * ratio) = 58 tokens of the 576 (the reference hard-codes init_n = 14*14, dynamic_vit.py:828,852) "
                         "instead of int(576 * ratio) = 172")
    ap.add_argument("--gemm-shapes-out", default=None, help="write the in-step per-shape GEMM times (layout M N K launches us TFLOP/s) to this file")
    ap.add_argument("--serial", action="store_true",
                    help="issue every kernel on one stream also in the timed region (no teacher / weight-gradient stream overlap): the form in which "
                         "per-kernel durations are meaningful - used for the rocprofv3 kernel-stats profile that has to agree with roofline.avg_launch_us")
    ap.add_argument("--batch-override", type=int, default=0,
                    help="per-GPU batch that also overrides a --config preset's batch (extra data points; config.workload names the batch that ran)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="off",
                    help="capture the step into a hipGraph and replay it (d2s.engine.TrainStep graph mode): auto = small per-rank batches only; "
                         "off by default - measured slower than eager issue on ROCm 7.2 (profiles/r03_a_graph_vs_eager.txt)")
    ap.add_argument("--lookahead", choices=["on", "off"], default="off",
                    help="hand TrainStep the NEXT batch as well (here: the same synthetic batch), so that the frozen teacher's forward for step "
                         "t+1 is issued during step t and shares the GPU with its backward; one teacher forward per step either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse the N-rank launch path without a GPU: the ranks rendezvous over gloo, all-reduce their rank ids and rank 0 "
                         "prints a line with n_gpus / ranks read back from the process group (tests/test_bench_launcher.py)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the instrumented pass (no per-kernel figures in the line)")
    ap.add_argument("--time-kernels-in-region", action="store_true",
                    help="put the per-kernel HIP events inside the timed region itself (they cost ~5 %% of the step: ~900 event records)")
    a = ap.parse_args()
    a.locs, a.keeps, a.arch, a.img = [3], [a.keep], "deit_small", 224
    if a.config == "c2":
        a.keeps = [0.7]
    elif a.config == "c3":
        a.locs, a.keeps, a.batch = [3, 6, 9], [0.7, 0.5, 0.3], 32
    elif a.config == "c4":
        a.arch, a.keeps, a.batch = "t2t_vit_14", [0.5], 64
    elif a.config == "c5":
        a.arch, a.img, a.keeps, a.batch, a.gemm_mode = "deit_base", 384, [0.3], 64, "bf16"
    if a.batch_override > 0:
        a.batch = a.batch_override
    if a.config != "headline":
        a.no_cpu_baseline = True
        a.keep = a.keeps[0]
    a.n_patches = (a.img // 16) ** 2
    a.init_n = 196 if (a.reference_quirk or a.img == 224) else a.n_patches      # the count the keep ratios multiply
    a.kept = [int(a.init_n * r) for r in a.keeps]
    return a


class KernelTimer:
    """HIP events around selected C-ABI launches, recorded on the stream the kernels run on (torch's current stream,
    which is the stream d2s.lib passes to every entry point)."""

    def __init__(self):
        self.records = {}   # key -> list of (start, end, work)
        self.shapes = {}    # (layout, M, N, K) -> list of (start, end): per-shape in-step GEMM times (--gemm-shapes-out)
        self.enabled = False

    def wrap(self, ops):
        timer = self
        orig_gemm, orig_gather = ops.gemm, ops.gather_pack

        def gemm(layout, A, lda, B, ldb, C, ldc, M, N, K, *a, **kw):
            if not timer.enabled:
                return orig_gemm(layout, A, lda, B, ldb, C, ldc, M, N, K, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_gemm(layout, A, lda, B, ldb, C, ldc, M, N, K, *a, **kw)
            e.record()
            timer.records.setdefault(("gemm_f32", ("NT", "NN", "TN")[layout]), []).append((s, e, 2.0 * M * N * K))
            timer.shapes.setdefault((("NT", "NN", "TN")[layout], M, N, K), []).append((s, e))
            return out

        def gather_pack(x, ids):
            if not timer.enabled:
                return orig_gather(x, ids)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_gather(x, ids)
            e.record()
            B, n, D = x.shape
            k = ids.shape[1]
            timer.records.setdefault(("gather_pack", ""), []).append((s, e, B * (2.0 * (k + 1) * D * 4 + 8.0 * k)))
            return out

        def timed(fn, key, work_of):
            def wrapper(*a, **kw):
                if not timer.enabled:
                    return fn(*a, **kw)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = fn(*a, **kw)
                e.record()
                timer.records.setdefault(key, []).append((s, e, work_of(*a, **kw)))
                return out
            return wrapper

        # HBM-bound kernels: algorithmic bytes = what the kernel must read + write once
        ops.layernorm_bwd = timed(ops.layernorm_bwd, ("layernorm_bwd", ""),
                                  lambda x, rowmap, dy, w, mean, rstd, dx, add_src, dw, db, rows, D, *a, **kw:
                                  rows * D * 4.0 * (3 + (1 if add_src is not None else 0)))
        ops.layernorm_fwd = timed(ops.layernorm_fwd, ("layernorm_fwd", ""), lambda x, rowmap, w, b, rows, D, eps, **kw: rows * D * 8.0)
        # bf16 data path: read fp32, write the bf16 copy (+ the fp32 result unless want_f32=False)
        ops.layernorm_fwd_bf16 = timed(ops.layernorm_fwd_bf16, ("layernorm_fwd", ""),
                                       lambda x, rowmap, w, b, rows, D, eps, stats=True, want_f32=True: rows * D * (10.0 if want_f32 else 6.0))
        ops.adamw_step = timed(ops.adamw_step, ("adamw", ""), lambda params, *a, **kw: params.numel() * 28.0)

        orig_wgrad = ops.linear_wgrad

        def linear_wgrad(dy, x, dW, *a, **kw):
            if not timer.enabled:
                return orig_wgrad(dy, x, dW, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_wgrad(dy, x, dW, *a, **kw)
            e.record()
            xin = x if x is not None else kw["x16"]          # bf16 data path: the layer input was saved in bf16 only
            dyin = dy if dy is not None else kw["dy16"]      # ... and so was this gradient
            timer.records.setdefault(("gemm_f32", "TN"), []).append((s, e, 2.0 * dyin.shape[0] * dyin.shape[1] * xin.shape[1]))
            timer.shapes.setdefault(("TN", dyin.shape[1], xin.shape[1], dyin.shape[0]), []).append((s, e))
            return out

        ops.linear_wgrad = linear_wgrad
        orig_scatter = ops.scatter_unpack

        def scatter_unpack(g, ids, n):
            if not timer.enabled:
                return orig_scatter(g, ids, n)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_scatter(g, ids, n)
            e.record()
            B, k1, D = g.shape
            timer.records.setdefault(("scatter_unpack", ""), []).append((s, e, B * (k1 * D * 4.0 + n * D * 4.0 + 8.0 * (k1 - 1))))
            return out

        ops.gemm, ops.gather_pack, ops.scatter_unpack = gemm, gather_pack, scatter_unpack
        from d2s import lib as _lib
        orig_call = _lib.call

        def counted_call(name, *a):
            if timer.counting:
                timer.launch_calls += 1
            return orig_call(name, *a)

        _lib.call = counted_call
        self.launch_calls = 0
        self.counting = False       # C-ABI calls are counted in the TIMED region (the instrumented pass issues the per-op sequence)

    def summary(self):
        out = {}
        for key, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[key] = dict(launches=len(recs), ms=ms, work=sum(w for _, _, w in recs))
            big = [(s, e, w) for s, e, w in recs if w >= LARGE_LAUNCH_BYTES]      # launches long enough for bandwidth, not launch latency, to set their time
            if big and key[0] != "gemm_f32":
                out[key]["large"] = dict(launches=len(big), ms=sum(s.elapsed_time(e) for s, e, _ in big), work=sum(w for _, _, w in big))
        return out


def hbm_copy_gbs(device, mbytes=1024, reps=10, nbuf=1):
    """Device-to-device copy rate in this run (confirm the HBM figure the fractions are priced against): bytes read +
    bytes written per second of hipMemcpyAsync D2D copies of `mbytes` MiB.  nbuf > 1 rotates over that many source / destination pairs
    so that small copies come from HBM, not from the 256 MB Infinity Cache."""
    src = [torch.empty(mbytes << 20, dtype=torch.uint8, device=device) for _ in range(nbuf)]
    dst = [torch.empty_like(src[0]) for _ in range(nbuf)]
    for a, b in zip(src, dst):
        b.copy_(a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(reps):
        dst[i % nbuf].copy_(src[i % nbuf])
    e.record()
    torch.cuda.synchronize()
    return round(2.0 * src[0].numel() * reps / (s.elapsed_time(e) * 1e-3) / 1e9, 1)


def build(device, keep, seed=0, arch="deit_small", locs=(3,), keeps=None, img=224, init_n=196):
    import vit_models
    torch.manual_seed(seed)
    keeps = list(keeps) if keeps else [keep]
    if arch == "t2t_vit_14":
        student = vit_models.t2t_vit_14_student(list(locs), keeps)
        teacher = vit_models.t2t_vit_14_teacher()
    elif arch == "deit_base":
        geom = dict(img_size=img, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True)
        student = vit_models.VisionTransformerDiffPruning(pruning_loc=list(locs), token_ratio=keeps, distill=True, topk_selection=True,
                                                          predictor_loss_type="kl_div", init_n=init_n, **geom)
        teacher = vit_models.VisionTransformerTeacher(**geom)
    else:
        student = vit_models.dynamic_vit_small_patch16_224_student(list(locs), keeps, topk_selection=True, predictor_loss_type="kl_div")
        teacher = vit_models.dynamic_vit_small_patch16_224_teacher()
    return student.to(device), teacher.to(device)


def cpu_baseline(keep, batch=32, warm=2, steps=10):
    """The oracle's full train step (teacher fwd, student fwd, losses, backward, torch.optim.AdamW) on the host CPU."""
    from oracle import d2s_oracle as O
    from d2s import synth
    import numpy as np
    cfg = O.make_cfg(dim=384, depth=12, heads=6, pruning_loc=(3,), token_ratio=(keep,))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd_s = {k: t(v) for k, v in synth.fill_state_dict(O.student_param_shapes(cfg), seed=0).items()}
    sd_t = {k: t(v) for k, v in synth.fill_state_dict(O.teacher_param_shapes(cfg), seed=1).items()}
    # the container's CPU share, not the host's core count (oversubscribing OpenMP threads stalls for minutes)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, int(os.environ.get("D2S_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    st = O.TrainState(sd_s, sd_t, cfg, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0)
    x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    y = torch.randint(0, 1000, (batch,), generator=torch.Generator().manual_seed(1))
    for _ in range(warm):
        st.step(x, y)
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(x, y)
    dt = time.perf_counter() - t0
    return dict(value=round(batch * steps / dt, 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} full train steps of the CPU oracle at batch {batch} (same model/config, fp32), after {warm} warm-up; {dt:.1f} s")


PMC_FILE = "r03_m_pmc_traffic.json"
PMC_FILE_C5 = "r03_m_pmc_traffic_c5.json"      # the same two passes over `bench.py --config c5`


def pmc_traffic(kernel_prefix, pmc_file=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes): the launch-weighted
    mean over every kernel whose name starts with `kernel_prefix` (the tile-shape variants of one GEMM layout), or None."""
    try:
        with open(os.path.join(REPO, "profiles", pmc_file or PMC_FILE)) as f:
            ks = json.load(f)["kernels"]
        sel = [v for k, v in ks.items() if k.startswith(kernel_prefix)]
        n = sum(v["launches"] for v in sel)
        return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n) if n else None
    except Exception:
        return None


def gather_large(device, B=2048, n=197, k=98, D=384, iters=20):
    """The gather at a size where launch latency no longer dominates: 624 MB of algorithmic traffic per launch."""
    from d2s import ops
    x = torch.randn((B, n, D), device=device)
    ids = torch.sort(torch.rand((B, n - 1), device=device).argsort(dim=1)[:, :k], dim=1)[0].contiguous()
    for _ in range(3):
        ops.gather_pack(x, ids)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gather_pack(x, ids)
    e.record()
    torch.cuda.synchronize()
    nbytes = B * (2.0 * (k + 1) * D * 4 + 8.0 * k)
    gbs = nbytes / (s.elapsed_time(e) * 1e-3 / iters) / 1e9
    return {"achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "bytes_per_launch": nbytes}


def gather_back_to_back(device, B, n=197, k=98, D=384, iters=100, nbuf=16):
    """The headline-shape gather launched back to back over rotating inputs (16 x 39 MB > the 256 MB Infinity Cache, so reads
    come from HBM): one event pair around all launches, so the per-launch figure carries launch gaps but not the ~3 us that an
    event pair adds around a single 10 us kernel in the timed region."""
    from d2s import ops
    xs = [torch.randn((B, n, D), device=device) for _ in range(nbuf)]
    ids = torch.sort(torch.rand((B, n - 1), device=device).argsort(dim=1)[:, :k], dim=1)[0].contiguous()
    for i in range(3):
        ops.gather_pack(xs[i], ids)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        ops.gather_pack(xs[i % nbuf], ids)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    nbytes = B * (2.0 * (k + 1) * D * 4 + 8.0 * k)
    return {"us_per_launch": round(us, 2), "achieved": round(nbytes / us / 1e3, 1), "unit": "GB/s", "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4)}


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def launch_ranks(args):
    """--gpus N > 1 without a launcher around us: start N fresh rank processes (one per GPU, RCCL rendezvous on 127.0.0.1) and relay
    rank 0's JSON line.  Runs BEFORE this process has made any GPU call (no torch.cuda.*, no lib.load()); the ranks are children,
    never an exec of this process.  Returns the exit code for the parent: the children's if non-zero, 1 if no line came back."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # this pool's driver only supports dmabuf IPC (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    log(f"starting {args.gpus} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for ln in proc.stdout:
        if ln.lstrip().startswith("{"):
            lines.append(ln.strip())
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        log(f"rank processes exited with code {rc}")
        return rc
    if len(lines) != 1:
        log(f"expected ONE JSON line from rank 0, got {len(lines)}")
        return 1
    got = json.loads(lines[0]).get("n_gpus")
    if got != args.gpus:
        log(f"rank 0 reports n_gpus = {got}, asked for {args.gpus}: not printing")
        return 1
    print(lines[0], flush=True)
    return 0


def launch_check(args):
    """One rank of the launch rehearsal (no GPU): gloo rendezvous from the launcher's environment, every rank contributes its id, rank 0
    prints what the process group says about itself."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64))
    total = torch.tensor([float(rank + 1)])
    dist.all_reduce(total)
    n = dist.get_world_size()
    ok = n == args.gpus and int(total.item()) == n * (n + 1) // 2
    dist.barrier()
    if rank == 0 and ok:
        print(json.dumps({"launch_check": True, "n_gpus": n, "ranks": [int(t.item()) for t in ids], "backend": "gloo",
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}), flush=True)
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"launch check failed: world size {n}, --gpus {args.gpus}")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.launch_check:
        return launch_check(args)
    # RCCL prints a version banner on STDOUT at communicator creation; the contract is ONE JSON line there, so everything
    # before the final print is routed to stderr at the file-descriptor level.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("D2S_FORCE_DIST") == "1"     # rehearse the RCCL path with a single rank
    distributed = world > 1 or force_dist
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: refusing to report a line for a different rank count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the d2s path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from d2s import ops, lib
    from d2s.engine import TrainStep
    lib.load()
    ops.set_gemm_mode({"exact": 0, "split": 1, "bf16": 2}[args.gemm_mode])
    timer = KernelTimer()
    timer.wrap(ops)

    student, teacher = build(device, args.keep, arch=args.arch, locs=args.locs, keeps=args.keeps, img=args.img, init_n=args.init_n)
    targs = types.SimpleNamespace(keep_ratios=list(args.keeps), mask_loss_type="kl_div", mixup=0.0, patch_score_threshold=None, step=0)
    ts = TrainStep(student, teacher, targs, lr=5e-4, min_lr=1e-5, weight_decay=0.05, epochs=25, warmup_steps=0,
                   distributed=distributed, graph={"auto": "auto", "on": True, "off": False}[args.graph])
    if ts._use_graph(torch.empty((args.batch, 1), device=device)) and args.warmup < ts.GRAPH_WARM_STEPS + 1:
        args.warmup = ts.GRAPH_WARM_STEPS + 1        # the capture itself must not fall into the timed region
    if args.serial:
        ts._teacher_stream = None
        ops._WGRAD_ENABLED = False
    if distributed:
        ts.reducer.force = force_dist
        dist.broadcast(ts.arena.params, src=0)

    g = torch.Generator(device=device).manual_seed(1234 + rank)
    images = torch.randn((args.batch, 3, args.img, args.img), device=device, generator=g)
    labels = torch.randint(0, 1000, (args.batch,), device=device, generator=g)

    nxt = images if args.lookahead == "on" else None
    log(f"models built on {device}; {args.warmup} warm-up steps")
    for i in range(args.warmup):
        ts(images, labels, nxt)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region runs WITHOUT per-kernel events: ~900 event records per step between the kernels cost about 5 % of the step
    # (measured: 42.5 vs 40.4 ms), i.e. they would perturb the throughput they sit beside.  The per-kernel figures come from an
    # identical instrumented pass of the same K steps right after it (same process, same stream, same inputs).
    in_region = args.time_kernels_in_region and not args.no_kernel_timing
    timer.enabled = in_region
    timer.counting = True
    if in_region:
        ops._BLOCK_COMPOSITE = False        # the per-kernel events sit in the per-op wrappers (see the instrumented pass below)
    if distributed:
        ts.reducer.timing = True
        ts.reducer._n_collectives = 0       # count the timed region's collectives only (the warm-up steps issued some)
    host_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h0 = time.perf_counter()
        info = ts(images, labels, nxt)
        host_s += time.perf_counter() - h0       # time the host spends issuing one step (no synchronisation inside)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    timer.counting = False
    captured = ts.last_step_captured
    loss = float(info["loss"])
    log(f"timed region done: {args.steps} steps in {elapsed:.3f} s, loss {loss:.5f}")
    instr_elapsed = elapsed if in_region else None
    if not args.no_kernel_timing and not in_region:
        timer.enabled = True
        # per-kernel durations are only meaningful when kernels do not share the GPU: the instrumented pass runs the teacher's forward
        # in series with the student's (the timed region above overlaps them on two streams)
        two_streams = getattr(ts, "_teacher_stream", None)
        ts._teacher_stream = None
        graph_mode, ts.graph = ts.graph, False      # per-kernel events need the kernels issued one by one
        # ... and from the per-op wrappers the events sit in: the composite block entries (one C-ABI call per block, csrc/block.hip) issue
        # the same launches - same kernels, arguments and order, asserted bit-identical by tests/test_graph_gpu.py - from C
        composite, ops._BLOCK_COMPOSITE = ops._BLOCK_COMPOSITE, False
        wgrad_async, ops._WGRAD_ENABLED = ops._WGRAD_ENABLED, False      # ... and the weight gradients on the main stream
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ts(images, labels)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        instr_elapsed = time.perf_counter() - t1
        timer.enabled = False
        ts._teacher_stream = two_streams
        ts.graph = graph_mode
        ops._BLOCK_COMPOSITE = composite
        ops._WGRAD_ENABLED = wgrad_async
        log(f"instrumented pass done: {args.steps} steps in {instr_elapsed:.3f} s")

    if distributed:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if distributed and dist.get_world_size() != args.gpus:
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    if rank == 0:
        n_gpus = dist.get_world_size() if distributed else 1       # read back from the process group, not from the flag
        imgs = args.batch * n_gpus * args.steps
        line = {
            "metric": "training images/s, DeiT-S 224 keep_ratio=0.5 (dense-to-sparse ViT train step, teacher fwd + student fwd/bwd + AdamW)"
                      if args.config == "headline" and args.keep == 0.5 else
                      f"training images/s, BASELINE config {args.config} (NOT the headline metric): {args.arch} {args.img}^2 keep {args.keeps}",
            "value": round(imgs / elapsed, 2), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "host_enqueue_ms_per_step": round(1000.0 * host_s / args.steps, 3),
            "step_issue": "hipGraph replay (forward + losses + backward captured once; gradient exchange, AdamW launch and loss running means issued per step)"
                          if captured else "eager (one C-ABI call per kernel from the autograd Functions)",
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"exact": "f32", "split": "f32 (bf16x3-split MFMA, fp32-class accuracy)", "bf16": "bf16 GEMM operands, f32 accumulate/elsewhere"}[args.gemm_mode], "data": "synthetic (N(0,1) images, uniform labels, random-init weights)",
            "config": {"workload": (f"DeiT-Small 224x224 patch16, 1-stage prune keep_ratio={args.keep} @ block 3 (196->{int(196 * args.keep)} tokens), "
                                    f"large LN predictor, kl_div mask loss, per-GPU batch {args.batch}") if args.config in ("headline", "c2") else
                                   f"BASELINE config {args.config}: {args.arch} {args.img}x{args.img}, prune @ {args.locs} keep {args.keeps} "
                                   f"({args.n_patches} -> {' -> '.join(str(k) for k in args.kept)} tokens; keep counts = int({args.init_n} * ratio)"
                                   + (" - the reference's hard-coded init_n = 196 quirk" if args.init_n != args.n_patches else "") + "), "
                                   f"large LN predictor, kl_div mask loss, per-GPU batch {args.batch}",
                       "global_batch": args.batch * n_gpus, "parallelism": f"dp{n_gpus}", "final_loss": round(loss, 5)},
        }
        summ = timer.summary()
        if instr_elapsed is not None:
            line["kernel_timing"] = {"method": "HIP events around every GEMM / gather / scatter / LayerNorm / AdamW launch on the launch stream, "
                                               + ("inside the timed region" if in_region else
                                                  "in a second pass of the same K steps after the timed region (the events cost ~5 % of the step), issued op by op (the timed region issues a transformer block's launches through one composite C-ABI call), with the teacher's forward "
                                                  "in series with the student's and the weight-gradient GEMMs in series with the rest of backward - the timed region "
                                                  "overlaps both pairs on separate HIP streams, where per-kernel durations would read low because kernels share the GPU"),
                                     "instrumented_ms_per_step": round(1000.0 * instr_elapsed / args.steps, 3)}
        gemms = {k: v for k, v in summ.items() if k[0] == "gemm_f32"}
        if gemms:
            tot_ms = sum(v["ms"] for v in gemms.values())
            tot_fl = sum(v["work"] for v in gemms.values())
            dom = max(gemms, key=lambda k: gemms[k]["ms"])
            ach = gemms[dom]["work"] / (gemms[dom]["ms"] * 1e-3) / 1e12
            # the kernel, instruction and peak follow the arithmetic mode that actually ran (wgrad = the TN layout stays on the exact
            # fp32 kernel in split mode)
            mode_key = "exact" if (dom[1] == "TN" and args.gemm_mode == "split") else args.gemm_mode
            kern, instr, peak = GEMM_MODE_ROOFLINE[mode_key]
            assert ach <= peak, f"achieved {ach:.1f} TFLOP/s above the {peak} peak: wrong peak or wrong FLOP count"
            # NT / NN run in the [row][k]-image kernel (template argument = B layout), TN (wgrad) in the [k][row]-image kernel
            pmc_prefix = {"NT": "gemm_f32_rk_kernel<0,", "NN": "gemm_f32_rk_kernel<1,", "TN": "gemm_f32_kernel<1, 1,"}[dom[1]] if mode_key == "exact" \
                else ("gemm_bf16_dma_kernel" if mode_key == "bf16" else "gemm_pieces_nt_kernel")
            if mode_key == "exact" and dom[1] != "TN":
                kern = "gemm_f32_rk_kernel"
            line["roofline"] = {"bound": "mfma", "kernel": f"{kern}, {dom[1]} layout ({instr})",
                                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                "frac": round(ach / peak, 4), "traffic": pmc_traffic(pmc_prefix, PMC_FILE_C5 if args.config == "c5" else None) if args.config in ("headline", "c5") and (mode_key == "exact" or args.config == "c5") else None,
                                "traffic_note": f"HBM bytes per launch, launch-weighted mean over this layout's tile-shape variants, from the committed PMC passes (profiles/{PMC_FILE_C5 if args.config == 'c5' else PMC_FILE}); not collected live",
                                "avg_launch_us": round(1000.0 * gemms[dom]["ms"] / gemms[dom]["launches"], 2),
                                "launches_per_step": gemms[dom]["launches"] / args.steps,
                                "all_gemm_layouts": {k[1]: {"TFLOP/s": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                                                            "ms_per_step": round(v["ms"] / args.steps, 3)} for k, v in gemms.items()},
                                "gemm_share_of_step": round(tot_ms / (1000.0 * (instr_elapsed or elapsed)), 4),
                                "gemm_family_TFLOP/s": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2)}
        if args.gemm_shapes_out and timer.shapes:
            with open(args.gemm_shapes_out, "w") as f:
                f.write(f"# in-step GEMM times per shape (HIP events, instrumented pass of {args.steps} steps): layout M N K launches/step us TFLOP/s ms/step\n")
                rows = []
                for (lay, M, N, K), recs in timer.shapes.items():
                    us = sum(a.elapsed_time(b) for a, b in recs) * 1e3 / len(recs)
                    rows.append((len(recs) * us / args.steps / 1e3, lay, M, N, K, len(recs) / args.steps, us, 2.0 * M * N * K / us / 1e6))
                for ms, lay, M, N, K, n, us, tf in sorted(rows, reverse=True):
                    f.write(f"{lay} {M:6d} {N:5d} {K:6d} {n:5.1f} {us:8.1f} {tf:7.1f} {ms:7.3f}\n")
        line["c_abi_calls_per_step"] = round(timer.launch_calls / args.steps, 1)
        line["hbm_reserved_GiB"] = round(torch.cuda.memory_stats()["reserved_bytes.all.peak"] / 2**30, 1)      # caching-allocator pool (the host runs steps ahead)
        # side streams are chosen so that they sit on another hardware queue than the main stream (d2s.ops.concurrent_stream): purpose, candidates tried, verified
        line["side_streams"] = [{"for": p, "candidates_tried": n, "runs_beside_main": ok} for p, n, ok in ops.stream_picks]
        sc = summ.get(("scatter_unpack", ""))
        if sc:
            gbs = sc["work"] / (sc["ms"] * 1e-3) / 1e9
            line["scatter"] = {"bound": "hbm", "kernel": "scatter_unpack_kernel", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(gbs / PEAK_HBM_GBS, 4), "bytes_per_launch": sc["work"] / sc["launches"],
                               "avg_launch_us": round(1000.0 * sc["ms"] / sc["launches"], 2), "traffic": pmc_traffic("scatter_unpack_kernel")}
        ga = summ.get(("gather_pack", ""))
        if ga:
            gbs = ga["work"] / (ga["ms"] * 1e-3) / 1e9
            line["gather"] = {"bound": "hbm", "kernel": "gather_pack_kernel", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                              "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                              "bytes_per_launch": ga["work"] / ga["launches"], "avg_launch_us": round(1000.0 * ga["ms"] / ga["launches"], 2),
                              "traffic": pmc_traffic("gather_pack_kernel"), "at_batch_2048": gather_large(device),
                              "back_to_back_hbm_cold": gather_back_to_back(device, args.batch, k=int(196 * args.keep)) if args.config == "headline" else None}
        for key, kern in ((("layernorm_bwd", ""), "ln_bwd_kernel (+ fold)"), (("layernorm_fwd", ""), "ln_fwd_kernel"), (("adamw", ""), "adamw_kernel")):
            rec = summ.get(key)
            if rec:
                gbs = rec["work"] / (rec["ms"] * 1e-3) / 1e9
                line[key[0]] = {"bound": "hbm", "kernel": kern, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": rec["launches"] / args.steps,
                                "ms_per_step": round(rec["ms"] / args.steps, 3)}
                if rec.get("large"):      # the block LayerNorms at n = 197; the rest (n = 99 rows, the predictor's narrow ones) are launch-sized
                    lg = rec["large"]
                    lgbs = lg["work"] / (lg["ms"] * 1e-3) / 1e9
                    line[key[0]]["large_launches"] = {"min_bytes": LARGE_LAUNCH_BYTES, "achieved": round(lgbs, 1), "frac": round(lgbs / PEAK_HBM_GBS, 4),
                                                      "launches_per_step": lg["launches"] / args.steps, "ms_per_step": round(lg["ms"] / args.steps, 3)}
        line["hbm_copy_GBps_measured"] = hbm_copy_gbs(device)
        # the same device-to-device copy at the size of ONE gather / LayerNorm launch of this workload (19.5 MB read + 19.5 MB written):
        # what a 39 MB transfer can reach at all, launch ramp included - the yardstick for the in-step gather / scatter / LayerNorm rates
        line["hbm_copy_GBps_at_39MB"] = hbm_copy_gbs(device, mbytes=19, reps=64, nbuf=16)      # 16 x 38 MB > the Infinity Cache
        if distributed:
            c = ts.reducer.comm_summary(args.steps * (2 if (instr_elapsed is not None and not in_region) else 1))   # both passes all-reduce
            if c:
                line["comm"] = c
                # did the stream the step is issued on share a hardware queue with the process group's RCCL stream, and was the step moved
                line["comm"]["rccl_stream_probe"] = ts.pg_probe
        if n_gpus == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle (bounded sample)")
            line["cpu_baseline"] = cpu_baseline(args.keep)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
