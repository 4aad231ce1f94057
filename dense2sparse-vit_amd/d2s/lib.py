"""ctypes binding of lib/libd2s_hip.so (the C ABI declared in include/d2s_hip.h).

There is no fallback: if the shared object is missing or a symbol cannot be resolved this raises, and every op
raises on a non-zero return code.  Tensors are passed as raw device pointers; the caller keeps them alive and all
work is enqueued on torch's current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D2S_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libd2s_hip.so")   # override: diagnostic builds

P, I, L, F, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t

# name -> (restype, [argtypes]); the trailing hipStream_t is appended automatically for int-returning entries
_SIGS = {
    "d2s_gemm_f32_workspace_bytes": (Z, [I, I, I, I, I]),
    "d2s_gemm_f32": (I, [I, P, L, P, L, P, L, I, I, I, I, P, P, L, P, I, I, I, I, I, P, Z]),
    "d2s_gemm_f32_bf16io": (I, [I, P, L, P, L, P, L, I, I, I, I, P, P, L, P, P, P, P, P, Z]),
    "d2s_convert_bf16": (I, [P, P, L]),
    "d2s_linear_wgrad_workspace_bytes": (Z, [I, I, I, I]),
    "d2s_linear_wgrad_f32": (I, [P, L, P, L, P, L, P, I, I, I, I, I, P, Z]),
    "d2s_linear_wgrad_f32_bf16x": (I, [P, P, L, P, L, P, L, P, I, I, I, I, P, Z]),
    "d2s_batchnorm_workspace_bytes": (Z, [L, I]),
    "d2s_batchnorm_fwd": (I, [P, P, P, P, P, P, P, P, L, I, F, F, I, P, Z]),
    "d2s_batchnorm_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, L, I, P, Z]),
    "d2s_transpose_f32": (I, [P, P, I, I]),
    "d2s_transpose_batched_f32": (I, [P, P, P, I]),
    "d2s_colsum_workspace_bytes": (Z, [I, I]),
    "d2s_colsum_f32": (I, [P, L, I, I, P, I, P, Z]),
    "d2s_layernorm_fwd": (I, [P, L, L, L, L, P, P, P, P, P, L, I, F]),
    "d2s_layernorm_fwd_bf16out": (I, [P, L, L, L, L, P, P, P, P, P, P, L, I, F]),
    "d2s_layernorm_bwd_workspace_bytes": (Z, [L, I]),
    "d2s_layernorm_bwd": (I, [P, L, L, L, L, P, P, P, P, P, P, P, P, I, I, L, I, P, Z]),
    "d2s_layernorm_bwd_bf16out": (I, [P, L, L, L, L, P, P, P, P, P, P, P, P, P, I, I, L, I, P, Z]),
    "d2s_softmax_rows": (I, [P, P, I, I]),
    "d2s_select_topk": (I, [P, I, I, I, P, P]),
    "d2s_gather_pack_fwd": (I, [P, P, P, I, I, I, I]),
    "d2s_scatter_unpack_bwd": (I, [P, P, P, I, I, I, I]),
    "d2s_half_mean_concat": (I, [P, P, P, I, I, I]),
    "d2s_im2col_patch": (I, [P, P, I, I, I, I, I]),
    "d2s_fill_cls": (I, [P, P, P, I, I, I]),
    "d2s_batch_sum": (I, [P, P, I, L, L, I]),
    "d2s_copy_rows": (I, [P, L, L, L, L, P, L, L, L, L, L, I, I]),
    "d2s_assemble_tokens": (I, [P, P, P, P, I, I, I]),
    "d2s_softmax_policy_fwd": (I, [P, P, P, I, I, I, F]),
    "d2s_softmax_policy_bwd": (I, [P, P, P, P, I, I, I, F]),
    "d2s_unfold_fwd": (I, [P, L, L, L, L, P, I, I, I, I, I, I, I]),
    "d2s_unfold_bwd": (I, [P, P, L, L, L, L, I, I, I, I, I, I, I]),
    "d2s_performer_workspace_bytes": (Z, [I, I]),
    "d2s_performer_attn_fwd": (I, [P, P, P, P, P, P, P, P, I, I, F, P, Z]),
    "d2s_performer_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, F, P, Z]),
    "d2s_attn_fwd_f32": (I, [P, P, P, P, I, I, I, F]),
    "d2s_attn_fwd_bf16": (I, [P, P, P, P, I, I, I, F]),
    "d2s_attn_fwd_bf16_bf16out": (I, [P, I, P, P, P, P, I, I, I, F]),
    "d2s_attn_bwd_f32": (I, [P, P, P, P, P, P, I, I, I, F]),
    "d2s_attn_bwd_bf16": (I, [P, P, P, P, P, P, I, I, I, F]),
    "d2s_attn_bwd_bf16_bf16out": (I, [P, I, P, P, P, P, P, P, I, I, I, F]),
    "d2s_attn_delta": (I, [P, P, P, I, I, I]),
    "d2s_attn_bwd_dq_f32": (I, [P, P, P, P, P, I, I, I, F]),
    "d2s_attn_bwd_dkv_f32": (I, [P, P, P, P, P, I, I, I, F]),
    "d2s_teacher_target": (I, [P, P, I, I, I, I]),
    "d2s_gather_renorm": (I, [P, P, P, I, I, I, I]),
    "d2s_kl_rows": (I, [P, L, L, L, L, P, L, L, L, L, P, P, P, P, L, I, I, P]),
    "d2s_select_threshold": (I, [P, I, I, F, P, L, I, P]),
    "d2s_gather_rows_i32": (I, [P, P, P, I, I]),
    "d2s_ragged_offsets": (I, [P, I, I, P]),
    "d2s_ragged_pack": (I, [P, P, P, P, P, I, I, I]),
    "d2s_mask_row_weights": (I, [P, L, P]),
    "d2s_dense_mask_agreement": (I, [P, P, I, I, P]),
    "d2s_patch_keep_mask": (I, [P, I, I, I, P]),
    "d2s_compose_ids": (I, [P, I, P, I, P, I]),
    "d2s_attn_policy_fwd_f32": (I, [P, P, P, P, P, P, I, I, I, F, F]),
    "d2s_attn_policy_bwd_f32": (I, [P, P, P, P, P, P, P, P, I, I, I, F]),
    "d2s_attn_varlen_fwd_f32": (I, [P, P, P, P, I, I, I, I, F]),
    "d2s_sum_scalar": (I, [P, L, F, P]),
    "d2s_scale_by_scalar": (I, [P, P, F, P, L]),
    "d2s_mask_agreement": (I, [P, P, I, I, I, P]),
    "d2s_act_grad": (I, [P, P, P, L, I]),
    "d2s_normal_noise": (I, [P, L, ctypes.c_ulonglong]),
    "d2s_perturbed_topk_workspace_bytes": (Z, [I, I, I]),
    "d2s_perturbed_topk_fwd": (I, [P, P, P, I, I, I, I, F, P, Z]),
    "d2s_perturbed_topk_bwd": (I, [P, P, P, P, I, I, I, I, F]),
    "d2s_block_saved_floats": (L, [I, I, I, I, I, I]),
    "d2s_block_bwd_scratch_floats": (L, [I, I, I, I, I]),
    "d2s_block_workspace_bytes": (Z, [I, I, I, I, I]),
    "d2s_block_wgrad_workspace_bytes": (Z, [I, I, I, I, I]),
    "d2s_block_fwd_f32": (I, [P, P, I, I, I, I, I, F, F, P, P, P, I, I, P, Z]),
    "d2s_block_bwd_f32": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, P, I, P, Z, P, Z, P]),
    "d2s_adamw_chunk_elems": (I, None),
    "d2s_adamw_step": (I, [P, P, P, P, P, I, F, F, F, I, F, P]),
}

_lib = None


class D2SError(RuntimeError):
    pass


def load():
    """Load the library once; raises D2SError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D2SError(f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or `make -C dense2sparse-vit_amd/csrc`). The d2s path has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = [] if args is None else list(args) + ([P] if res is I else [])
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise D2SError("d2s ops need device tensors (no CPU fallback)")
    return t.data_ptr()


# The host side of a step is ~430 C-ABI calls; at small per-GPU batches (BASELINE config 3: 32 images) and in the bf16 arithmetic mode the
# step is bound by how fast Python can issue them (measured: 12.8 ms of enqueue per 12.8 ms step at B = 32), so the per-call overhead
# matters: the raw handle of torch's current stream comes from the C-level getter (0.3 us; torch.cuda.current_stream().cuda_stream builds
# a Stream object: 8 us), and the bound ctypes functions are looked up once.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
_fns = {}


def stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _fn(name):
    f = _fns.get(name)
    if f is None:
        f = _fns[name] = getattr(load(), name)
    return f


def call(name, *args):
    """Invoke an int-returning entry point on the current stream; raise on error code."""
    rc = _fn(name)(*args, stream())
    if rc != 0:
        raise D2SError(f"{name} failed with code {rc}")


def query(name, *args):
    return _fn(name)(*args)
