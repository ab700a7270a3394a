"""ctypes binding of libfiber_hip.so (the C ABI declared in include/fiber_hip.h).

PyTorch is plumbing only: tensors provide device memory (``data_ptr()``) and the current HIP stream.  There is no
CPU or eager fallback -- if the shared library is missing or a tensor is not on a HIP device the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FIBER_HIP_LIB") or os.path.join(_HERE, "libfiber_hip.so")   # env: A/B builds (tools/)

P, I, F, L, U64 = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_uint64

# name -> argtypes (stream appended automatically)
SIGNATURES = {
    "fiber_gemm_nt_bf16": [P, P, P, P, P, P, P, I, P, I, P, I, I, I, I, I, I, I, I],
    "fiber_gemm_tn_bf16": [P, P, P, P, P, I, I, I, I, I, P, I, F],
    "fiber_gemm_tn_slabs_bf16": [P, P, P, P, P, I, I, I, I, I, P, I, F],
    "fiber_gemm_tn_rowmap_bf16": [P, P, P, P, P, I, I, I, I, I, P, I, F, P],
    "fiber_tn_fold_multi": [P, I, I],
    "fiber_ln_mlp_fwd_bf16": [P, P, P, P, P, P, P, P, I, I, I, F],
    "fiber_ln_mlp_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, I, I, I, F],
    "fiber_layernorm_fwd_bf16": [P, P, P, P, P, P, I, I, F],
    "fiber_layernorm_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, I, I],
    "fiber_layernorm_fwd_stream": [P, P, P, P, P, P, P, I, I, F, I],
    "fiber_layernorm_bwd_stream": [P, P, P, P, P, P, P, P, P, P, I, I, I],
    "fiber_patch_merge_ln_fwd_stream": [P, P, P, P, P, P, I, I, I, I, F, I],
    "fiber_patch_merge_ln_bwd_stream": [P, P, P, P, P, P, P, P, P, I, I, I, I, I],
    "fiber_stream_add": [P, I, P, P, P, P, L, F, U64, F, U64, P, P, P, L],
    "fiber_stream_add_bwd": [P, P, P, P, L, F, U64, F, U64, P, P, P, P, L],
    "fiber_cast_f32_bf16": [P, P, L],
    "fiber_patch_merge_ln_fwd_bf16": [P, P, P, P, P, P, I, I, I, I, F],
    "fiber_patch_merge_ln_bwd_bf16": [P, P, P, P, P, P, P, P, P, I, I, I, I],
    "fiber_window_attn_fwd_bf16": [P, P, P, P, I, I, I, I, I, I, I, I],
    "fiber_window_attn_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I],
    "fiber_mha_fwd_bf16": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, U64, P],
    "fiber_mha_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, F, F, U64, P],
    "fiber_roberta_embed_fwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, U64, P],
    "fiber_roberta_embed_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, U64, P],
    "fiber_im2col_patch4": [P, P, I, I, I],
    "fiber_im2col_patch4_pair": [P, P, P, P, I, I, I],
    "fiber_gelu_bwd_bf16": [P, P, P, L],
    "fiber_gelu_bwd_colsum_bf16": [P, P, P, P, P, I, I],
    "fiber_scale_add_bf16": [P, P, P, F, P, L],
    "fiber_dot_bf16": [P, P, P, L],
    "fiber_colsum_bf16": [P, P, P, I, I, I],
    "fiber_fold_rows_f32": [P, P, I, I],
    "fiber_dropout_bf16": [P, P, L, F, U64, P],
    "fiber_droppath_scale_f32": [P, I, F, U64, P],
    "fiber_rowscale_add_bf16": [P, P, P, P, L, L],
    "fiber_rowscale_colsum_bf16": [P, P, P, P, P, I, I, I],
    "fiber_ce_fwd_bf16": [P, P, P, P, P, I, I, L],
    "fiber_colsum_labelled_bf16": [P, P, P, P, I, I, L],
    "fiber_ce_bwd_bf16": [P, P, P, P, P, I, I, L],
    "fiber_adamw_multi_f32": [P, P, P, I, F, F, F, F, F, I, P],
    "fiber_resize_bicubic_norm_u8": [P, I, P, P, P, I, I, P, P],
    "fiber_mlm_mask_i64": [P, P, P, L, U64, C.c_uint, I, I, I, I],
    "fiber_transpose_multi_bf16": [P, I, I],
    "fiber_rowperm_cast_multi_bf16": [P, I, I],
    "fiber_dcn_gather_bf16": [P, P, P, P, I, I, I, I, I, I, I, I, I, I],
    "fiber_dcn_scatter_bf16": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I],
    "fiber_dcn_dx_bf16": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I],
}
# host-side helpers without a stream argument
PLAIN = {"fiber_layernorm_bwd_grid": [I], "fiber_window_attn_bwd_slices": [I, I], "fiber_window_attn_colsum_rows": [I, I, I], "fiber_colsum_slabs": [I, I], "fiber_tn_fold_blocks": [I, I, I, I], "fiber_colsum_labelled_slabs": [I], "fiber_gemm_row_tile": [I, I, I], "fiber_gemm_tn_splits": [I, I, I],
         "fiber_adamw_chunk": [], "fiber_resample_ksize": [I, I], "fiber_dcn_dx_workspace": [I, I, I, I, I, I, I]}

PLAIN_LONG = {"fiber_dcn_dx_workspace"}          # helpers returning a 64-bit count
_lib = None


class FiberHipError(RuntimeError):
    pass


def load():
    """Load libfiber_hip.so and declare every prototype.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FiberHipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C fiber_amd/csrc` (there is no fallback path)")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args + [P]
        fn.restype = I
    for name, args in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = L if name in PLAIN_LONG else I
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES) + list(PLAIN)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses anything that is not contiguous HIP memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise FiberHipError("fiber_amd ops need HIP device tensors (no CPU fallback)")
    return t.data_ptr()


_TRACE = [] if os.environ.get("FIBER_TRACE_LAUNCHES") else None     # debug: (stream, entry point, scalar args, event after launch)


def call(name, *args):
    lib = load()
    cur = torch.cuda.current_stream()
    stream = cur.cuda_stream
    rc = getattr(lib, name)(*args, stream)
    if _TRACE is not None:
        ev = torch.cuda.Event()
        ev.record(cur)
        _TRACE.append((stream, name, [a for a in args if isinstance(a, (int, float))], ev))
        del _TRACE[:-4000]
    if rc != 0:
        raise FiberHipError(f"{name} failed with code {rc} ({ {1: 'invalid argument', 2: 'launch failure'}.get(rc, '?')})")


def plain(name, *args):
    return getattr(load(), name)(*args)
