"""Autograd wrappers around the fiber_hip C ABI (fiber_amd/lib.py).

Activations are bf16, parameters stay fp32 masters (state-dict compatible with the reference); each op casts the
weights it needs to bf16 once per parameter version.  Forward GEMMs with fused epilogues, LayerNorm, all attention
cores, embeddings and the element-wise glue are hand-written HIP kernels, and so are both backward GEMMs of every linear:
dX = dY.W on the NT kernel (transposed bf16 working copy of W), dW = dY^T.X (+ the bias gradient) on the TN kernel
(csrc/gemm_tn.hip).  Only the caller-side heads (vocabulary decoder, 2-way ITM, VQA classifier) use the library GEMM.
Nothing here runs on the CPU: tensors must live on a HIP device.
"""
import math
import threading
import weakref

import torch

from . import lib

import os

BF16 = torch.bfloat16
# Derived copies of parameters (bf16 / transposed / permuted working copies), keyed by (kind, id(parameter)).  The parameter
# itself is held WEAKLY: when a module is deleted its entries go with it (a finalizer purges them), so the cache neither keeps
# dead models' weights alive nor hands a stale copy to a new tensor that happens to reuse the id.
_wcache = {}


def _cache_get(key, w):
    hit = _wcache.get(key)
    if hit is not None and hit[2]() is w:
        return hit
    return None


def _cache_put(key, stamp, value, w):
    _wcache[key] = (stamp, value, weakref.ref(w, lambda _r, k=key: _wcache.pop(k, None)))
# (dY.W2^T)*gelu'(H) + fc1 bias gradient as ONE hand-written GEMM instead of library GEMM + gelu_bwd_colsum.  Measured at 512
# images (tools/op_bench.py 512 mlpbwd, persistent 256x256 kernel + select-free gelu', library dgrad in NT form): 3866 vs 4294 us
# at C=128, 2084 vs 2369 at C=256, 1284 vs 1394 at C=512, 899 vs 859 at C=1024 in round 2 (hence "auto" = below C = 1024 then).  Round 4:
# with both wave groups of the q8 kernel in the epilogue together the gelu' form gained 5-12 %, and the whole step is 0.8 ms faster with
# stage 3 fused as well (275.1 / 275.6 -> 274.5 / 274.7 ms, same box) -> "auto" = everywhere; FIBER_FUSED_MLP_BWD=0 / 1 forces it off / on.
_FUSED_MLP_BWD = os.environ.get("FIBER_FUSED_MLP_BWD", "auto")
_WIN_COLSUM = os.environ.get("FIBER_WIN_COLSUM", "0") == "1"     # dqkv column sums inside the window backward (see _WindowAttn.backward)


_gen = [0]


def mark_weights_dirty():
    """Invalidate every cached bf16 working copy.  Fused / foreach optimizers update parameters through TensorList
    kernels that do NOT bump `Tensor._version`, so the version counter alone would leave stale weights in use;
    fiber_utils.set_schedule registers this as an optimizer step post-hook."""
    _gen[0] += 1


def _stamp(w):
    return (w._version, _gen[0])


def bf16_weight(w):
    """bf16 copy of an fp32 parameter, refreshed when the parameter changes (version counter or optimizer step)."""
    key = id(w)
    hit = _cache_get(key, w)
    if hit is not None and hit[0] == _stamp(w):
        return hit[1]
    if hit is not None and hit[1].shape == w.shape and hit[1].device == w.device:
        wb = hit[1]
        wb.copy_(w.detach())                             # same storage: views into packed buffers / optimizer tables stay valid
    else:
        wb = w.detach().to(BF16).contiguous()
    _cache_put(key, _stamp(w), wb, w)
    return wb


def bf16_weight_t(w):
    """bf16 TRANSPOSED copy of an fp32 [N, K] parameter -> [K, N] (the B operand of the dgrad GEMM dX = dY . W in the
    kernel's NT form), refreshed when the parameter's version counter changes.  Copies made here are remembered: after an
    optimizer step refresh_transposed_copies() rewrites all of them in one launch instead of one strided copy per weight."""
    key = ("T", id(w))
    hit = _cache_get(key, w)
    if hit is not None and hit[0] == _stamp(w):
        return hit[1]
    wb = bf16_weight(w)
    if hit is not None and hit[1].shape == (wb.shape[1], wb.shape[0]) and hit[1].device == wb.device:
        wt = hit[1]
        wt.copy_(wb.t())                                 # same storage: pointer tables built on it stay valid
    else:
        wt = wb.t().contiguous()
        _t_registry_version[0] += 1
    _cache_put(key, _stamp(w), wt, w)
    return wt


_t_registry_version = [0]            # bumped whenever a transposed copy gets new storage
_t_table = {}                        # device -> (registry version, members, device table, tiles)


def refresh_transposed_copies():
    """Rewrite every cached transposed copy whose plain bf16 copy is current (i.e. was just rewritten by the fused optimizer) in
    ONE kernel launch and mark it current.  Called by FiberAdamW.step() after it restamped the plain copies."""
    if not _wcache:
        return
    by_dev = {}
    for key, (stamp, wt, ref) in list(_wcache.items()):
        if not (isinstance(key, tuple) and key[0] == "T"):
            continue
        w = ref()
        if w is None or not wt.is_cuda:
            continue
        plain = _cache_get(id(w), w)
        if plain is None or plain[0] != _stamp(w) or plain[1].shape != (wt.shape[1], wt.shape[0]):
            continue                                     # plain copy stale or gone: the lazy path rebuilds both
        if (wt.shape[0] % 8) or (wt.shape[1] % 8):
            continue
        by_dev.setdefault(wt.device, []).append((key, w, plain[1], wt, ref))
    for pk in list(_packs.values()):                      # packed projections (linear_packed): ONE descriptor per pack
        ws = [r() for r in pk["refs"]]
        if any(w is None for w in ws) or not pk["plain"].is_cuda:
            continue
        hits = [_cache_get(id(w), w) for w in ws]
        if any(h is None or h[0] != _stamp(w) for h, w in zip(hits, ws)) or not _pack_views_ok(pk, ws):
            continue                                         # some member stale: the lazy path (_pack_get) refreshes
        by_dev.setdefault(pk["plain"].device, []).append((None, pk, pk["plain"], pk["t"], None))
    for dev, items in by_dev.items():
        members = tuple((p.data_ptr(), t.data_ptr()) for _, _, p, t, _ in items)
        ent = _t_table.get(dev)
        if ent is None or ent[0] != members:
            rows, tile0 = [], 0
            for _, _, p, t, _ in items:
                N, K = p.shape
                tk = -(-K // 64)
                rows.append((p.data_ptr(), t.data_ptr(), N | (K << 32), tile0 | (tk << 32)))
                tile0 += -(-N // 64) * tk
            ent = (members, torch.tensor(rows, dtype=torch.int64).to(dev), tile0)
            _t_table[dev] = ent
        with torch.cuda.device(dev):
            lib.call("fiber_transpose_multi_bf16", lib.ptr(ent[1]), len(items), ent[2])
        for key, w, _, wt, ref in items:
            if key is None:
                w["t_stamp"] = tuple(_stamp(r()) for r in w["refs"])     # a pack: its transposed copy is current
            else:
                _wcache[key] = (_stamp(w), wt, ref)


_hm_table = {}                       # device -> (members, device table, tiles)
_HM_REFRESH = os.environ.get("FIBER_HM_REFRESH", "1") != "0"      # 0: the lazy ATen refresh in the next forward (A/B)


def refresh_head_major_copies():
    """Rewrite every cached head-major qkv working copy (permuted bf16 weight, its transpose, permuted fp32 bias: _LinearQKVHeadMajor) from the
    fp32 parameters in ONE launch and mark it current.  Called by FiberAdamW.step() after the parameter update, like refresh_transposed_copies();
    copies the optimizer did not touch are rewritten too (same values)."""
    if not _HM_REFRESH:
        return
    by_dev = {}
    for key, (stamp, val, ref) in list(_wcache.items()):
        if not (isinstance(key, tuple) and key[0] == "HM"):
            continue
        w = ref()
        b = val[3]() if len(val) > 3 else None
        if w is None or b is None or not val[0].is_cuda or w.dtype != torch.float32 or b.dtype != torch.float32 or not w.is_contiguous():
            continue
        if (w.shape[0] % 8) or (w.shape[1] % 8):
            continue
        by_dev.setdefault(w.device, []).append((key, w, b, val, ref))
    for dev, items in by_dev.items():
        members = tuple((w.data_ptr(), b.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr()) for _, w, b, v, _ in items)
        ent = _hm_table.get(dev)
        if ent is None or ent[0] != members:
            rows, tile0 = [], 0
            for _, w, b, v, _ in items:
                N, K = w.shape
                tk = -(-K // 64)
                pm = _qkv_perm32(K, v[4], dev)
                rows.append((w.data_ptr(), pm.data_ptr(), v[0].data_ptr(), v[2].data_ptr(), b.data_ptr(), v[1].data_ptr(), N | (K << 32), tile0 | (tk << 32)))
                tile0 += -(-N // 64) * tk
            ent = (members, torch.tensor(rows, dtype=torch.int64).to(dev), tile0)
            _hm_table[dev] = ent
        with torch.cuda.device(dev):
            lib.call("fiber_rowperm_cast_multi_bf16", lib.ptr(ent[1]), len(items), ent[2])
        for key, w, b, v, ref in items:
            _wcache[key] = ((_stamp(w), _stamp(b)), v, ref)


def bf16_copy_if_cached(w):
    """The cached bf16 working copy of `w` (whatever its state), or None -- for the fused optimizer, which rewrites it."""
    hit = _cache_get(id(w), w)
    return hit[1] if hit is not None else None


def restamp_bf16_copies(params, bump=True):
    """After an optimizer step that rewrote the plain bf16 copies in place: every other derived cache entry (transposed /
    permuted copies) is invalidated (`bump`, once per step), the rewritten ones are marked current."""
    if bump:
        _gen[0] += 1
    for w in params:
        hit = _cache_get(id(w), w)
        if hit is not None:
            _wcache[id(w)] = (_stamp(w), hit[1], hit[2])


def cast_bf16(w):
    """Autograd-visible bf16 view of an fp32 parameter for library GEMMs (gradient flows back in fp32)."""
    return w.to(BF16)


def clear_weight_cache():
    _wcache.clear()
    _packs.clear()
    _hm_table.clear()


def _rows(x):
    return x.numel() // x.shape[-1]


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- fp32 residual stream ---------------------------------------------------------------------------------------------------
# config["residual_dtype"] = "fp32" keeps the residual streams of both backbones (Swin: x = x + branch, reference
# swin_transformer.py:388-391; RoBERTa: the LayerNorm inputs a + h / dense + input and the LayerNorm outputs that become the next
# residual, roberta.py:485,417-423) in fp32 instead of rounding them to bf16 at every block.  A stream tensor then travels as a
# PAIR: the bf16 tensor autograd sees (the "shadow": what GEMMs read, what gradients -- bf16 -- flow along) carrying the fp32
# payload as the attribute `_f32`.  Ops that sit on the stream (layernorm_res / layernorm / patch_merge_ln read it; the proj / fc2 /
# dense epilogues and stream_add write it) use the payload when it is there and fall back to the bf16 tensor when it is not (a
# tensor that was cloned / sliced by a caller simply loses the extra precision).  Gradients stay bf16 either way.
_RESIDUAL_FP32 = [os.environ.get("FIBER_RESIDUAL_DTYPE", "bf16").lower() in ("fp32", "float32")]


def set_residual_dtype(dtype):
    """"fp32" / torch.float32: new residual streams start in fp32 (see above); "bf16": the round-1/2 behaviour."""
    _RESIDUAL_FP32[0] = dtype in ("fp32", "float32", torch.float32)


def residual_fp32():
    return _RESIDUAL_FP32[0]


def f32_of(t):
    """The fp32 payload of a stream tensor, or None."""
    return getattr(t, "_f32", None) if t is not None else None


def with_f32(t16, t32):
    if t32 is not None:
        t16._f32 = t32
    return t16


def start_stream(t16):
    """A bf16 tensor that begins a residual stream (PatchMerging output, text embeddings): in fp32 mode attach its fp32 copy."""
    if _RESIDUAL_FP32[0] and f32_of(t16) is None:
        t16._f32 = t16.detach().float()
    return t16


def gemm_nt(x2, wb, bias=None, residual=None, act=0, want_pre=False, rowscale=None, rows_per_sample=0, aux=None,
            want_colsum=False, out_fp32=False, res32=None):
    """y = act(x2 @ wb^T + bias) + residual  on the HIP kernel.  x2 [M,K] bf16 (row stride may exceed K).
    out_fp32 (plain / bias only): the result is stored in fp32.
    res32 (fp32 [M, N], instead of `residual`): the fp32 residual-stream form -- returns (y16, y32): the sum in fp32 and its
    bf16 shadow."""
    M, K = x2.shape
    N = wb.shape[0]
    assert x2.dtype == BF16 and wb.dtype == BF16 and x2.stride(1) == 1 and wb.stride(1) == 1
    if res32 is not None:
        assert res32.dtype == torch.float32 and res32.stride(1) == 1 and not act and residual is None and not want_colsum and not out_fp32
        y16 = torch.empty((M, N), dtype=BF16, device=x2.device)
        y32 = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        lib.call("fiber_gemm_nt_bf16", lib.ptr(x2), lib.ptr(wb), lib.ptr(bias), lib.ptr(res32), lib.ptr(y16), lib.ptr(y32),
                 lib.ptr(rowscale), rows_per_sample, None, 0, None, M, N, K, x2.stride(0), wb.stride(0), N, res32.stride(0), 0x800)
        return y16, y32
    y = torch.empty((M, N), dtype=torch.float32 if out_fp32 else BF16, device=x2.device)
    if out_fp32:
        assert not act and residual is None and not want_colsum
        act = 0x100
    pre = torch.empty((M, N), dtype=BF16, device=x2.device) if (want_pre and act) else None
    colpart = None
    if want_colsum:
        tiles = -(-M // lib.plain("fiber_gemm_row_tile", M, N, K))
        colpart = torch.empty((tiles, N), dtype=torch.float32, device=x2.device)
    lib.call("fiber_gemm_nt_bf16", lib.ptr(x2), lib.ptr(wb), lib.ptr(bias), lib.ptr(residual), lib.ptr(y), lib.ptr(pre),
             lib.ptr(rowscale), rows_per_sample, lib.ptr(aux), aux.stride(0) if aux is not None else 0, lib.ptr(colpart),
             M, N, K, x2.stride(0), wb.stride(0), N, residual.stride(0) if residual is not None else 0, act)
    if want_colsum:
        cs = torch.empty(N, dtype=torch.float32, device=x2.device)
        lib.call("fiber_fold_rows_f32", lib.ptr(colpart), lib.ptr(cs), colpart.shape[0], N)
        return y, cs
    return y, pre


def gelu_bwd_colsum(dg, h):
    """dh = dg * gelu'(h) and its column sums (bias gradient) in one pass."""
    M, N = dg.shape
    dh = torch.empty_like(dg)
    db = torch.empty(N, dtype=torch.float32, device=dg.device)
    slabs = lib.plain("fiber_colsum_slabs", M, N)
    ws = torch.empty(slabs * N, dtype=torch.float32, device=dg.device) if slabs > 1 else None
    lib.call("fiber_gelu_bwd_colsum_bf16", lib.ptr(dg), lib.ptr(h), lib.ptr(dh), lib.ptr(db), lib.ptr(ws), M, N)
    return dh, db


def rowscale_colsum(dy2, rowscale):
    """ds = per-sample scale * dy and its column sums (bias gradient) in one pass."""
    M, N = dy2.shape
    ds = torch.empty_like(dy2)
    db = torch.empty(N, dtype=torch.float32, device=dy2.device)
    slabs = lib.plain("fiber_colsum_slabs", M, N)
    ws = torch.empty(slabs * N, dtype=torch.float32, device=dy2.device) if slabs > 1 else None
    lib.call("fiber_rowscale_colsum_bf16", lib.ptr(dy2), lib.ptr(rowscale), lib.ptr(ds), lib.ptr(db), lib.ptr(ws), M, N,
             M // rowscale.numel())
    return ds, db


def colsum(x2):
    M, N = x2.shape
    out = torch.empty(N, dtype=torch.float32, device=x2.device)
    slabs = lib.plain("fiber_colsum_slabs", M, N)
    ws = torch.empty(slabs * N, dtype=torch.float32, device=x2.device) if slabs > 1 else None
    lib.call("fiber_colsum_bf16", lib.ptr(x2), lib.ptr(out), lib.ptr(ws), M, N, x2.stride(0))
    return out


_bmm_fp32 = [None]

# ---- library GEMMs are kept in ONE global order across HIP streams ------------------------------------------------
# hipBLASLt picks stream-K kernels for many of these shapes (…_SK3_… in the rocprof traces): persistent grids whose
# workgroups spin on each other's partial tiles and therefore assume they are all resident.  Two of them in flight on two
# streams (the fused branch runs the text stack on a second stream) can each hold half the CUs and wait for the other half
# forever -- observed as a 100 %-busy GPU with both streams parked behind a library GEMM.  Every library GEMM issued from
# this package therefore waits for the previous one, whichever stream it was issued on (one event wait, only when the stream
# changes); library GEMMs still overlap with every other kernel.
_lib_gemm_last = {"event": None, "stream": None}


class lib_gemm:
    def __enter__(self):
        self.capturing = torch.cuda.is_current_stream_capturing()
        if self.capturing:                                   # one stream by construction: nothing to order
            return self
        self.cur = torch.cuda.current_stream()
        last = _lib_gemm_last
        if last["event"] is not None and last["stream"] != self.cur.cuda_stream:
            self.cur.wait_event(last["event"])
        return self

    def __exit__(self, *exc):
        if self.capturing:
            return False
        ev = torch.cuda.Event()
        ev.record(self.cur)
        _lib_gemm_last["event"], _lib_gemm_last["stream"] = ev, self.cur.cuda_stream
        return False


def lib_matmul(a, b):
    with lib_gemm():
        return torch.matmul(a, b)


class _LibLinear(torch.autograd.Function):
    """y = x.W^T + b on the library GEMM (caller-side heads: vocabulary decoder, ITM / VQA classifiers), forward and both
    backward GEMMs inside the global library-GEMM order."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        with lib_gemm():
            return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        with lib_gemm():
            if ctx.needs_input_grad[0]:
                dx = torch.matmul(dy2, w).view(x.shape)
            if ctx.needs_input_grad[1]:
                dw = torch.matmul(dy2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            hint = _take_labelled_rows(dy2)
            if hint is not None:                             # dlogits of _CrossEntropy: its ignored rows are zero, only the labelled ones are read
                lab, ignore = hint
                M, N = dy2.shape
                slabs = lib.plain("fiber_colsum_labelled_slabs", M)
                db32 = torch.empty(N, dtype=torch.float32, device=dy2.device)
                ws = torch.empty(slabs * N, dtype=torch.float32, device=dy2.device) if slabs > 1 else None
                lib.call("fiber_colsum_labelled_bf16", lib.ptr(dy2), lib.ptr(lab), lib.ptr(db32), lib.ptr(ws), M, N, ignore)
                db = db32.to(dy2.dtype)                      # (what dy2.sum(0) returns: fp32 accumulation, one rounding)
            else:
                db = dy2.sum(0)
        return dx, dw, db


# _CrossEntropy.backward -> the linear that produced the logits: "the rows of this gradient whose label is the ignore index are zero".
_rows_tls = threading.local()


def _offer_labelled_rows(dx, labels, ignore):
    _rows_tls.slot = (dx.data_ptr(), tuple(dx.shape), labels, ignore)


def _take_labelled_rows(t2d):
    h = getattr(_rows_tls, "slot", None)
    _rows_tls.slot = None
    if h is not None and t2d.is_cuda and t2d.dtype == BF16 and t2d.is_contiguous() and h[0] == t2d.data_ptr() and h[1] == tuple(t2d.shape):
        return h[2], h[3]
    return None


class _CrossEntropy(torch.autograd.Function):
    """mean over the non-ignored rows of logsumexp(x) - x[label] for bf16 logits [rows, V] (F.cross_entropy semantics with
    ignore_index), forward and backward on csrc/loss.hip: no fp32 copy of the logits, no log-softmax tensor."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        rows, V = logits.shape
        x = logits if logits.is_contiguous() else logits.contiguous()
        lab = labels.contiguous()
        loss = torch.empty(rows, dtype=torch.float32, device=x.device)
        lse = torch.empty_like(loss)
        pred = torch.empty(rows, dtype=torch.int32, device=x.device)
        lib.call("fiber_ce_fwd_bf16", lib.ptr(x), lib.ptr(lab), lib.ptr(loss), lib.ptr(lse), lib.ptr(pred), rows, V, int(ignore_index))
        # arg max of the labelled rows (-1 on the others), for the accuracy metric: handed over on the logits object (fiber_utils.Accuracy)
        logits._fiber_argmax = (pred, lab, int(ignore_index))
        nvalid = (lab != ignore_index).sum().clamp(min=1).float()
        ctx.save_for_backward(x, lab, lse, nvalid)
        ctx.ignore = int(ignore_index)
        return loss.sum() / nvalid

    @staticmethod
    def backward(ctx, g):
        x, lab, lse, nvalid = ctx.saved_tensors
        scale = (g.float() / nvalid).reshape(1).contiguous()
        dx = torch.empty_like(x)
        lib.call("fiber_ce_bwd_bf16", lib.ptr(x), lib.ptr(lab), lib.ptr(lse), lib.ptr(scale), lib.ptr(dx), x.shape[0], x.shape[1],
                 ctx.ignore)
        _offer_labelled_rows(dx, lab, ctx.ignore)
        return dx, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """F.cross_entropy(logits.float(), labels, ignore_index=...) for bf16 logits [rows, V] (mean over valid rows)."""
    assert logits.dtype == BF16 and logits.dim() == 2 and labels.dtype == torch.int64
    return _CrossEntropy.apply(logits, labels, ignore_index)


# Call sites of the library GEMM (file names), recorded when a test switches it on: the library is allowed for the caller-side heads
#  and the ITC similarity matrices only -- tests/test_hip_modules.py::test_library_gemm_only_from_heads.
lib_gemm_sites = None


def _note_lib_site(depth=2):
    if lib_gemm_sites is not None:
        import sys
        f = sys._getframe(depth)
        lib_gemm_sites[os.path.basename(f.f_code.co_filename) + ":" + f.f_code.co_name] = lib_gemm_sites.get(
            os.path.basename(f.f_code.co_filename) + ":" + f.f_code.co_name, 0) + 1


def lib_linear(x, w, b=None):
    _note_lib_site()
    return _LibLinear.apply(x, w, b)


# Weight-gradient GEMMs on a stream of their own: an EXPERIMENT, off unless FIBER_WGRAD_STREAM=1 (ops.enable_wgrad_stream(model) is
# called by bench.py and is a no-op without the variable).  dW is needed by nobody before the optimizer step (or DDP's bucket hook),
# while the next layer's backward only needs dX, so the TN kernel of layer l could share the GPU with the LayerNorm / attention
# backward of layer l - 1.  Round-6 result at B = 256, same box, interleaved: 265.1 / 266.7 ms with the second stream against 259.8 /
# 259.8 without -- the kernels contend more than they fill each other's gaps.  (A first, racy version read 239-246 ms: its garbage
# gradients turned every tensor into NaN after one optimizer step, and a GPU multiplying NaNs draws less power and clocks higher.)
# Who reads dW first is autograd's AccumulateGrad node: for gradients that come out of a Python Function it CLONES them (it never
# steals them: tools/probes/wgrad_stream_dbg5.py), on the stream that was current when the parameter's accumulator node was created.
# So the accumulator nodes of the model's parameters are created under the weight-gradient stream and kept alive (as DDP's reducer
# keeps them): the clone -- or the add into an existing .grad -- then runs on that stream, in order behind the TN kernel, the engine
# makes that stream wait for whatever produced the other gradients, and at the end of backward() the caller's stream waits for it.
_WGRAD_STREAM = [False]
_wg_streams = {}
_wg_pending = set()
_wg_accumulators = {}                # id(parameter) -> (weakref to it, its AccumulateGrad node)


def wgrad_stream(device):
    st = _wg_streams.get(device)
    if st is None:
        st = _wg_streams[device] = torch.cuda.Stream(device=device)
    return st


def enable_wgrad_stream(module, on=True):
    """Route the weight-gradient GEMMs of `module`'s parameters (and the accumulation of ALL its gradients) to a second stream.
    Call it after module.to(device) and BEFORE the first forward / DistributedDataParallel wrap: an accumulator node that already
    exists keeps the stream it was created on.  enable_wgrad_stream(module, False) switches the routing of the GEMMs off again."""
    _WGRAD_STREAM[0] = bool(on) and os.environ.get("FIBER_WGRAD_STREAM", "0") == "1"
    if not _WGRAD_STREAM[0]:
        return False
    for p_ in module.parameters():
        if not (p_.requires_grad and p_.is_cuda) or id(p_) in _wg_accumulators:
            continue
        with torch.cuda.stream(wgrad_stream(p_.device)):
            node = p_.view_as(p_).grad_fn.next_functions[0][0]      # the AccumulateGrad node: created here, on this stream
        _wg_accumulators[id(p_)] = (weakref.ref(p_, lambda _r, k=id(p_): _wg_accumulators.pop(k, None)), node)
    return True


def wgrad_stream_active(t):
    """inside a backward pass (grad mode off), not during graph capture, for GEMMs worth a hand-over"""
    return (_WGRAD_STREAM[0] and t.is_cuda and t.shape[0] >= 4096 and not torch.is_grad_enabled()
            and not torch.cuda.is_current_stream_capturing())


def set_wgrad_stream(on):
    """Raw switch (tests / probes): routes the GEMMs without touching any accumulator node."""
    _WGRAD_STREAM[0] = bool(on)


def join_wgrad_stream():
    """The current stream of every device with weight gradients in flight waits for them (DDP bucket hook; the autograd engine itself
    joins the accumulation streams at the end of backward())."""
    for dev in list(_wg_pending):
        torch.cuda.current_stream(dev).wait_stream(_wg_streams[dev])
    _wg_pending.clear()


# ---- deferred folds -----------------------------------------------------------------------------------------------------------------------
# The TN kernel splits its M reduction into S slabs and a small fold kernel sums them: 187 fold launches of ~14 us per step at the bench
# batch.  With ops.set_fold_defer(True) a weight gradient whose value nobody needs before the end of backward() (no `post`) leaves its
# slabs in the workspace and ONE multi-tensor launch folds them all from the autograd engine's end-of-backward callback.  Only valid
# when autograd takes the returned tensor over as .grad without reading it (zero_grad(set_to_none=True), one process: an existing .grad
# would be added to -- and DDP's reducer would copy it -- before the fold has run), hence opt-in.
# Round-6 result (same box, interleaved, B = 256): 267.2 / 262.0 ms per step deferred against 261.0 / 259.6 immediate -- a fold that runs
# right behind its GEMM reads the slabs out of the 256-MB Infinity Cache, the one launch at the end reads all 4 GB of them from HBM; at
# B = 32 (48.5-49.7 against 48.8-48.9 ms) the step is not launch-bound either.  So bench.py / Trainer call set_fold_defer, which stays a
# no-op unless FIBER_TN_FOLD_DEFER=1.
_FOLD_DEFER = [False]
_fold_pending = {}                   # device -> [(ws, dW pointer, db pointer, storage weak refs, S, N, K, stream)]
_fold_cb = [False]                   # an end-of-backward callback is queued


def set_fold_defer(on):
    _FOLD_DEFER[0] = bool(on) and os.environ.get("FIBER_TN_FOLD_DEFER", "0") == "1"


def _fold_defer_active(t):
    if not (_FOLD_DEFER[0] and t.is_cuda) or torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
        return False
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def flush_folds():
    """Fold every pending weight gradient (one launch per device) on the current stream, behind the streams their GEMMs ran on.
    Runs as the autograd engine's end-of-backward callback; the optimizer calls it as well (a backward that raised never ran its callbacks)."""
    _fold_cb[0] = False
    for dev, items in list(_fold_pending.items()):
        if not items:
            continue
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            rows, block0 = [], 0
            for ws, dwp, dbp, alive, S, N, K, st in items:
                # the outputs are NOT held here (a second reference would make autograd copy them instead of taking them over as .grad);
                # their storages must still be alive -- a gradient that was copied after all has been freed by now
                if any(w_.expired() for w_ in alive):
                    raise lib.FiberHipError("a weight gradient with a deferred fold was copied, not taken over, by autograd "
                                            "(ops.set_fold_defer needs .grad = None before backward, one process, no gradient hooks)")
                if st != cur:
                    cur.wait_stream(st)
                    ws.record_stream(cur)
                nk4 = N * K // 4
                rows.append((ws.data_ptr(), dwp, dbp, S | (N << 32), nk4 | (block0 << 32)))
                block0 += lib.plain("fiber_tn_fold_blocks", S, N, K, 1 if dbp else 0)
            table = torch.tensor(rows, dtype=torch.int64).to(dev)      # (blocking: the host staging tensor dies with this statement)
            lib.call("fiber_tn_fold_multi", lib.ptr(table), len(rows), block0)
        items.clear()
    _fold_pending.clear()


def wgrad(dh, x2, want_bias=False, row_mask=None, scale=1.0, post=None, row_map=None):
    """dW[N,K] = dh[M,N]^T . x2[M,K] in fp32 on the hand-written TN kernel (csrc/gemm_tn.hip): both operands are read as
    they lie (row-major, M slow) and transposed on the LDS -> register path; the M reduction is split inside the launch
    (fp32 slabs + one fold).  want_bias: also return the column sums of dh (the bias gradient) from the same pass.
    row_mask / scale: DropPath backward folded in -- samples whose factor in `row_mask` is 0 are skipped, the result is
    multiplied by `scale` (= 1/keep); see droppath_foldable().
    post(dw, db): anything the caller computes FROM the result (row permutations, reshapes, the LayerNorm unfolding of ops._LnMlp)
    -- it runs on the stream the kernel ran on, and its return value is returned instead of (dw, db).
    row_map (int32 [N], a permutation): row n of the result (entry n of the bias sums) is written at row_map[n]."""
    M, N = dh.shape
    K = x2.shape[1]
    assert dh.dtype == BF16 and x2.dtype == BF16 and dh.stride(1) == 1 and x2.stride(1) == 1 and x2.shape[0] == M
    dw = torch.empty((N, K), dtype=torch.float32, device=dh.device)
    db = torch.empty(N, dtype=torch.float32, device=dh.device) if want_bias else None
    S = lib.plain("fiber_gemm_tn_splits", M, N, K)
    ws = torch.empty(S * (N * K + N), dtype=torch.float32, device=dh.device) if (S > 1 or row_map is not None) else None   # (the row map is applied by the fold)
    rps = (M // row_mask.numel()) if row_mask is not None else 0
    args = (lib.ptr(dh), lib.ptr(x2), lib.ptr(dw), lib.ptr(db), lib.ptr(ws), M, N, K, dh.stride(0), x2.stride(0), lib.ptr(row_mask), rps,
            float(scale))
    finish = (lambda: post(dw, db)) if post is not None else (lambda: (dw, db) if want_bias else dw)
    if wgrad_stream_active(dh):
        cur = torch.cuda.current_stream(dh.device)
        side = wgrad_stream(dh.device)
        side.wait_stream(cur)                               # the operands' producers
        with torch.cuda.stream(side):
            if row_map is not None:
                lib.call("fiber_gemm_tn_rowmap_bf16", *args, lib.ptr(row_map))
            else:
                lib.call("fiber_gemm_tn_bf16", *args)
            out = finish()
        for t in (dh, x2, row_mask, ws, dw, db):            # memory handed back on `cur` must not be reused under the side stream
            if t is not None:
                t.record_stream(side)
        if not _wg_pending:
            torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)    # end of this backward pass
        _wg_pending.add(dh.device)
        return out
    if S > 1 and post is None and row_map is None and _fold_defer_active(dh):
        lib.call("fiber_gemm_tn_slabs_bf16", *args)
        if not _fold_cb[0]:
            torch.autograd.Variable._execution_engine.queue_callback(flush_folds)          # end of this backward pass
            _fold_cb[0] = True
        from torch.multiprocessing.reductions import StorageWeakRef
        alive = [StorageWeakRef(t.untyped_storage()) for t in (dw, db) if t is not None]
        _fold_pending.setdefault(dh.device, []).append((ws, dw.data_ptr(), db.data_ptr() if db is not None else 0, alive, S, N, K,
                                                        torch.cuda.current_stream(dh.device)))
        return finish()
    if row_map is not None:
        lib.call("fiber_gemm_tn_rowmap_bf16", *args, lib.ptr(row_map))
    else:
        lib.call("fiber_gemm_tn_bf16", *args)
    return finish()


def droppath_foldable(rows, rowscale, rs_value):
    """The DropPath backward factor s_b in {0, 1/keep} can ride in the consumers of the branch gradient instead of a pass of
    its own -- (s dY) W = s (dY W) in the dgrad epilogue, dW = (1/keep) * sum over kept samples in the weight-gradient kernel --
    when the caller knows 1/keep (`rs_value`) and a sample's rows are whole 64-row K tiles (true for Swin stages 0-2 at
    384^2: 9216 / 2304 / 576 tokens per image; stage 3 has 144)."""
    return (_DROPPATH_FOLD and rowscale is not None and rs_value and rows % rowscale.numel() == 0
            and (rows // rowscale.numel()) % 64 == 0)


_DROPPATH_FOLD = os.environ.get("FIBER_DROPPATH_FOLD", "1") != "0"      # A/B switch (tools): 0 = the separate s_b * dy pass



# Column sums that a backward kernel produced together with its output (window attention: the qkv bias gradient).  The
# producer leaves (data_ptr, shape, sums) here; the NEXT linear backward takes the slot -- and uses it only if it is about
# the very tensor it received as dY.  Every linear backward clears the slot, so it can never be matched against a later
# tensor that happens to reuse the address; a missed hand-over only costs the separate column-sum pass.
import threading

_hint_tls = threading.local()          # one hand-over slot per autograd thread (one per device): no cross-thread coupling


def _clear_colsum():
    _hint_tls.slot = None


def _offer_colsum(t2d, sums):
    """Called from inside a backward(): the offer never outlives the autograd pass it was made in."""
    _hint_tls.slot = (t2d.data_ptr(), tuple(t2d.shape), sums)
    torch.autograd.Variable._execution_engine.queue_callback(_clear_colsum)


def _take_colsum(t2d):
    h = getattr(_hint_tls, "slot", None)
    _hint_tls.slot = None
    if h is not None and h[0] == t2d.data_ptr() and h[1] == tuple(t2d.shape):
        return h[2]
    return None


def _dgrad(dh, weight):
    """dX = dY . W on the hand-written NT kernel: the B operand is the transposed bf16 working copy of W ([K, N], one small
    transposing kernel per weight and optimizer step), so the contraction is contiguous on both sides."""
    y, _ = gemm_nt(dh, bf16_weight_t(weight))
    return y


class _Linear(torch.autograd.Function):
    """Returns (y, y32): y32 is the fp32 payload of the output when `res32` (the fp32 payload of the residual) was given."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, rowscale, rs_value=None, res32=None):
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        wb = bf16_weight(weight)
        need_pre = bool(act) and any(ctx.needs_input_grad[:3])
        rps = (x2.shape[0] // rowscale.numel()) if rowscale is not None else 0
        oshp = (*shp[:-1], weight.shape[0])
        if res32 is not None:
            assert residual is not None and not act
            y, y32 = gemm_nt(x2, wb, bias, None, 0, False, rowscale, rps, res32=_c(res32).view(-1, weight.shape[0]))
            pre, y32 = None, y32.view(oshp)
            ctx.mark_non_differentiable(y32)
        else:
            r2 = _c(residual).view(-1, weight.shape[0]) if residual is not None else None
            y, pre = gemm_nt(x2, wb, bias, r2, act, need_pre, rowscale, rps)
            y32 = None
        ctx.save_for_backward(x2, weight, pre, rowscale)
        ctx.act, ctx.has_bias, ctx.has_res, ctx.shp, ctx.rs_value = act, bias is not None, residual is not None, shp, rs_value
        return y.view(oshp), y32

    @staticmethod
    def backward(ctx, dy, _dy32=None):
        x2, weight, pre, rowscale = ctx.saved_tensors
        dy2 = _c(dy).view(-1, weight.shape[0])
        dres = dy if ctx.has_res else None
        db = None
        hint = _take_colsum(dy2)
        fold = not ctx.act and dy2.shape[1] % 8 == 0 and droppath_foldable(dy2.shape[0], rowscale, ctx.rs_value)
        if rowscale is not None and not fold:         # branch gradient = per-sample scale * dy, as a pass of its own
            if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.act and dy2.shape[1] % 8 == 0:
                dy2, db = rowscale_colsum(dy2, rowscale)       # ... and the bias gradient from the same pass
            else:
                ds = torch.empty_like(dy2)
                lib.call("fiber_rowscale_add_bf16", None, lib.ptr(dy2), lib.ptr(rowscale), lib.ptr(ds), dy2.numel(),
                         dy2.numel() // rowscale.numel())
                dy2 = ds
        if ctx.act:
            dh, db = gelu_bwd_colsum(dy2, pre)
        else:
            dh = dy2
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        if fold:                                       # DropPath factor inside the dgrad epilogue and the weight-gradient kernel
            rps = dh.shape[0] // rowscale.numel()
            dx = gemm_nt(dh, bf16_weight_t(weight), None, None, 0, False, rowscale, rps)[0].view(ctx.shp) if ctx.needs_input_grad[0] else None
            dw = db = None
            if ctx.needs_input_grad[1]:
                dw = wgrad(dh, x2, want_bias=need_db, row_mask=rowscale, scale=ctx.rs_value)
                if need_db:
                    dw, db = dw
            elif need_db:
                db = rowscale_colsum(dh, rowscale)[1]
            return dx, dw, db, dres, None, None, None, None
        dx = _dgrad(dh, weight).view(ctx.shp) if ctx.needs_input_grad[0] else None
        if need_db and db is None and hint is not None and dh is dy2:
            db = hint                                  # produced by the kernel that wrote dy (window attention backward)
        want = need_db and db is None                  # otherwise the bias gradient rides in the weight-gradient pass
        dw = None
        if ctx.needs_input_grad[1]:
            dw = wgrad(dh, x2, want_bias=want)
            if want:
                dw, db = dw
        elif want:
            db = colsum(dh)
        if not need_db:
            db = None
        return dx, dw, db, dres, None, None, None, None


def linear(x, weight, bias=None, residual=None, act=None, rowscale=None, rowscale_value=None):
    """nn.Linear with fused bias / exact GELU / residual, forward and both backward GEMMs on the hand-written kernels.
    A library GEMM is used only for shapes the tile kernels do not cover (N % 8 != 0, e.g. the 2-way ITM head, or K % 8 != 0)."""
    N, K = weight.shape
    if (N % 8) or (K % 8):                                 # no backbone linear has such a shape; odd test / head shapes only
        _note_lib_site()
        y = _LibLinear.apply(x, weight.to(BF16), bias.to(BF16) if bias is not None else None)
        if act:
            y = torch.nn.functional.gelu(y)
        assert rowscale is None
        return y + residual if residual is not None else y
    y, y32 = _Linear.apply(x, weight, bias, residual, 1 if act else 0, rowscale, rowscale_value, f32_of(residual))
    return with_f32(y, y32)


class _MLP(torch.autograd.Function):
    """y = [rowscale *] (fc2(gelu(fc1(x)))) + residual  (timm Mlp inside a Swin block / RoBERTa intermediate+output).

    Forward: two MFMA GEMMs (bias+GELU epilogue saving the pre-activation; bias+DropPath scale+residual epilogue).
    Backward: dH = (dY . W2) * gelu'(H) is ONE hand-written GEMM whose epilogue applies the GELU derivative to the saved
    pre-activation (and the DropPath factor of the branch) -- dG never exists in HBM and the separate gelu_bwd pass over the
    4C-wide tensor disappears.  dX = dH . W1 runs on the same NT kernel family (transposed working copy of W1), dW1 / dW2 and
    both bias gradients on the TN kernel (csrc/gemm_tn.hip); no library GEMM is involved."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, rowscale, rs_value=None, res32=None):
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        need_bwd = any(ctx.needs_input_grad[:5])            # inference (no_grad / frozen): the pre-activation copy is not stored
        g, h = gemm_nt(x2, bf16_weight(w1), b1, None, 1, need_bwd)
        rps = (x2.shape[0] // rowscale.numel()) if rowscale is not None else 0
        oshp = (*shp[:-1], w2.shape[0])
        if res32 is not None:                         # fp32 residual stream: (bf16 shadow, fp32 payload)
            y, y32 = gemm_nt(g, bf16_weight(w2), b2, None, 0, False, rowscale, rps, res32=_c(res32).view(-1, w2.shape[0]))
            y32 = y32.view(oshp)
            ctx.mark_non_differentiable(y32)
        else:
            r2 = _c(residual).view(-1, w2.shape[0]) if residual is not None else None
            y, _ = gemm_nt(g, bf16_weight(w2), b2, r2, 0, False, rowscale, rps)
            y32 = None
        if need_bwd:
            ctx.save_for_backward(x2, w1, w2, h, g, rowscale)
        ctx.has_res, ctx.shp, ctx.rs_value = residual is not None, shp, rs_value
        return y.view(oshp), y32

    @staticmethod
    def backward(ctx, dy, _dy32=None):
        dres = dy if ctx.has_res else None
        if not ctx.saved_tensors:                      # frozen block on a trainable stream: only the residual needs a gradient
            return (None,) * 5 + (dres, None, None, None)
        x2, w1, w2, h, g, rowscale = ctx.saved_tensors
        dy2 = _c(dy).view(-1, w2.shape[0])
        db2 = None
        C, C4 = dy2.shape[1], h.shape[1]
        fused = C % 64 == 0 and C4 % 8 == 0 and _FUSED_MLP_BWD != "0"
        fold = fused and dy2.shape[1] % 8 == 0 and droppath_foldable(dy2.shape[0], rowscale, ctx.rs_value)
        rs_arg, rps, mask, scale = None, 0, None, 1.0
        if fold:                                       # DropPath factor rides in the gelu' GEMM epilogue and the wgrad kernel
            rs_arg, rps, mask, scale = rowscale, dy2.shape[0] // rowscale.numel(), rowscale, ctx.rs_value
        elif rowscale is not None:
            if dy2.shape[1] % 8 == 0:
                dy2, db2 = rowscale_colsum(dy2, rowscale)      # DropPath backward + fc2 bias gradient in one pass
            else:
                ds = torch.empty_like(dy2)
                lib.call("fiber_rowscale_add_bf16", None, lib.ptr(dy2), lib.ptr(rowscale), lib.ptr(ds), dy2.numel(),
                         dy2.numel() // rowscale.numel())
                dy2 = ds
        if fused:
            dh, _ = gemm_nt(dy2, bf16_weight_t(w2), None, None, 2, False, rs_arg, rps, aux=h)
            db1 = None
        else:                                         # shapes the DMA kernel does not cover (e.g. Swin-T C=96), or FIBER_FUSED_MLP_BWD=0
            dh, db1 = gelu_bwd_colsum(_dgrad(dy2, w2), h)
        if db2 is None:
            dw2, db2 = wgrad(dy2, g, want_bias=True, row_mask=mask, scale=scale)
        else:
            dw2 = wgrad(dy2, g)
        dx = _dgrad(dh, w1).view(ctx.shp)
        if db1 is None:
            dw1, db1 = wgrad(dh, x2, want_bias=True)   # fc1 bias gradient = column sums of dH, from the same pass
        else:
            dw1 = wgrad(dh, x2)
        return dx, dw1, db1, dw2, db2, dres, None, None, None


def mlp(x, w1, b1, w2, b2, residual=None, rowscale=None, rowscale_value=None):
    y, y32 = _MLP.apply(x, w1, b1, w2, b2, residual, rowscale, rowscale_value, f32_of(residual))
    return with_f32(y, y32)


# ---- LayerNorm + Mlp + DropPath + residual as one kernel per direction (csrc/mlp_rows.hip; Swin stages with C = 128 / 256) -------------
_LN_MLP = os.environ.get("FIBER_LN_MLP", "1") != "0"          # A/B switch: 0 = layernorm_res + mlp (separate kernels)
_fa_perms = {}


def _fa_perm(K, device):
    """Column order of the kernel's weight copies: position p holds logical index p with bits 2 and 3 swapped (mlp_rows.hip)."""
    key = (K, device)
    if key not in _fa_perms:
        i = torch.arange(K, device=device)
        _fa_perms[key] = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)
    return _fa_perms[key]


def _fa(m16):
    return m16[:, _fa_perm(m16.shape[1], m16.device)].contiguous()


def _ln_mlp_weights(gamma, beta, w1, b1, w2):
    """(w1p, b1p, w2p, w2tp, w1tp): LayerNorm's affine part folded into fc1 (W1' = W1 diag(gamma), b1' = b1 + W1 beta, formed in
    fp32, ONE bf16 rounding); the copies whose K dimension is the hidden one (w2p, w1tp) in the kernel's K order; rebuilt when any of
    the five parameters changes."""
    key = ("LNMLP", id(w1))
    stamp = tuple(_stamp(t) for t in (gamma, beta, w1, b1, w2))
    hit = _cache_get(key, w1)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        w1f = w1.detach().float()
        w1p16 = (w1f * gamma.detach().float()[None, :]).to(BF16)
        b1p = torch.addmv(b1.detach().float(), w1f, beta.detach().float()).contiguous()
        w2_16 = w2.detach().to(BF16)
        val = (w1p16.contiguous(), b1p, _fa(w2_16), w2_16.t().contiguous(), _fa(w1p16.t()))
    _cache_put(key, stamp, val, w1)
    return val


# widths the fused kernels are USED at (they exist for 128 and 256; FIBER_LN_MLP_WIDTHS=128,256 for A/B runs).  Measured at 512 images,
# forward + backward (tools/lnmlp_bench.py, gpurun_out/r06_lnmlp_bench_5.log): C = 128 9.24 ms against 11.11 ms for the separate kernels,
# C = 256 6.68 against 6.12 (one wave per SIMD at that width: nothing runs under its GELU arithmetic) -- so 128 only.
_LN_MLP_WIDTHS = tuple(int(v) for v in os.environ.get("FIBER_LN_MLP_WIDTHS", "128").split(",") if v)


def ln_mlp_eligible(x, C):
    return _LN_MLP and C in _LN_MLP_WIDTHS and C in (128, 256) and x.is_cuda and x.dtype == BF16 and f32_of(x) is None


class _LnMlp(torch.autograd.Function):
    """y = x + rowscale * Mlp(LN(x))  (swin_transformer.py:391) on csrc/mlp_rows.hip: the forward reads x, writes y and (when a
    backward will follow) gelu(H) -- no LayerNorm output, no pre-activation in HBM; the backward recomputes the pre-activation from
    x, returns dx (LayerNorm backward and residual gradient included) and hands dH / xhat to the fc1 weight-gradient GEMM.
    The LayerNorm's gamma / beta ride in the fc1 weight copy, so their gradients come out of dW1' and db1:
    dW1 = dW1' diag(gamma) + db1 (x) beta, dgamma = colsum(dW1' * W1), dbeta = W1^T db1."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w1, b1, w2, b2, rowscale, rs_value):
        shp = x.shape
        C = shp[-1]
        x2 = _c(x).view(-1, C)
        M = x2.shape[0]
        w1p, b1p, w2p, _, _ = _ln_mlp_weights(gamma, beta, w1, b1, w2)
        y = torch.empty_like(x2)
        need_bwd = any(ctx.needs_input_grad[:8])
        g = torch.empty((M, 4 * C), dtype=BF16, device=x2.device) if need_bwd else None    # gelu(H): the fc2 weight gradient's operand
        rps = M // rowscale.numel() if rowscale is not None else 0
        lib.call("fiber_ln_mlp_fwd_bf16", lib.ptr(x2), lib.ptr(w1p), lib.ptr(b1p), lib.ptr(w2p), lib.ptr(b2.detach()), lib.ptr(rowscale),
                 lib.ptr(y), lib.ptr(g), M, C, rps, float(eps))
        if need_bwd:
            ctx.save_for_backward(x2, gamma, beta, w1, b1, w2, rowscale, g)
        ctx.eps, ctx.shp, ctx.rs_value = float(eps), shp, rs_value
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, beta, w1, b1, w2, rowscale, g = ctx.saved_tensors
        M, C = x2.shape
        H = 4 * C
        dy2 = _c(dy).view(M, C)
        w1p, b1p, _, w2tp, w1tp = _ln_mlp_weights(gamma, beta, w1, b1, w2)
        dx = torch.empty_like(x2)
        dh = torch.empty((M, H), dtype=BF16, device=x2.device)
        xhat = torch.empty_like(x2)
        rps = M // rowscale.numel() if rowscale is not None else 0
        lib.call("fiber_ln_mlp_bwd_bf16", lib.ptr(x2), lib.ptr(dy2), lib.ptr(w1p), lib.ptr(b1p), lib.ptr(w2tp), lib.ptr(w1tp),
                 lib.ptr(rowscale), lib.ptr(dx), lib.ptr(dh), lib.ptr(xhat), M, C, rps, ctx.eps)
        w1f, gf, bf_ = w1.detach().float(), gamma.detach().float(), beta.detach().float()

        def unfold(dw1p, db1):      # gradients of LayerNorm's gamma / beta and of W1 out of dW1' (xn = xhat gamma + beta feeds fc1)
            return (dw1p * w1f).sum(0), torch.mv(w1f.t(), db1), torch.addcmul(torch.outer(db1, bf_), dw1p, gf[None, :]), db1
        dgamma, dbeta, dw1, db1 = wgrad(dh, xhat, want_bias=True, post=unfold)
        del dh, xhat
        if rowscale is None:
            dw2, db2 = wgrad(dy2, g, want_bias=True)
        elif droppath_foldable(M, rowscale, ctx.rs_value):
            dw2, db2 = wgrad(dy2, g, want_bias=True, row_mask=rowscale, scale=ctx.rs_value)
        else:
            dys, db2 = rowscale_colsum(dy2, rowscale)
            dw2 = wgrad(dys, g)
        return dx.view(ctx.shp), dgamma, dbeta, None, dw1, db1, dw2, db2, None, None


def ln_mlp(x, gamma, beta, eps, w1, b1, w2, b2, rowscale=None, rowscale_value=None):
    """x + DropPath(Mlp(LayerNorm(x))) of a Swin block; the fused kernels when ln_mlp_eligible(x, C), else layernorm_res + mlp."""
    if ln_mlp_eligible(x, x.shape[-1]) and w1.shape[0] == 4 * x.shape[-1]:
        return _LnMlp.apply(x, gamma, beta, eps, w1, b1, w2, b2, rowscale, rowscale_value)
    v, r = layernorm_res(x, gamma, beta, eps)
    return mlp(v, w1, b1, w2, b2, residual=r, rowscale=rowscale, rowscale_value=rowscale_value)


def _ln_fwd(x2, x32, gamma, beta, eps, want_f32):
    """LayerNorm rows of x2 (bf16) or of its fp32 payload x32 -> (y bf16, y32 or None, mean, rstd)."""
    rows, C = x2.shape
    y = torch.empty((rows, C), dtype=BF16, device=x2.device)
    y32 = torch.empty((rows, C), dtype=torch.float32, device=x2.device) if want_f32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty_like(mean)
    if x32 is None and not want_f32:
        lib.call("fiber_layernorm_fwd_bf16", lib.ptr(x2), lib.ptr(gamma), lib.ptr(beta), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd), rows, C, eps)
    else:
        src = x32 if x32 is not None else x2
        lib.call("fiber_layernorm_fwd_stream", lib.ptr(src), lib.ptr(gamma), lib.ptr(beta), lib.ptr(y), lib.ptr(y32), lib.ptr(mean),
                 lib.ptr(rstd), rows, C, eps, 1 if x32 is not None else 0)
    return y, y32, mean, rstd


def _ln_bwd(dy2, xs, is32, gamma, mean, rstd, dr2):
    """LayerNorm backward from the saved input xs (bf16, or fp32 when is32); dr2: residual-path gradient added into dx."""
    rows, C = dy2.shape
    dx = torch.empty((rows, C), dtype=BF16, device=dy2.device)
    dg = torch.empty(C, dtype=torch.float32, device=dy2.device)
    db = torch.empty_like(dg)
    ws = torch.empty(lib.plain("fiber_layernorm_bwd_grid", rows) * 8 * C, dtype=torch.float32, device=dy2.device)
    if is32:
        lib.call("fiber_layernorm_bwd_stream", lib.ptr(dy2), lib.ptr(xs), lib.ptr(gamma), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dr2),
                 lib.ptr(dx), lib.ptr(dg), lib.ptr(db), lib.ptr(ws), rows, C, 1)
    else:
        lib.call("fiber_layernorm_bwd_bf16", lib.ptr(dy2), lib.ptr(xs), lib.ptr(gamma), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dr2),
                 lib.ptr(dx), lib.ptr(dg), lib.ptr(db), lib.ptr(ws), rows, C)
    return dx, dg, db


class _LayerNorm(torch.autograd.Function):
    """(LN(x), fp32 copy of it or None).  x32: fp32 payload of x (the fp32 residual stream); want_f32: also return the output in
    fp32 (post-LN text stack: the LayerNorm output is the next residual)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, x32=None, want_f32=False):
        C = x.shape[-1]
        x2 = _c(x).view(-1, C)
        xs = _c(x32).view(-1, C) if x32 is not None else None
        y, y32, mean, rstd = _ln_fwd(x2, xs, gamma, beta, eps, want_f32)
        ctx.save_for_backward(xs if xs is not None else x2, gamma, mean, rstd)
        ctx.is32 = xs is not None
        if y32 is not None:
            y32 = y32.view(x.shape)
            ctx.mark_non_differentiable(y32)
        return y.view(x.shape), y32

    @staticmethod
    def backward(ctx, dy, _dy32=None):
        xs, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = _ln_bwd(_c(dy).view(xs.shape), xs, ctx.is32, gamma, mean, rstd, None)
        return dx.view(dy.shape), dg, db, None, None, None


def layernorm(x, gamma, beta, eps=1e-5, want_f32=False):
    """LayerNorm over the last dim.  Reads the fp32 payload of a stream tensor when it has one.  want_f32 attaches the output's own fp32
    payload: only where the OUTPUT starts or continues a residual stream (patch-embedding norm, the post-LN text stack: those callers ask
    for it).  The pre-LN consumers and the final norms do not -- a payload nothing reads as a residual is an fp32 copy written for nothing,
    and it would make downstream ops treat their input as a stream."""
    x32 = f32_of(x)
    y, y32 = _LayerNorm.apply(x, gamma, beta, eps, x32, bool(want_f32))
    return with_f32(y, y32)


class _LayerNormRes(torch.autograd.Function):
    """(LN(x), x): the second output is x itself, to be used for the residual connection.  Its incoming gradient is
    added inside the LayerNorm backward kernel, so autograd never launches a separate activation-sized add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, x32=None):
        C = x.shape[-1]
        x2 = _c(x).view(-1, C)
        xs = _c(x32).view(-1, C) if x32 is not None else None
        y, _, mean, rstd = _ln_fwd(x2, xs, gamma, beta, eps, False)
        ctx.save_for_backward(xs if xs is not None else x2, gamma, mean, rstd)
        ctx.is32 = xs is not None
        return y.view(x.shape), x2.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dres):
        xs, gamma, mean, rstd = ctx.saved_tensors
        rows, C = xs.shape
        if dy is None:
            return dres, None, None, None, None
        dr2 = _c(dres).view(rows, C) if dres is not None else None
        dx, dg, db = _ln_bwd(_c(dy).view(rows, C), xs, ctx.is32, gamma, mean, rstd, dr2)
        return dx.view(dy.shape), dg, db, None, None


def layernorm_res(x, gamma, beta, eps=1e-5):
    """Returns (LN(x), x_for_residual); the residual keeps x's fp32 payload."""
    x32 = f32_of(x)
    y, r = _LayerNormRes.apply(x, gamma, beta, eps, x32)
    return y, with_f32(r, x32)


class _PatchMergeLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, H, W, eps, x32=None):
        B, L, C = x.shape
        xs = _c(x32) if x32 is not None else _c(x)
        rows = B * (H // 2) * (W // 2)
        y = torch.empty((B, rows // B, 4 * C), dtype=BF16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if x32 is not None:
            lib.call("fiber_patch_merge_ln_fwd_stream", lib.ptr(xs), lib.ptr(gamma), lib.ptr(beta), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd),
                     B, H, W, C, eps, 1)
        else:
            lib.call("fiber_patch_merge_ln_fwd_bf16", lib.ptr(xs), lib.ptr(gamma), lib.ptr(beta), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd), B, H, W, C, eps)
        ctx.save_for_backward(xs, gamma, mean, rstd)
        ctx.dims, ctx.is32 = (B, H, W, C), x32 is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, gamma, mean, rstd = ctx.saved_tensors
        B, H, W, C = ctx.dims
        rows = B * (H // 2) * (W // 2)
        dy = _c(dy)
        dx = torch.empty(xs.shape, dtype=BF16, device=dy.device)
        dg = torch.empty(4 * C, dtype=torch.float32, device=dy.device)
        db = torch.empty_like(dg)
        ws = torch.empty(lib.plain("fiber_layernorm_bwd_grid", rows) * 8 * 4 * C, dtype=torch.float32, device=dy.device)
        if ctx.is32:
            lib.call("fiber_patch_merge_ln_bwd_stream", lib.ptr(dy), lib.ptr(xs), lib.ptr(gamma), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dx),
                     lib.ptr(dg), lib.ptr(db), lib.ptr(ws), B, H, W, C, 1)
        else:
            lib.call("fiber_patch_merge_ln_bwd_bf16", lib.ptr(dy), lib.ptr(xs), lib.ptr(gamma), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dx),
                     lib.ptr(dg), lib.ptr(db), lib.ptr(ws), B, H, W, C)
        return dx, dg, db, None, None, None, None


def patch_merge_ln(x, gamma, beta, H, W, eps=1e-5):
    return _PatchMergeLN.apply(x, gamma, beta, H, W, eps, f32_of(x))


class _WindowAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias_table, B, H, W, heads, ws, shift, head_major):
        C = qkv.shape[-1] // 3
        qkv = _c(qkv)
        rows = B * H * W
        o = torch.empty((rows, C), dtype=BF16, device=qkv.device)
        lse = torch.empty((rows, heads), dtype=torch.float32, device=qkv.device)
        lib.call("fiber_window_attn_fwd_bf16", lib.ptr(qkv), lib.ptr(bias_table), lib.ptr(o), lib.ptr(lse), B, H, W, C, heads, ws, shift, head_major)
        ctx.save_for_backward(qkv, bias_table, o, lse)
        ctx.dims = (B, H, W, C, heads, ws, shift, head_major)
        return o.view(B, H * W, C)

    @staticmethod
    def backward(ctx, do):
        qkv, bias_table, o, lse = ctx.saved_tensors
        B, H, W, C, heads, ws, shift, head_major = ctx.dims
        do = _c(do)
        rows, N = B * H * W, ws * ws
        dqkv = torch.empty_like(qkv)
        dtab = torch.empty_like(bias_table)
        delta = torch.empty((rows, heads), dtype=torch.float32, device=do.device)
        nz = lib.plain("fiber_window_attn_bwd_slices", rows // N, heads)
        part = torch.empty(nz * heads * N * N, dtype=torch.float32, device=do.device)
        # Column sums of dqkv (= bias gradient of the qkv linear) CAN come out of the same passes (round 2: -5 ms per step against a pass of
        # their own over dqkv).  Round 4: they cost the window backward 6-12 % (tools/win_colsum_probe.py: stage 2 780 -> 730 us, stage 0
        # 2950 -> 2610), while the weight-gradient kernel of the qkv linear now takes its bias sums for ~1 % -- so by default the qkv linear's
        # backward asks that kernel (no offer is made); FIBER_WIN_COLSUM=1 restores the in-kernel sums.
        cs_rows = lib.plain("fiber_window_attn_colsum_rows", rows // N, heads, ws) if _WIN_COLSUM else 0
        csum = torch.empty(3 * C, dtype=torch.float32, device=do.device) if cs_rows else None
        cs_ws = torch.empty(cs_rows * 3 * C, dtype=torch.float32, device=do.device) if cs_rows else None
        lib.call("fiber_window_attn_bwd_bf16", lib.ptr(qkv), lib.ptr(bias_table), lib.ptr(o), lib.ptr(do), lib.ptr(lse), lib.ptr(dqkv),
                 lib.ptr(dtab), lib.ptr(delta), lib.ptr(part), lib.ptr(csum), lib.ptr(cs_ws), B, H, W, C, heads, ws, shift, head_major)
        if csum is not None:
            _offer_colsum(dqkv.view(-1, 3 * C), csum)
        return dqkv, dtab, None, None, None, None, None, None, None


def window_attention(qkv, bias_table, B, H, W, heads, ws, shift, head_major=False):
    """qkv [B, H*W, 3C] in image-token order -> attention output [B, H*W, C] (shift/partition/reverse folded in).
    head_major: qkv channels are [heads][3][32] (see linear_qkv_head_major) instead of the reference [3][heads][32]."""
    return _WindowAttn.apply(qkv, bias_table, B, H, W, heads, ws, shift, int(head_major))


def head_major_supported(ws):
    """The specialised window kernels (N = ws*ws <= 160) accept the head-major qkv layout."""
    return ws * ws <= 160


_perm_cache = {}


def _qkv_perm(C, heads, device):
    """perm[r'] = r: row r' = h*96 + which*32 + d of the head-major projection is row r = which*C + h*32 + d of qkv.weight."""
    key = (C, heads, str(device))
    if key not in _perm_cache:
        h, which, d = torch.meshgrid(torch.arange(heads), torch.arange(3), torch.arange(32), indexing="ij")
        perm = (which * C + h * 32 + d).reshape(-1).to(device)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(3 * C, device=device)
        _perm_cache[key] = (perm, inv)
    return _perm_cache[key]


_TN_ROWMAP = os.environ.get("FIBER_TN_ROWMAP", "1") != "0"


def _qkv_perm32(C, heads, device):
    key = ("i32", C, heads, str(device))
    if key not in _perm_cache:
        _perm_cache[key] = _qkv_perm(C, heads, device)[0].to(torch.int32).contiguous()
    return _perm_cache[key]


class _LinearQKVHeadMajor(torch.autograd.Function):
    """qkv = x . W^T + b with the OUTPUT channels reordered to [heads][3][32]: q|k|v of one head become one contiguous
    192-byte run per token, so the per-head gathers of the window-attention kernels use 3/4 of every cache line they touch
    instead of 1/2.  Only the bf16 working copy of the weight is permuted; parameters / gradients keep the reference layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, heads):
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        C = weight.shape[1]
        perm, inv = _qkv_perm(C, heads, x.device)
        key = ("HM", id(weight))
        hit = _cache_get(key, weight)
        if hit is None or hit[0] != (_stamp(weight), _stamp(bias)):
            wp = weight.detach()[perm].to(BF16).contiguous()
            _cache_put(key, (_stamp(weight), _stamp(bias)), (wp, bias.detach()[perm].contiguous(), wp.t().contiguous(), weakref.ref(bias), heads), weight)
            _hm_table.clear()                                # (new storage: the one-launch refresh rebuilds its descriptor table)
        wp, bp = _wcache[key][1][:2]
        y, _ = gemm_nt(x2, wp, bp)
        ctx.save_for_backward(x2, weight)
        ctx.shp, ctx.heads = shp, heads
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = _c(dy).view(-1, weight.shape[0])
        hint = _take_colsum(dy2)
        perm, inv = _qkv_perm(weight.shape[1], ctx.heads, dy.device)
        wp, _, wpt = _wcache[("HM", id(weight))][1][:3]
        dx = gemm_nt(dy2, wpt)[0].view(ctx.shp)
        if _TN_ROWMAP:                                      # row r' of the head-major gradient is written at row perm[r'] by the kernel
            pmap = _qkv_perm32(weight.shape[1], ctx.heads, dy.device)
            if hint is not None:
                dw, db = wgrad(dy2, x2, row_map=pmap), hint[inv]
            else:
                dw, db = wgrad(dy2, x2, want_bias=True, row_map=pmap)
        elif hint is not None:                             # (A/B: FIBER_TN_ROWMAP=0 -- index kernels behind the GEMM)
            dw, db = wgrad(dy2, x2, post=lambda w_, _b: w_[inv]), hint[inv]
        else:
            dw, db = wgrad(dy2, x2, want_bias=True, post=lambda w_, b_: (w_[inv], b_[inv]))
        return dx, dw, db, None


def linear_qkv_head_major(x, weight, bias, heads):
    return _LinearQKVHeadMajor.apply(x, weight, bias, heads)


# ---- packed projections ---------------------------------------------------------------------------------------------------------
# q / k / v of RobertaSelfAttention (roberta.py:256-290) and key / value of the t2i cross-attention are separate nn.Linear modules
# (separate state-dict keys) applied to the SAME input.  Here they run as ONE GEMM: the bf16 working copies of the member weights
# are row slices of one [sum N, K] buffer (registered in the weight cache as the members' own working copies, so the fused
# optimizer kernel rewrites the packed buffer in place), its transposed copy is one entry of the multi-tensor transpose, and the
# member BIASES (fp32 parameters) are re-pointed at slices of one fp32 buffer (same values, same Parameter objects, shared
# storage) -- no per-step gather of either.  One forward GEMM, one dgrad, one wgrad (+ fold) instead of three each, and no
# autograd fan-in adds of the three input gradients.
_packs = {}
_NO_PACK = os.environ.get("FIBER_NO_PACK", "0") == "1"      # A/B and debugging: separate GEMMs + cat


def _pack_views_ok(pk, ws):
    off = 0
    for w, n in zip(ws, pk["Ns"]):
        hit = _cache_get(id(w), w)
        if hit is None or hit[1].data_ptr() != pk["plain"].data_ptr() + off * pk["K"] * 2:
            return False
        off += n
    return True


def _pack_get(weights, biases):
    key = tuple(id(w) for w in weights)
    pk = _packs.get(key)
    dev = weights[0].device
    if pk is None or pk["plain"].device != dev or any(r() is not w for r, w in zip(pk["refs"], weights)):
        Ns, K = [w.shape[0] for w in weights], weights[0].shape[1]
        pk = {"plain": torch.empty((sum(Ns), K), dtype=BF16, device=dev), "t": torch.empty((K, sum(Ns)), dtype=BF16, device=dev),
              "refs": [weakref.ref(w, lambda _r, k=key: _packs.pop(k, None)) for w in weights], "Ns": Ns, "K": K, "t_stamp": None,
              "bias": None}
        _packs[key] = pk
        _t_registry_version[0] += 1
    if not _pack_views_ok(pk, weights):                     # (re-)register the members' working copies as views of the pack
        off = 0
        for w, n in zip(weights, pk["Ns"]):
            view = pk["plain"][off:off + n]
            view.copy_(w.detach())
            _cache_put(id(w), _stamp(w), view, w)
            off += n
        pk["t_stamp"] = None
    else:
        for w in weights:                                   # stale member (parameter changed outside the fused optimizer)
            if _cache_get(id(w), w)[0] != _stamp(w):
                bf16_weight(w)                              # in-place refresh of the view
                pk["t_stamp"] = None
    stamps = tuple(_stamp(w) for w in weights)
    if pk["t_stamp"] != stamps:
        pk["t"].copy_(pk["plain"].t())
        pk["t_stamp"] = stamps
    if biases[0] is not None:
        bp, off, ok = pk["bias"], 0, pk["bias"] is not None
        for b, n in zip(biases, pk["Ns"]):
            ok = ok and b.data_ptr() == bp.data_ptr() + off * 4
            off += n
        if not ok:                                          # first use / the parameters moved (model.to, load with assign)
            if any(b.dtype != torch.float32 for b in biases):   # the members' .data is re-pointed at the pack: it must keep their dtype
                raise TypeError("linear_packed: biases must be fp32 masters (got " + ", ".join(str(b.dtype) for b in biases) + ")")
            with torch.no_grad():
                bp = torch.cat([b.detach().reshape(-1) for b in biases])
                off = 0
                for b, n in zip(biases, pk["Ns"]):
                    if b.is_cuda:
                        # the cat above may still be QUEUED on this (second) stream when the old storage -- allocated on the default
                        # stream -- is dropped here: without this the other stream's next allocation reuses and overwrites it first
                        b.data.record_stream(torch.cuda.current_stream(b.device))
                    b.data = bp[off:off + n]                # same values, shared storage: optimizer updates land in the pack
                    off += n
            pk["bias"] = bp
    return pk


class _LinearPacked(torch.autograd.Function):
    """fork: the node also returns an ALIAS of its input for the input's other consumer (the image block that follows the text layer whose t2i
    keys / values this projection forms).  The other consumer's gradient then arrives HERE instead of at autograd's fan-in, and the dX GEMM adds it
    in its residual epilogue: one read of that gradient instead of a three-tensor `add` pass over [B*L, C] (8 of them per step, 0.9 GB each at stage 2)."""

    @staticmethod
    def forward(ctx, x, n, fork, *wb):
        weights, biases = wb[0::2], wb[1::2]
        pk = _pack_get(weights, biases)
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        y, _ = gemm_nt(x2, pk["plain"], pk["bias"])
        ctx.save_for_backward(x2)
        ctx.pk, ctx.shp, ctx.has_bias = pk, shp, biases[0] is not None
        y = y.view(*shp[:-1], pk["plain"].shape[0])
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        (x2,) = ctx.saved_tensors
        pk = ctx.pk
        dy2 = _c(dy).view(-1, pk["plain"].shape[0])
        _take_colsum(dy2)
        dx = None
        if ctx.needs_input_grad[0]:
            if dalias is not None:
                da2 = _c(dalias).view(-1, ctx.shp[-1])
                ones = _ones_rows(dy2.device)
                dx = gemm_nt(dy2, pk["t"], None, da2, rowscale=ones, rows_per_sample=dy2.shape[0])[0].view(ctx.shp)   # dY W + the alias' gradient
            else:
                dx = gemm_nt(dy2, pk["t"])[0].view(ctx.shp)
        out = wgrad(dy2, x2, want_bias=ctx.has_bias)
        dw, db = out if ctx.has_bias else (out, None)
        grads, off = [], 0
        for n in pk["Ns"]:
            grads += [dw[off:off + n], db[off:off + n] if db is not None else None]
            off += n
        return (dx, None, None, *grads)


_ones_cache = {}


def _ones_rows(device):
    t = _ones_cache.get(device)
    if t is None:
        t = _ones_cache[device] = torch.ones(1, dtype=torch.float32, device=device)
    return t


def linear_packed(x, linears, fork_sink=None):
    """[x W0^T + b0 | x W1^T + b1 | ...] for nn.Linear-like (weight, bias) pairs that share the input: one GEMM (see above).
    Falls back to separate GEMMs + cat for shapes the tile kernels do not cover."""
    weights = [w for w, _ in linears]
    biases = [b for _, b in linears]
    K = weights[0].shape[1]
    if (_NO_PACK or any(w.shape[1] != K or w.shape[0] % 8 for w in weights) or K % 8
            or any((b is None) != (biases[0] is None) for b in biases)):
        return torch.cat([linear(x, w, b) for w, b in linears], dim=-1)
    flat = [t for pair in zip(weights, biases) for t in pair]
    if fork_sink is not None and torch.is_grad_enabled() and x.requires_grad and f32_of(x) is None and x.dtype == BF16:
        y, alias = _LinearPacked.apply(x, len(weights), True, *flat)
        fork_sink.append(alias)
        return y
    return _LinearPacked.apply(x, len(weights), False, *flat)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


class _MHA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kmask, B, heads, scale, p_drop, seed):
        # q [B*Lq, heads*D], k/v [B*Lk, heads*D] (possibly column views of a packed projection)
        D = q.shape[1] // heads
        Lq, Lk = q.shape[0] // B, k.shape[0] // B
        o = torch.empty((q.shape[0], heads * D), dtype=BF16, device=q.device)
        lse = torch.empty((q.shape[0], heads), dtype=torch.float32, device=q.device)
        lib.call("fiber_mha_fwd_bf16", lib.ptr(q), lib.ptr(k), lib.ptr(v), lib.ptr(kmask), lib.ptr(o), lib.ptr(lse), B, heads, Lq, Lk, D,
                 _ld(q), _ld(k), _ld(v), _ld(o), scale, p_drop, seed, seed_base_ptr())
        ctx.save_for_backward(q, k, v, kmask, o, lse)
        ctx.cfg = (B, heads, Lq, Lk, D, scale, p_drop, seed, seed_base_ptr())
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, kmask, o, lse = ctx.saved_tensors
        B, heads, Lq, Lk, D, scale, p_drop, seed, base = ctx.cfg
        do = _c(do)
        dq = torch.empty((q.shape[0], heads * D), dtype=BF16, device=q.device)
        dk = torch.empty((k.shape[0], heads * D), dtype=BF16, device=q.device)
        dv = torch.empty_like(dk)
        delta = torch.empty((q.shape[0], heads), dtype=torch.float32, device=q.device)
        lib.call("fiber_mha_bwd_bf16", lib.ptr(q), lib.ptr(k), lib.ptr(v), lib.ptr(kmask), lib.ptr(o), lib.ptr(do), lib.ptr(lse),
                 lib.ptr(dq), lib.ptr(dk), lib.ptr(dv), lib.ptr(delta), B, heads, Lq, Lk, D, _ld(q), _ld(k), _ld(v), _ld(o), _ld(do),
                 _ld(dq), _ld(dk), _ld(dv), scale, p_drop, seed, base)
        return dq, dk, dv, None, None, None, None, None, None


def mha(q, k, v, kmask, B, heads, scale, p_drop=0.0, seed=0):
    """softmax(q.k^T*scale + kmask).v per (sample, head); q/k/v are 2-D [B*L, heads*D] bf16, kmask fp32 [B, Lk] or None."""
    if kmask is not None:
        kmask = _c(kmask.view(B, -1).float())
    return _MHA.apply(q, k, v, kmask, B, heads, float(scale), float(p_drop), int(seed))


class _MHAPacked(torch.autograd.Function):
    """mha() on packed projections: mode 0: `a` = [q | k | v] ([B*L, 3*heads*D], self-attention), mode 1: `a` = q, `b` = [k | v].
    The backward kernel writes dq / dk / dv straight into the column blocks of ONE packed gradient per input, so the projection's
    backward is one dgrad + one wgrad and autograd never assembles slices."""

    @staticmethod
    def forward(ctx, a, b, kmask, mode, B, heads, scale, p_drop, seed):
        a = _c(a)
        b = _c(b) if b is not None else None
        if mode == 0:
            C = a.shape[1] // 3
            q, k, v = a[:, :C], a[:, C:2 * C], a[:, 2 * C:]
        else:
            C = a.shape[1]
            q, k, v = a, b[:, :C], b[:, C:]
        D = C // heads
        Lq, Lk = q.shape[0] // B, k.shape[0] // B
        o = torch.empty((q.shape[0], C), dtype=BF16, device=a.device)
        lse = torch.empty((q.shape[0], heads), dtype=torch.float32, device=a.device)
        base = seed_base_ptr()
        lib.call("fiber_mha_fwd_bf16", lib.ptr(q), lib.ptr(k), lib.ptr(v), lib.ptr(kmask), lib.ptr(o), lib.ptr(lse), B, heads, Lq, Lk, D,
                 _ld(q), _ld(k), _ld(v), _ld(o), scale, p_drop, seed, base)
        ctx.save_for_backward(a, b, kmask, o, lse)
        ctx.cfg = (mode, B, heads, Lq, Lk, D, C, scale, p_drop, seed, base)
        return o

    @staticmethod
    def backward(ctx, do):
        a, b, kmask, o, lse = ctx.saved_tensors
        mode, B, heads, Lq, Lk, D, C, scale, p_drop, seed, base = ctx.cfg
        do = _c(do)
        da = torch.empty_like(a)
        db = torch.empty_like(b) if b is not None else None
        if mode == 0:
            q, k, v = a[:, :C], a[:, C:2 * C], a[:, 2 * C:]
            dq, dk, dv = da[:, :C], da[:, C:2 * C], da[:, 2 * C:]
        else:
            q, k, v = a, b[:, :C], b[:, C:]
            dq, dk, dv = da, db[:, :C], db[:, C:]
        delta = torch.empty((q.shape[0], heads), dtype=torch.float32, device=a.device)
        lib.call("fiber_mha_bwd_bf16", lib.ptr(q), lib.ptr(k), lib.ptr(v), lib.ptr(kmask), lib.ptr(o), lib.ptr(do), lib.ptr(lse),
                 lib.ptr(dq), lib.ptr(dk), lib.ptr(dv), lib.ptr(delta), B, heads, Lq, Lk, D, _ld(q), _ld(k), _ld(v), _ld(o), _ld(do),
                 _ld(dq), _ld(dk), _ld(dv), scale, p_drop, seed, base)
        return da, db, None, None, None, None, None, None, None


def mha_qkv_packed(qkv, kmask, B, heads, scale, p_drop=0.0, seed=0):
    """Self-attention on a packed projection qkv [B*L, 3*heads*D] = [q | k | v] (linear_packed)."""
    if kmask is not None:
        kmask = _c(kmask.view(B, -1).float())
    return _MHAPacked.apply(qkv, None, kmask, 0, B, heads, float(scale), float(p_drop), int(seed))


def mha_kv_packed(q, kv, kmask, B, heads, scale, p_drop=0.0, seed=0):
    """Cross-attention with the key / value projections packed: q [B*Lq, heads*D], kv [B*Lk, 2*heads*D] = [k | v]."""
    if kmask is not None:
        kmask = _c(kmask.view(B, -1).float())
    return _MHAPacked.apply(q, kv, kmask, 1, B, heads, float(scale), float(p_drop), int(seed))


class _ScaleAdd(torch.autograd.Function):
    """out = a + alpha * b with a learnable scalar alpha (alpha_i2t / alpha_t2i)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        lib.call("fiber_scale_add_bf16", lib.ptr(a), lib.ptr(b), lib.ptr(alpha), 1.0, lib.ptr(out), a.numel())
        ctx.save_for_backward(b, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, alpha = ctx.saved_tensors
        dout = _c(dout)
        db = torch.empty_like(b)
        lib.call("fiber_scale_add_bf16", None, lib.ptr(dout), lib.ptr(alpha), 1.0, lib.ptr(db), dout.numel())
        dalpha = torch.zeros(1, dtype=torch.float32, device=dout.device)
        lib.call("fiber_dot_bf16", lib.ptr(dout), lib.ptr(b), lib.ptr(dalpha), dout.numel())
        return dout, db, dalpha


def scale_add(a, b, alpha):
    return _ScaleAdd.apply(a, b, alpha)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        lib.call("fiber_scale_add_bf16", lib.ptr(a), lib.ptr(b), None, 1.0, lib.ptr(out), a.numel())
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout, dout


def add(a, b):
    """Residual add of two same-shape bf16 tensors."""
    return _Add.apply(a, b)


class _StreamAdd(torch.autograd.Function):
    """(out16, out32) = res + rowscale[sample] * (drop_a(a) + alpha * drop_b(b))  -- csrc/elementwise.hip stream_add_kernel."""

    @staticmethod
    def forward(ctx, res, a, b, alpha, rowscale, p_a, seed_a, p_b, seed_b, res32, want32):
        a = _c(a)
        b = _c(b) if b is not None else None
        kind, rsrc = 0, None
        if res32 is not None:
            kind, rsrc = 2, _c(res32)
        elif res is not None:
            kind, rsrc = 1, _c(res)
        out16 = torch.empty_like(a)
        out32 = torch.empty(a.shape, dtype=torch.float32, device=a.device) if want32 else None
        per = (a.numel() // rowscale.numel()) if rowscale is not None else 0
        base = seed_base_ptr()
        lib.call("fiber_stream_add", lib.ptr(rsrc), kind, lib.ptr(a), lib.ptr(b), lib.ptr(alpha), lib.ptr(rowscale), per, float(p_a), int(seed_a),
                 float(p_b), int(seed_b), base, lib.ptr(out32), lib.ptr(out16), a.numel())
        ctx.save_for_backward(b if (alpha is not None and b is not None) else None, alpha, rowscale)
        ctx.cfg = (per, float(p_a), int(seed_a), float(p_b), int(seed_b), base, res is not None, b is not None)
        if out32 is not None:
            ctx.mark_non_differentiable(out32)
        return out16, out32

    @staticmethod
    def backward(ctx, dy, _d32=None):
        b, alpha, rowscale = ctx.saved_tensors
        per, p_a, seed_a, p_b, seed_b, base, has_res, has_b = ctx.cfg
        dy = _c(dy)
        plain_a = rowscale is None and p_a == 0.0                       # d a = dy itself
        da = dy if plain_a else torch.empty_like(dy)
        db = torch.empty_like(dy) if has_b else None
        dalpha = torch.zeros(1, dtype=torch.float32, device=dy.device) if (alpha is not None and has_b and ctx.needs_input_grad[3]) else None
        if not plain_a or has_b:
            lib.call("fiber_stream_add_bwd", lib.ptr(dy), lib.ptr(b), lib.ptr(alpha), lib.ptr(rowscale), per, p_a, seed_a, p_b, seed_b, base,
                     None if plain_a else lib.ptr(da), lib.ptr(db), lib.ptr(dalpha), dy.numel())
        return (dy if has_res else None), da, db, dalpha, None, None, None, None, None, None, None


def stream_add(res, a, b=None, alpha=None, rowscale=None, p_a=0.0, p_b=0.0, training=True):
    """res + rowscale * (dropout_pa(a) + alpha * dropout_pb(b)) in one pass.  `res` may be a stream pair (its fp32 payload is read
    and the result is a pair again), a plain bf16 tensor, or None (with residual_fp32() a NEW stream starts when res is None
    only if the caller attaches it).  Dropout keys are drawn here (one per active dropout, a before b)."""
    p_a = float(p_a) if training else 0.0
    p_b = float(p_b) if (training and b is not None) else 0.0
    seed_a = next_seed() if p_a > 0 else 0
    seed_b = next_seed() if p_b > 0 else 0
    r32 = f32_of(res)
    want32 = r32 is not None or (_RESIDUAL_FP32[0] and res is not None)
    y, y32 = _StreamAdd.apply(res, a, b, alpha, rowscale, p_a, seed_a, p_b, seed_b, r32, want32)
    return with_f32(y, y32)


def stream_bf16(x):
    """The bf16 tensor a GEMM may read for a stream tensor: the shadow itself (every producer on the stream writes one)."""
    return x


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(x)
        y = torch.empty_like(x)
        base = seed_base_ptr()
        lib.call("fiber_dropout_bf16", lib.ptr(x), lib.ptr(y), x.numel(), p, seed, base)
        ctx.cfg = (p, seed, base)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        lib.call("fiber_dropout_bf16", lib.ptr(dy), lib.ptr(dx), dy.numel(), *ctx.cfg)
        return dx, None, None


# Dropout / DropPath keys.  key = BASE(seed, step) + call-site counter, where the counter restarts at every training step:
#   eager:  the whole key is passed to the kernels by value;
#   graph:  (enable_graph_rng) BASE lives in a device int64 scalar that set_rng_step() rewrites before every replay, the
#           kernels receive only the counter by value + the pointer -- a captured step draws new masks on every replay and
#           the SAME masks an eager run of that step would draw.
_seed_state = {"seed": 0x5EED, "ctr": 0, "step": None, "base_dev": None}
_M64 = (1 << 64) - 1


def _base_value():
    hi, step = _seed_state["seed"], _seed_state["step"]
    if step is None:
        return (hi << 32) & _M64
    hi = (hi ^ ((int(step) >> 12) * 0x9E3779B1)) & 0xFFFFFFFF
    return ((hi << 32) | ((int(step) & 0xFFF) << 20)) & _M64


def _write_base_dev():
    t = _seed_state["base_dev"]
    if t is not None:
        v = _base_value()
        t.fill_(v - (1 << 64) if v >= (1 << 63) else v)    # the bit pattern, through torch's signed int64


def manual_seed(seed):
    """Seed of the dropout / DropPath streams (lightning.seed_everything and the Trainer call this with config["seed"] + rank)."""
    _seed_state["seed"], _seed_state["ctr"], _seed_state["step"] = int(seed) & 0xFFFFFFFF, 0, None
    _seed_state["collate_ctr"] = 0
    _write_base_dev()


def set_rng_step(step):
    """Called once per training step with the optimizer-step index: the per-call-site counter restarts inside a window of
    2^20 keys owned by that step, so a run resumed at step s draws the masks of step s, not those of step 0."""
    if _seed_state["step"] != step:
        _seed_state["step"] = step
        _seed_state["ctr"] = 0
        if _seed_state["base_dev"] is not None and not torch.cuda.is_current_stream_capturing():
            _write_base_dev()


def enable_graph_rng(device):
    """Keep the per-step part of every dropout key in device memory (see above); returns the int64 scalar."""
    if _seed_state["base_dev"] is None or _seed_state["base_dev"].device != torch.device(device):
        _seed_state["base_dev"] = torch.zeros((), dtype=torch.int64, device=device)
    _write_base_dev()
    return _seed_state["base_dev"]


def disable_graph_rng():
    _seed_state["base_dev"] = None


def seed_base_ptr():
    t = _seed_state["base_dev"]
    return None if t is None else t.data_ptr()


def next_seed():
    """The by-value part of a fresh 64-bit dropout key for one call site (forward and backward share it through ctx)."""
    _seed_state["ctr"] += 1
    if _seed_state["base_dev"] is not None:
        return _seed_state["ctr"]
    return (_base_value() + _seed_state["ctr"]) & _M64


def collate_seed():
    """A fresh by-value 64-bit key for the input pipeline (MLM masking in data.device_collate).  The collate runs OUTSIDE the
    training step (and outside a captured hipGraph), before set_rng_step() of the step that will consume the batch, so it has its
    own stream: (seed, batch counter) in a key range no dropout site uses -- never the device-resident base, whose by-value part
    restarts every step (the same tokens would be masked on every replay)."""
    _seed_state["collate_ctr"] = _seed_state.get("collate_ctr", 0) + 1
    hi = (_seed_state["seed"] ^ 0xC011A7E5) & 0xFFFFFFFF
    return ((hi << 32) | (_seed_state["collate_ctr"] & 0xFFFFFFFF)) & _M64


def collate_counter():
    """Position of the collate key stream (saved in checkpoints so a resumed run does not replay the masks of batches 1..k)."""
    return int(_seed_state.get("collate_ctr", 0))


def set_collate_counter(n):
    _seed_state["collate_ctr"] = int(n)


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p), next_seed())


class _RowScaleAdd(torch.autograd.Function):
    """out = resid + scale[b] * x  (per-sample DropPath on a residual branch)."""

    @staticmethod
    def forward(ctx, resid, x, scale):
        resid, x = _c(resid), _c(x)
        out = torch.empty_like(x)
        per = x.numel() // x.shape[0]
        lib.call("fiber_rowscale_add_bf16", lib.ptr(resid), lib.ptr(x), lib.ptr(scale), lib.ptr(out), x.numel(), per)
        ctx.save_for_backward(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        (scale,) = ctx.saved_tensors
        dout = _c(dout)
        dx = torch.empty_like(dout)
        per = dout.numel() // dout.shape[0]
        lib.call("fiber_rowscale_add_bf16", None, lib.ptr(dout), lib.ptr(scale), lib.ptr(dx), dout.numel(), per)
        return dout, dx, None


def drop_path_scale(batch, p, device):
    """timm 0.4.12 DropPath factor per sample: floor(keep + U[0,1)) / keep (fp32 [batch]), U from the path's counter-based
    key stream (csrc/elementwise.hip) -- no host generator, so the draw is capturable in a hipGraph."""
    out = torch.empty(batch, dtype=torch.float32, device=device)
    lib.call("fiber_droppath_scale_f32", lib.ptr(out), batch, 1.0 - p, next_seed(), seed_base_ptr())
    return out


def rowscale_add(resid, x, scale):
    return _RowScaleAdd.apply(resid, x, scale)


def drop_path_add(resid, x, p, training):
    """resid + DropPath_p(x): timm 0.4.12 per-sample mask floor(keep + U[0,1)) scaled by 1/keep."""
    if not training or p <= 0.0:
        return add(resid, x)
    return _RowScaleAdd.apply(resid, x, drop_path_scale(x.shape[0], p, x.device))


class _RobertaEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, word, pos_tab, type_tab, gamma, beta, pad, eps, p_drop, seed):
        B, S = ids.shape
        C = word.shape[1]
        ids = _c(ids)
        y = torch.empty((B, S, C), dtype=BF16, device=ids.device)
        pos = torch.empty((B, S), dtype=torch.int32, device=ids.device)
        mean = torch.empty(B * S, dtype=torch.float32, device=ids.device)
        rstd = torch.empty_like(mean)
        lib.call("fiber_roberta_embed_fwd", lib.ptr(ids), lib.ptr(word), lib.ptr(pos_tab), lib.ptr(type_tab), lib.ptr(gamma), lib.ptr(beta),
                 lib.ptr(y), lib.ptr(pos), lib.ptr(mean), lib.ptr(rstd), B, S, C, pad, eps, p_drop, seed, seed_base_ptr())
        ctx.save_for_backward(ids, pos, word, pos_tab, type_tab, gamma, mean, rstd)
        ctx.cfg = (B, S, C, pad, p_drop, seed, seed_base_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, pos, word, pos_tab, type_tab, gamma, mean, rstd = ctx.saved_tensors
        B, S, C, pad, p_drop, seed, base = ctx.cfg
        dy = _c(dy)
        dword, dpos, dtype = torch.zeros_like(word), torch.zeros_like(pos_tab), torch.zeros_like(type_tab)
        dg = torch.zeros(C, dtype=torch.float32, device=dy.device)
        db = torch.zeros_like(dg)
        lib.call("fiber_roberta_embed_bwd", lib.ptr(dy), lib.ptr(ids), lib.ptr(pos), lib.ptr(word), lib.ptr(pos_tab), lib.ptr(type_tab),
                 lib.ptr(gamma), lib.ptr(mean), lib.ptr(rstd), lib.ptr(dword), lib.ptr(dpos), lib.ptr(dtype), lib.ptr(dg), lib.ptr(db),
                 B, S, C, pad, p_drop, seed, base)
        return None, dword, dpos, dtype, dg, db, None, None, None, None


def roberta_embed(ids, word, pos_tab, type_tab, gamma, beta, pad=1, eps=1e-5, p_drop=0.0, training=False):
    p = float(p_drop) if training else 0.0
    return _RobertaEmbed.apply(ids, word, pos_tab, type_tab, gamma, beta, pad, eps, p, next_seed() if p > 0 else 0)


class ImagePair:
    """The image batch of the one-pass MLM + ITM step, [img ; where(sel, img, alt)] (objectives.compute_mlm_itm_fused), as its two sources:
    the patch embedding gathers its im2col rows from them (fiber_im2col_patch4_pair), so neither the `where` nor the `cat` pass over the fp32
    images runs.  `tensor()` builds the batch for any other consumer."""

    def __init__(self, img, alt, sel):
        assert img.shape == alt.shape and sel.numel() == img.shape[0]
        self.img, self.alt, self.sel = img, alt, sel.reshape(-1).to(torch.bool)
        self.shape = torch.Size((2 * img.shape[0],) + tuple(img.shape[1:]))
        self.device, self.dtype, self.is_cuda = img.device, img.dtype, img.is_cuda

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def tensor(self):
        return torch.cat([self.img, torch.where(self.sel.view(-1, 1, 1, 1), self.img, self.alt)], 0)


class _PatchEmbedProj(torch.autograd.Function):
    """Conv2d(3->C, k=4, s=4) + bias as im2col + MFMA GEMM (no input gradient: the image is data)."""

    @staticmethod
    def forward(ctx, img, weight, bias, pair=None):
        Cout = weight.shape[0]
        if pair is not None:
            B, _, H, W = pair.shape
            rows = B * (H // 4) * (W // 4)
            a, b, sel = _c(pair.img.float()), _c(pair.alt.float()), pair.sel.to(torch.uint8)
            cols = torch.empty((rows, 64), dtype=BF16, device=a.device)
            lib.call("fiber_im2col_patch4_pair", lib.ptr(a), lib.ptr(b), lib.ptr(sel), lib.ptr(cols), B // 2, H, W)
            img = a
        else:
            B, _, H, W = img.shape
            img = _c(img.float())
            rows = B * (H // 4) * (W // 4)
            cols = torch.empty((rows, 64), dtype=BF16, device=img.device)
            lib.call("fiber_im2col_patch4", lib.ptr(img), lib.ptr(cols), B, H, W)
        key = ("pe", id(weight))
        hit = _cache_get(key, weight)
        if hit is None or hit[0] != _stamp(weight):
            wp = torch.zeros((Cout, 64), dtype=BF16, device=img.device)
            wp[:, :48] = weight.detach().reshape(Cout, 48).to(BF16)
            _cache_put(key, _stamp(weight), wp, weight)
        wp = _wcache[key][1]
        y, _ = gemm_nt(cols, wp, bias)
        ctx.save_for_backward(cols)
        ctx.wshape = weight.shape
        return y.view(B, rows // B, Cout)

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        dy2 = _c(dy).view(-1, dy.shape[-1])
        def crop(w_, b_):                                   # a fresh tensor, not a view of a temporary: autograd takes it over as .grad
            out = torch.empty(ctx.wshape, dtype=w_.dtype, device=w_.device)
            out.view(w_.shape[0], 48).copy_(w_[:, :48])
            return out, b_
        dw, db = wgrad(dy2, cols, want_bias=True, post=crop)
        return None, dw, db, None


def patch_embed_proj(img, weight, bias):
    if isinstance(img, ImagePair):
        if img.is_cuda:
            return _PatchEmbedProj.apply(None, weight, bias, img)
        img = img.tensor()
    return _PatchEmbedProj.apply(img, weight, bias)


# ---------------------------------------------------------------------------------------------------------------------------
# Modulated deformable convolution (DyHead of the fine-grained model): csrc/dcn.hip gather / scatter around the
# hand-written NT / TN GEMMs.  Channels-last throughout.
def _conv_weight_rows(weight, transposed=False):
    """[Cout, Cin, kh, kw] fp32 parameter -> bf16 [Cp, kh*kw*Cin] in the tap-major column order of the gather kernel (rows padded
    with zeros to a multiple of 8 -- the 27-channel offset convolution), or its transpose [kh*kw*Cin, Cp] for the dgrad GEMM."""
    key = ("KCT" if transposed else "KC", id(weight))
    hit = _cache_get(key, weight)
    if hit is not None and hit[0] == _stamp(weight):
        return hit[1]
    if transposed:
        v = _conv_weight_rows(weight).t().contiguous()
    else:
        Cout = weight.shape[0]
        v = weight.detach().permute(0, 2, 3, 1).reshape(Cout, -1).to(BF16)
        if Cout % 8:
            v = torch.nn.functional.pad(v, (0, 0, 0, 8 - Cout % 8))
        v = v.contiguous()
    _cache_put(key, _stamp(weight), v, weight)
    return v


def _conv_out(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


class _DeformConv(torch.autograd.Function):
    """y[B,Ho,Wo,Cout] = conv(x[B,H,W,Cin] sampled at taps + offset, * mask) + bias.  offset [M, 2*taps] / mask [M, taps] fp32 or
    None (ordinary convolution).  Forward: one gather launch + one MFMA GEMM over the whole batch; backward: dgrad GEMM -> one
    scatter launch (dx, doffset, dmask), weight + bias gradient on the TN kernel."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, pad, out_fp32=False):
        B, H, W, C = x.shape
        Cout, _, kh, kw = weight.shape
        Ho, Wo = _conv_out(H, kh, stride, pad), _conv_out(W, kw, stride, pad)
        x = _c(x)
        offset = _c(offset.float()) if offset is not None else None
        mask = _c(mask.float()) if mask is not None else None
        M = B * Ho * Wo
        cols = torch.empty((M, kh * kw * C), dtype=BF16, device=x.device)
        lib.call("fiber_dcn_gather_bf16", lib.ptr(x), lib.ptr(offset), lib.ptr(mask), lib.ptr(cols), B, H, W, C, Ho, Wo, kh, kw, stride, pad)
        wr = _conv_weight_rows(weight)
        Cp = wr.shape[0]
        bp = bias
        if bias is not None and Cp != Cout:
            bp = torch.nn.functional.pad(bias.detach().float(), (0, Cp - Cout))
        y, _ = gemm_nt(cols, wr, bp, out_fp32=out_fp32)
        if Cp != Cout:
            y = y[:, :Cout].contiguous()
        ctx.save_for_backward(x, offset, mask, cols, weight)
        ctx.geom = (B, H, W, C, Ho, Wo, kh, kw, stride, pad, Cout, Cp)
        ctx.has_bias = bias is not None
        return y.view(B, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, offset, mask, cols, weight = ctx.saved_tensors
        B, H, W, C, Ho, Wo, kh, kw, stride, pad, Cout, Cp = ctx.geom
        dy2 = _c(dy.to(BF16)).view(-1, Cout)
        if Cp != Cout:
            dy2 = torch.nn.functional.pad(dy2, (0, Cp - Cout))
        dx = doff = dmask = dw = db = None
        need_x, need_off, need_mask = ctx.needs_input_grad[0], offset is not None and ctx.needs_input_grad[1], mask is not None and ctx.needs_input_grad[2]
        if need_x or need_off or need_mask:
            dcols, _ = gemm_nt(dy2, _conv_weight_rows(weight, transposed=True))
            tiled = need_x and kh == 3 and kw == 3 and pad == 1 and stride in (1, 2) and C % 16 == 0      # atomics-free input gradient
            dxf = torch.zeros((B, H, W, C), dtype=torch.float32, device=x.device) if (need_x and not tiled) else None
            doff = torch.empty_like(offset) if need_off else None
            dmask = torch.empty_like(mask) if need_mask else None
            if dxf is not None or need_off or need_mask:
                lib.call("fiber_dcn_scatter_bf16", lib.ptr(dcols), lib.ptr(x), lib.ptr(offset), lib.ptr(mask), lib.ptr(dxf), lib.ptr(doff),
                         lib.ptr(dmask), B, H, W, C, Ho, Wo, kh, kw, stride, pad)
            if tiled:
                n = lib.plain("fiber_dcn_dx_workspace", B, H, W, C, Ho, Wo, stride)
                ws = torch.empty(n, dtype=torch.float32, device=x.device)
                ws[n - B * H * W * C - 4:].zero_()
                dx = torch.empty((B, H, W, C), dtype=BF16, device=x.device)
                lib.call("fiber_dcn_dx_bf16", lib.ptr(dcols), lib.ptr(offset), lib.ptr(mask), lib.ptr(dx), lib.ptr(ws), B, H, W, C, Ho, Wo,
                         kh, kw, stride, pad)
            elif need_x:
                dx = dxf.to(BF16)
        need_db = ctx.has_bias and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[3]:
            dw, db = wgrad(dy2, cols, want_bias=need_db, post=lambda w_, b_: (
                w_[:Cout].view(Cout, kh, kw, C).permute(0, 3, 1, 2).contiguous(), b_[:Cout] if b_ is not None else None))
        elif need_db:
            db = colsum(dy2)[:Cout]
        return dx, doff, dmask, dw, db, None, None, None


def deform_conv(x, offset, mask, weight, bias=None, stride=1, pad=1, out_fp32=False):
    """Channels-last modulated deformable convolution; offset = mask = None gives the ordinary convolution.  out_fp32: the result
    is stored in fp32 (the offset / mask predictor: the reference runs the whole operator in fp32, deform_conv.py:335, and sampling
    positions rounded to bf16 move every bilinear weight by up to 2^-9 of the offset)."""
    return _DeformConv.apply(x, offset, mask, weight, bias, stride, pad, out_fp32)
