"""Where the bench step goes, per (C-ABI entry point, shape): every launch of a single-stream step is bracketed by HIP events
(one event after each launch; on one stream consecutive events bracket one entry point's kernels), aggregated over a few steps.

For the GEMM entry points the row carries the algorithmic FLOPs and bytes, so the table gives achieved TFLOP/s and TB/s per shape --
the per-shape view rocprofv3's per-kernel-name statistics cannot give (one kernel name serves many shapes).  bench.py reads the
committed result (the latest profiles/rNN_shape_breakdown.json) to pick the dominant kernel it times live.

    FIBER_NO_OVERLAP=1 python tools/shape_breakdown.py [batch=256] [steps=3]   ->  gpurun_out/shape_breakdown.json
"""
import ctypes as C
import json
import os
import sys

os.environ.setdefault("FIBER_NO_OVERLAP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from fiber_amd import lib, ops, parallel
from fiber_amd.config import named_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils


def gemm_row(name, sc):
    """(flops, algorithmic bytes, label) of a GEMM launch from its scalar arguments."""
    if name == "fiber_gemm_nt_bf16":
        rps, ldaux, M, N, K, ldx, ldw, ldy, ldr, act = sc
        mode = act & 0xff
        by = 2 * (M * K + N * K + M * N)
        kind = "plain/bias"
        if act & 0x800:
            by += 4 * M * N + 4 * M * N          # fp32 residual in, fp32 sum out (+ the bf16 shadow counted above)
            kind = "fp32-residual"
        elif ldr:
            by += 2 * M * N
            kind = "residual" + ("+droppath" if rps else "")
        if mode == 1:
            by += 2 * M * N
            kind = "gelu+pre"
        if mode == 2:
            by += 2 * M * N
            kind = "gelu'*aux" + ("+droppath" if rps else "")
        return 2.0 * M * N * K, by, f"NT {kind}", (M, N, K)
    if name in ("fiber_gemm_tn_bf16", "fiber_gemm_tn_rowmap_bf16"):
        M, N, K, lddy, ldx, rps, scale = sc
        return 2.0 * M * N * K, 2 * M * (N + K) + 4 * N * K, "TN wgrad" + ("+droppath" if rps else ""), (M, N, K)
    return None


def stream_row(name, sc, nn):
    """(flops, algorithmic bytes, label) of a non-GEMM launch: `sc` its scalar arguments, `nn` = which pointer arguments were non-NULL
    (in signature order).  Bytes = every tensor read or written once (round 4: so that the per-shape roofline table covers the
    streaming kernels too)."""
    if name == "fiber_layernorm_fwd_bf16":
        rows, C = sc[0], sc[1]
        return 0.0, 4 * rows * C, "LayerNorm fwd"
    if name == "fiber_layernorm_bwd_bf16":
        rows, C = sc[0], sc[1]
        dres = nn[5]
        return 0.0, (8 if dres else 6) * rows * C, "LayerNorm bwd" + (" + residual grad" if dres else "")
    if name in ("fiber_window_attn_fwd_bf16", "fiber_window_attn_bwd_bf16"):
        Bn, H, W, C, heads, ws = sc[:6]
        T, N = Bn * H * W, ws * ws
        if name.endswith("fwd_bf16"):
            return 4.0 * T * N * C, 8 * T * C, f"window attention fwd ws{ws}"
        return 10.0 * T * N * C, 16 * T * C, f"window attention bwd ws{ws}"
    if name in ("fiber_mha_fwd_bf16", "fiber_mha_bwd_bf16"):
        Bn, heads, Lq, Lk, D = sc[:5]
        e = Bn * heads * D * 2
        if name.endswith("fwd_bf16"):
            return 4.0 * Bn * heads * Lq * Lk * D, e * (2 * Lq + 2 * Lk), f"attention fwd {Lq}x{Lk} d{D}"
        return 10.0 * Bn * heads * Lq * Lk * D, e * (4 * Lq + 4 * Lk), f"attention bwd {Lq}x{Lk} d{D}"
    if name in ("fiber_ln_mlp_fwd_bf16", "fiber_ln_mlp_bwd_bf16"):
        M, C = sc[0], sc[1]
        if name.endswith("fwd_bf16"):                      # x in, y out, G out (when a backward follows: nn[7]); two products
            return 2.0 * M * C * 4 * C * 2, 2 * (2 * M * C + (4 * M * C if nn[7] else 0)), f"LN+MLP fused fwd [{M}, {C}]"
        return 2.0 * M * C * 4 * C * 3, 2 * (4 * M * C + 4 * M * C), f"LN+MLP fused bwd [{M}, {C}]"    # x, dy in; dx, xhat, dh out
    if name == "fiber_stream_add":
        n, res_kind = sc[-1], sc[0]
        by = (2 if res_kind == 1 else 4 if res_kind == 2 else 0) + 2 + (2 if nn[2] else 0) + (4 if nn[6] else 0) + (2 if nn[7] else 0)
        return 0.0, by * n, "residual glue fwd"
    if name == "fiber_stream_add_bwd":
        n = sc[-1]
        by = 2 + (2 if nn[1] else 0) + 2 + (2 if nn[6] else 0)
        return 0.0, by * n, "residual glue bwd"
    return None


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    lib.load()
    torch.manual_seed(0)
    ops.manual_seed(0)
    cfg = named_config("task_pretrain_mlm_itm", per_gpu_batchsize=B, num_gpus=1, max_steps=100000, warmup_steps=10000, draw_false_image=1)
    model = FIBERTransformerSS(cfg)
    for n, p in model.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)
    parallel.freeze_unused(model, model.unused_parameter_names())
    model.to(dev).train()
    fiber_utils.set_task(model)
    (opt,), (sched,) = model.configure_optimizers()
    batch = bench.synth_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], dev, seed=0)

    log = []                                       # (name, scalar args, event after the launch)
    orig_call = lib.call
    recording = [False]

    def traced(name, *args):
        orig_call(name, *args)
        if recording[0]:
            sig = lib.SIGNATURES[name]
            sc = tuple(a for a, t in zip(args, sig) if t is not lib.P)
            nn = tuple(a is not None for a, t in zip(args, sig) if t is lib.P)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            log.append((name, sc, e, nn))

    lib.call = traced
    ops.lib.call = traced

    def step():
        ops.set_rng_step(model.global_step)
        out = model(batch)
        loss = sum(v for k, v in out.items() if "loss" in k)
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        opt.zero_grad(set_to_none=True)
        model.global_step += 1

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    agg = {}
    total_ms = 0.0
    for _ in range(steps):
        log.clear()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        recording[0] = True
        step()
        recording[0] = False
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        torch.cuda.synchronize()
        total_ms += e0.elapsed_time(e1) / steps
        prev = e0
        for name, sc, e, nn in log:
            # time since the previous traced launch: this entry point's kernels plus whatever ATen work ran in between
            ms = prev.elapsed_time(e)
            prev = e
            if name.startswith("fiber_mha_"):
                sc = sc[:-1] + (0,)                  # the dropout seed differs per call: not part of the shape
            key = (name, sc, nn if name in ("fiber_layernorm_bwd_bf16", "fiber_stream_add", "fiber_stream_add_bwd", "fiber_ln_mlp_fwd_bf16") else ())
            a = agg.setdefault(key, [0.0, 0])
            a[0] += ms / steps
            a[1] += 1
    rows = []
    for (name, sc, nn), (ms, n) in agg.items():
        calls = n / steps
        r = {"entry": name, "args": list(sc), "ms_per_step": round(ms, 3), "calls_per_step": round(calls, 1), "us_per_call": round(ms / calls * 1e3, 1)}
        g = gemm_row(name, sc)
        st = None if g else stream_row(name, list(sc), nn if nn else tuple(True for _ in range(16)))
        if st:
            fl, by, kind = st
            r.update(kind=kind, TFLOPs=round(fl * calls / ms / 1e9, 1), TBps=round(by * calls / ms / 1e9, 3), flop_per_call=fl, bytes_per_call=by)
        if g:
            fl, by, kind, shp = g
            r.update(kind=kind, shape=list(shp), TFLOPs=round(fl * calls / ms / 1e9, 1), TBps=round(by * calls / ms / 1e9, 3),
                     flop_per_call=fl, bytes_per_call=by)
        rows.append(r)
    rows.sort(key=lambda r: -r["ms_per_step"])
    by_entry = {}
    for r in rows:
        by_entry[r["entry"]] = round(by_entry.get(r["entry"], 0.0) + r["ms_per_step"], 2)
    out = {"batch": B, "steps": steps, "single_stream_step_ms": round(total_ms, 2),
           "note": "ms attributed to an entry point = time since the previous traced launch (includes untraced ATen kernels in between)",
           "by_entry_ms": dict(sorted(by_entry.items(), key=lambda kv: -kv[1])), "rows": rows}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/shape_breakdown.json", "w"), indent=1)
    print(f"single-stream step {total_ms:.1f} ms at B={B}")
    for k, v in list(out["by_entry_ms"].items())[:14]:
        print(f"  {k:34s} {v:8.2f} ms")
    print("top (entry, shape):")
    for r in rows[:28]:
        extra = (f"{r['kind']:22s} {str(r.get('shape', r.get('bytes', ''))):24s} {r.get('TFLOPs', 0.0):7.1f} TF/s {r.get('TBps', 0.0):6.3f} TB/s" if "kind" in r
                 else str(r["args"])[:60])
        print(f"  {r['ms_per_step']:7.2f} ms  x{r['calls_per_step']:5.1f}  {r['us_per_call']:8.1f} us  {r['entry'][6:28]:22s} {extra}")


if __name__ == "__main__":
    main()
