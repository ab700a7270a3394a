"""Per-stage time budget of the bench step: ms per step forward / backward and achieved TFLOP/s for

    text prefix (embeddings + layers 0-5) | patch embed | Swin stage 0 | stage 1 | stage 2 plain blocks 0-13 |
    stage 2 fused steps 14-17 (+ text layers 6-9) | merge + stage 3 fused steps (+ text layers 10-11) | heads + losses | AdamW

HIP events at the stage boundaries of the forward pass and, through tensor hooks, at the same boundaries of the backward pass.
Single stream (FIBER_NO_OVERLAP=1: with the text stack on its own stream the segments overlap and cannot be attributed).

    FIBER_NO_OVERLAP=1 python tools/stage_breakdown.py [batch=256] [steps=5]  ->  gpurun_out/stage_breakdown.json + a markdown table
"""
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

# algorithmic forward GFLOP per image-text pair
GF = {"text_prefix": 6 * 0.571, "patch_embed": 0.113, "stage0": 2 * 4.30 + 0.604, "stage1": 2 * 3.96 + 0.604,
      "stage2_plain": 14 * 3.79, "stage2_fused": 4 * (3.79 + 0.714) + 4 * (0.571 + 1.071),
      "stage3_fused": 0.604 + 2 * (3.71 + 0.753) + 2 * (0.571 + 0.565), "heads": 0.27 + 3.14 / 2}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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

    marks = []                                     # (label, phase, event) in execution order

    def mark(label, phase):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label, phase, e))

    def wrap(module, label, pick=lambda o: o):
        orig = module.forward

        def fwd(*a, **k):
            out = orig(*a, **k)
            t = pick(out)
            mark(label, "fwd")
            if t.requires_grad:
                t.register_hook(lambda g, label=label: mark(label, "bwd"))
            return out
        module.forward = fwd

    vit, txt = model.vit_model, model.text_transformer
    npt = model.num_text_layer - model.num_fuse_block
    wrap(txt.encoder.layer[npt - 1], "text_prefix", lambda o: o[0])
    wrap(vit.patch_embed, "patch_embed")
    wrap(vit.layers[0], "stage0")
    wrap(vit.layers[1], "stage1")
    wrap(vit.layers[2].blocks[8 + npt - 1], "stage2_plain")
    wrap(vit.layers[2].blocks[-1], "stage2_fused")
    wrap(vit.layers[3].blocks[-1], "stage3_fused")

    def step(record):
        ops.set_rng_step(model.global_step)
        marks.clear()
        mark("start", "fwd")
        out = model(batch)
        loss = sum(v for k, v in out.items() if "loss" in k)
        mark("heads", "fwd")
        loss.backward()
        mark("input", "bwd")
        opt.step()
        sched["scheduler"].step()
        opt.zero_grad(set_to_none=True)
        mark("adamw", "opt")
        model.global_step += 1
        if not record:
            return None
        torch.cuda.synchronize()
        seg = {}
        for (l0, p0, e0), (l1, p1, e1) in zip(marks, marks[1:]):
            # a forward event closes the segment it is named after; a backward hook on a stage's OUTPUT fires when everything
            # downstream has been enqueued, i.e. it closes the backward of the NEXT stage in forward order
            seg[(l1, p1)] = seg.get((l1, p1), 0.0) + e0.elapsed_time(e1)
        return seg

    for _ in range(3):
        step(False)
    acc = {}
    for _ in range(steps):
        for k, v in step(True).items():
            acc[k] = acc.get(k, 0.0) + v / steps
    order = ["text_prefix", "patch_embed", "stage0", "stage1", "stage2_plain", "stage2_fused", "stage3_fused", "heads"]
    # Backward attribution.  The hook on the OUTPUT of stage s fires when everything downstream of it has been enqueued, i.e. it
    # closes the backward of the stage AFTER s.  The text prefix is created first in the forward, so autograd (highest sequence
    # number first) runs its backward LAST, after the patch embedding's: the two are reported together on the text_prefix row.  The
    # hook on the text prefix's output fires inside the backward of the first fusion step and only splits that segment.
    bwd_of = {"heads": [("stage3_fused", "bwd")], "stage3_fused": [("stage2_fused", "bwd")],
              "stage2_fused": [("stage2_plain", "bwd"), ("text_prefix", "bwd")], "stage2_plain": [("stage1", "bwd")],
              "stage1": [("stage0", "bwd")], "stage0": [("patch_embed", "bwd")], "patch_embed": [], "text_prefix": [("input", "bwd")]}
    rows = []
    imgs = 2 * B                                     # one fused pass over 2B image-text pairs (MLM + ITM)
    for name in order:
        f = acc.get((name, "fwd"), 0.0)
        b = sum(acc.get(k, 0.0) for k in bwd_of[name])
        gf_b = GF[name] + (GF["patch_embed"] if name == "text_prefix" else 0.0)
        rows.append({"segment": name + (" (+ patch-embed backward)" if name == "text_prefix" else ""), "fwd_ms": round(f, 2),
                     "bwd_ms": round(b, 2), "gflop_fwd_per_pair": GF[name], "fwd_TFLOPs": round(GF[name] * imgs / f, 1) if f else None,
                     "bwd_TFLOPs": round(2 * gf_b * imgs / b, 1) if b else None})
    raw = [(l, p, round(acc[(l, p)], 3)) for (l, p) in acc]
    opt_ms = acc.get(("adamw", "opt"), 0.0)
    total = sum(acc.values())
    out = {"batch": B, "steps": steps, "single_stream": True, "step_ms": round(total, 2), "adamw_ms": round(opt_ms, 2), "segments": rows,
           "raw_segments_in_first_seen_order": raw}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/stage_breakdown.json", "w"), indent=1)
    print(f"| segment | fwd ms | bwd ms | share of step | fwd TFLOP/s | bwd TFLOP/s |\n|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['segment']} | {r['fwd_ms']} | {r['bwd_ms']} | {(r['fwd_ms'] + r['bwd_ms']) / total * 100:.1f} % | {r['fwd_TFLOPs']} | {r['bwd_TFLOPs']} |")
    print(f"| AdamW + weight copies | {opt_ms:.2f} | | {opt_ms / total * 100:.1f} % | | |")
    print(f"step (single stream) {total:.1f} ms at B={B}")


if __name__ == "__main__":
    main()
