"""Loss-curve parity at a REALISTIC per-GPU batch (round-3 review, item 6).  Runs on a GPU box; the oracle (CPU restatement of the
reference, test infrastructure) runs on the host cores and is only the checker.

The committed curve tests (tests/test_hip_modules.py) run at batch 2 on five cycled batches: a 12-token MLM mean and a 2-sample ITM loss,
the regime that maximises rounding noise.  This tool runs the same comparison -- identical weights, identical batches, the reference's
optimizer structure (6 parameter groups, lr x5 on heads / cross-modal, HF AdamW, warm-up + poly decay), dropout / DropPath off -- with
FRESH batches every step at:
    swin_t      Swin-T + RoBERTa-base, 224^2, S = 40, B = 32, 50 steps      (BASELINE.json configs[0] shape)
    fiber_base  FIBER-Base (Swin-B 384^2 + RoBERTa-base), S = 40, B = 8, 20 steps   (the reference's per-GPU batch, cg/README.md:35)
for four runs started from the same point:
    oracle fp32 (the yardstick) | oracle under torch.autocast(bfloat16) (the reference's own mixed precision, config.py:92)
    HIP, bf16 residual stream (the default)  |  HIP, fp32 residual stream (config["residual_dtype"] = "fp32")
and reports, per run, the gap to the fp32 oracle: median / p90 / max and the number of steps within the north star's +-1e-3.

    python tools/loss_curve_study.py swin_t|fiber_base [--B N] [--steps K] [--threads T]      -> gpurun_out/r05_loss_curve_<case>_b<B>.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import cases, detgen
from oracle import fiber_ref as R


def _to_dev(b, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else [t.to(dev) for t in v] if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v)
            for k, v in b.items()}


def hf_groups(model_named, opt_groups, name_of, rparams):
    groups = [{"params": [], "weight_decay": g["weight_decay"], "lr": g["initial_lr"]} for g in opt_groups]
    for gi, g in enumerate(opt_groups):
        for p in g["params"]:
            groups[gi]["params"].append(rparams[name_of[id(p)]])
    return groups


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=["swin_t", "fiber_base"])
    ap.add_argument("--B", type=int, default=0)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--no-autocast", action="store_true")
    ap.add_argument("--only-autocast", action="store_true",
                    help="only the autocast-oracle column (per-step losses appended to gpurun_out/<tag>_autocast.log as they come): "
                         "completes a study whose other columns exist already")
    a = ap.parse_args()
    cfg, size = (dict(cases.SWIN_T), 224) if a.case == "swin_t" else (dict(cases.SWIN_B), 384)
    B = a.B or (32 if a.case == "swin_t" else 8)
    steps = a.steps or (50 if a.case == "swin_t" else 20)
    warm = max(1, steps // 10)
    torch.set_num_threads(max(1, min(a.threads, os.cpu_count() or 1)))
    from fiber_amd import lib, ops, parallel
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils
    from fiber_amd.optim import HFAdamW
    from tests.test_hip_modules import load_from_oracle
    lib.load()
    dev = torch.device("cuda", 0)
    hyper = dict(learning_rate=a.lr, lr_mult_head=5, lr_mult_cross_modal=5, warmup_steps=warm, max_steps=steps, weight_decay=0.01,
                 end_lr=0, decay_power=1)
    batches = [detgen.synth_batch(B, size, 40, 50265, seed=1000 + i, min_len=8) for i in range(steps)]

    def fresh_ref():
        torch.manual_seed(0)
        ref = detgen.fill_(R.FiberRef(cfg).train())
        for n, p in ref.named_parameters():
            if "alpha_" in n:
                p.data.fill_(0.5)
        return ref

    curves, secs = {}, {}
    # ---- HIP runs (each builds its own product model from the same oracle weights; the optimizer's groups define the oracle's)
    group_spec = None
    for mode in ("hip_bf16_stream", "hip_fp32_stream"):
        ref = fresh_ref()
        model = FIBERTransformerSS(make_config(**cfg, **hyper, residual_dtype="fp32" if mode == "hip_fp32_stream" else "bf16")).train()
        load_from_oracle(model, ref)
        for n, p in model.named_parameters():
            if "alpha_" in n:
                p.data.fill_(0.5)
        model.to(dev)
        fiber_utils.set_task(model)
        parallel.freeze_unused(model, model.unused_parameter_names())
        (opt,), (sched,) = model.configure_optimizers()
        if group_spec is None:
            name_of = {id(p): n for n, p in model.named_parameters()}
            group_spec = [{"names": [name_of[id(p)] for p in g["params"]], "weight_decay": g["weight_decay"], "lr": g["initial_lr"]}
                          for g in opt.param_groups]
        t0, ls = time.time(), []
        for step in range(0 if a.only_autocast else steps):
            bd = _to_dev(batches[step], dev)
            bd["itm_labels_override"] = bd["itm_labels"]
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(bd, step)
            loss.backward()
            opt.step()
            sched["scheduler"].step()
            ls.append(loss.item())
        curves[mode], secs[mode] = ls, time.time() - t0
        print(mode, f"{secs[mode]:.0f}s", [round(v, 4) for v in ls], flush=True)
        del model, opt
        if a.only_autocast:
            break
        ops.set_residual_dtype("bf16")
        torch.cuda.empty_cache()
    # ---- oracle runs on the host cores
    for mode in (["oracle_autocast_bf16"] if a.only_autocast else ["oracle_fp32"] + ([] if a.no_autocast else ["oracle_autocast_bf16"])):
        ref = fresh_ref()
        rparams = dict(ref.named_parameters())
        groups = [{"params": [rparams[n] for n in g["names"]], "weight_decay": g["weight_decay"], "lr": g["lr"]} for g in group_spec]
        ropt = HFAdamW(groups, lr=a.lr, eps=1e-8, betas=(0.9, 0.98))
        rsched = torch.optim.lr_scheduler.LambdaLR(ropt, lambda s_: fiber_utils.poly_decay_lambda(s_, warm, steps, a.lr, 0, 1))
        t0, ls = time.time(), []
        for step in range(steps):
            b = batches[step]
            ropt.zero_grad(set_to_none=True)
            if mode == "oracle_fp32":
                rl = ref.training_loss(b, b["itm_labels"])
            else:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    rl = ref.training_loss(b, b["itm_labels"])
                rl = rl.float()
            rl.backward()
            ropt.step()
            rsched.step()
            ls.append(rl.item())
            if a.only_autocast:
                os.makedirs("gpurun_out", exist_ok=True)
                with open(f"gpurun_out/r05_loss_curve_{a.case}_b{B}_autocast.log", "a") as fh:
                    fh.write(f"{step} {rl.item():.6f} {time.time() - t0:.0f}s\n")
            if step == 0:
                print(f"  {mode}: first step {time.time() - t0:.0f}s", flush=True)
        curves[mode], secs[mode] = ls, time.time() - t0
        print(mode, f"{secs[mode]:.0f}s", [round(v, 4) for v in ls], flush=True)
    if a.only_autocast:
        return
    want = curves["oracle_fp32"]
    summary = {}
    for mode, ls in curves.items():
        if mode == "oracle_fp32":
            continue
        gaps = sorted(abs(x - y) for x, y in zip(ls, want))
        summary[mode] = {"gap_median": gaps[len(gaps) // 2], "gap_p90": gaps[int(0.9 * (len(gaps) - 1))], "gap_max": gaps[-1],
                         "steps_within_1e-3": sum(g <= 1e-3 for g in gaps), "step0": abs(ls[0] - want[0])}
    out = {"what": "MLM+ITM loss curves from identical weights on identical FRESH batches; gap of each run to the fp32 oracle",
           "case": a.case, "image_size": size, "B": B, "steps": steps, "warmup": warm, "lr": a.lr, "mlm_tokens_per_step": None,
           "loss_first": want[0], "loss_last": want[-1], "seconds": secs, "summary": summary, "curves": {k: [round(v, 5) for v in c] for k, c in curves.items()}}
    print(json.dumps(summary, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/r05_loss_curve_{a.case}_b{B}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
