"""TEST INFRASTRUCTURE (never imported by the product path): the fp32 oracle's MLM+ITM loss curve on a fixed cycle of synthetic
batches, with the reference's optimisation recipe -- 6 parameter groups by name (coarse_grained/fiber/modules/fiber_utils.py:160-245),
transformers-4.6.0 AdamW (fiber_utils.py:248), linear warm-up + polynomial decay (fiber_utils.py:274), dropout / DropPath 0 --
as one function that the GPU curve tests call live, or replace by the committed fixture tests/golden/curve_<tag>.json
(oracle/gen_curve_golden.py writes those with this very function; the oracle itself is pinned by tests/test_oracle_golden.py).

The curve is a deterministic function of (cfg, size, batch, steps, nb, warm, seeds): the weights come from oracle/detgen.py's
counter-based generator, the batches from detgen.synth_batch(seed = 100 + i)."""
import os

import torch

from . import detgen, fiber_ref as R

HYPER = dict(learning_rate=2e-5, lr_mult_head=5, lr_mult_cross_modal=5, weight_decay=0.01, end_lr=0, decay_power=1)

# tag -> (cases name, image size, batch, steps, number of cycled batches, warm-up steps): the curves of tests/test_hip_modules.py
CURVES = {
    "swin_t": ("SWIN_T", 224, 2, 50, 5, 5),
    "fiber_base": ("SWIN_B", 384, 2, 8, 2, 2),
    "swin_t_b32": ("SWIN_T", 224, 32, 6, 6, 1),
}


def spec_of(tag):
    name, size, batch, steps, nb, warm = CURVES[tag]
    return {"cases": name, "size": size, "batch": batch, "steps": steps, "nb": nb, "warm": warm, "hyper": HYPER, "alpha": 0.5,
            "batch_seed0": 100, "max_text_len": 40, "vocab": 50265, "min_len": 8}


def build_ref(cfg):
    """The oracle with detgen's weights and the gates opened (reference init 0 would switch the fusion branches off)."""
    torch.manual_seed(0)
    ref = detgen.fill_(R.FiberRef(cfg).train())
    for n, p in ref.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)
    return ref


def batches_of(size, batch, nb):
    return [detgen.synth_batch(batch, size, 40, 50265, seed=100 + i, min_len=8) for i in range(nb)]


def oracle_curve(cfg, size, batch, steps, nb, warm, ref=None, frozen=()):
    """-> list of `steps` fp32 losses.  `frozen` = parameter names outside the optimizer (the product freezes the parameters the
    fused path never touches; they get no gradient here either, so listing them changes nothing but the bookkeeping)."""
    from fiber_amd.modules import fiber_utils           # name -> group rule and the schedule: host-side Python, no kernels
    from fiber_amd.optim import HFAdamW                  # plain-torch statement of transformers 4.6.0 AdamW
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))    # (more threads make the many small ops slower: r06_summary.md section 6)
    ref = build_ref(cfg) if ref is None else ref
    lr, wd = HYPER["learning_rate"], HYPER["weight_decay"]
    mult = [1, 1, HYPER["lr_mult_head"], HYPER["lr_mult_head"], HYPER["lr_mult_cross_modal"], HYPER["lr_mult_cross_modal"]]
    groups = [{"params": [], "weight_decay": wd if i % 2 == 0 else 0.0, "lr": lr * mult[i]} for i in range(6)]
    frozen = set(frozen)
    for n, p in ref.named_parameters():
        gi = fiber_utils.param_group_index(n)
        if gi is not None and n not in frozen and not n.startswith("rank_output."):
            groups[gi]["params"].append(p)
    opt = HFAdamW(groups, lr=lr, eps=1e-8, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s_: fiber_utils.poly_decay_lambda(s_, warm, steps, lr, 0, 1))
    batches = batches_of(size, batch, nb)
    out = []
    for step in range(steps):
        b = batches[step % nb]
        opt.zero_grad(set_to_none=True)
        loss = ref.training_loss(b, b["itm_labels"])
        loss.backward()
        opt.step()
        sched.step()
        out.append(loss.item())
    return out


# The small fixed-batch curve of tests/test_hip_modules.py::test_loss_curve_tracks_oracle_over_optimizer_steps: cases.TINY, ONE batch
# (synth_batch(4, 96, 12, 1000, seed=11, min_len=6)) for 8 steps at the constant group rates of lr = 1e-4 (no warm-up, no decay step taken),
# gates at detgen's values.
TINY_SPEC = {"cases": "TINY", "batch": [4, 96, 12, 1000, 11, 6], "steps": 8, "lr": 1e-4, "lr_mult_head": 5, "lr_mult_cross_modal": 5,
             "weight_decay": 0.01}


def oracle_curve_tiny(ref=None, frozen=()):
    from fiber_amd.modules import fiber_utils
    from fiber_amd.optim import HFAdamW
    from . import cases
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))    # (more threads make the many small ops slower: r06_summary.md section 6)
    if ref is None:
        ref = detgen.fill_(R.FiberRef(dict(cases.TINY)).train())
    lr, wd = TINY_SPEC["lr"], TINY_SPEC["weight_decay"]
    mult = [1, 1, TINY_SPEC["lr_mult_head"], TINY_SPEC["lr_mult_head"], TINY_SPEC["lr_mult_cross_modal"], TINY_SPEC["lr_mult_cross_modal"]]
    groups = [{"params": [], "weight_decay": wd if i % 2 == 0 else 0.0, "lr": lr * mult[i]} for i in range(6)]
    frozen = set(frozen)
    for n, p in ref.named_parameters():
        gi = fiber_utils.param_group_index(n)
        if gi is not None and n not in frozen and not n.startswith("rank_output."):
            groups[gi]["params"].append(p)
    opt = HFAdamW(groups, lr=lr, eps=1e-8, betas=(0.9, 0.98))
    b = detgen.synth_batch(*TINY_SPEC["batch"][:4], seed=TINY_SPEC["batch"][4], min_len=TINY_SPEC["batch"][5])
    out = []
    for _ in range(TINY_SPEC["steps"]):
        opt.zero_grad(set_to_none=True)
        loss = ref.training_loss(b, b["itm_labels"])
        loss.backward()
        opt.step()
        out.append(loss.item())
    return out

