"""set_metrics / epoch_wrapup / set_task / set_schedule for the metric path (reference
coarse_grained/fiber/modules/fiber_utils.py:14-41, 44-140, 151-287; coarse_grained/fiber/gadgets/my_metrics.py).  AdamW and the polynomial-decay-with-warmup schedule restate
transformers==4.6.0 (`AdamW(correct_bias=True)`: decoupled weight decay, eps 1e-8, betas (0.9, 0.98);
`get_polynomial_decay_schedule_with_warmup`)."""
import math
import os

import torch


def _world_sum(t):
    """dist_reduce_fx="sum" of the reference's Metric states (my_metrics.py:8-9,34-35,52-53): metric state is summed over
    the data-parallel ranks when an epoch value is computed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class _Running:
    """Stand-in for pytorch_lightning.metrics.Metric as the reference uses it (my_metrics.py): calling the metric updates
    the epoch state AND returns the value of this batch (Lightning 1.3 `Metric.forward`); compute() = state over the epoch
    (summed over ranks), reset() clears it.  State lives on the device of the inputs as two fp32 scalars that are only
    added to -- nothing here synchronises the host on the hot path."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.num, self.den, self.last = None, None, None

    def _update(self, num, den):
        num, den = num.detach().float(), den.detach().float()
        if self.num is None:
            self.num, self.den = num.clone(), den.clone()
        else:
            self.num += num
            self.den += den

    def compute(self):
        if self.num is None:
            return torch.tensor(float("nan"))
        state = _world_sum(torch.stack([self.num, self.den]))
        return state[0] / state[1]


class Scalar(_Running):
    """my_metrics.py:31-46: mean of the logged scalars."""

    def __call__(self, scalar):
        v = scalar.detach().float() if isinstance(scalar, torch.Tensor) else torch.tensor(float(scalar))
        self._update(v, torch.ones((), device=v.device))
        self.last = v
        return v


class Accuracy(_Running):
    """my_metrics.py:5-28: arg-max accuracy over the targets that are not -100."""

    def __call__(self, logits, target):
        with torch.no_grad():
            tgt = target.detach().to(logits.device)
            valid = tgt != -100
            handed = getattr(logits, "_fiber_argmax", None)     # ops.cross_entropy left the arg max of its labelled rows on the logits it was given
            if (handed is not None and handed[2] == -100 and handed[0].numel() == tgt.numel() and handed[1].data_ptr() == tgt.data_ptr()):
                preds = handed[0].view(tgt.shape).long()        # (-1 on ignored rows: never equal to a target the metric counts)
            else:
                preds = logits.detach().argmax(dim=-1)
            correct = ((preds == tgt) & valid).sum().float()
            total = valid.sum().float()
            self._update(correct, total)
            self.last = correct / total.clamp(min=1)
        return self.last


class VQAScore(_Running):
    """my_metrics.py:49-69: soft VQA accuracy = score of the arg-max answer, averaged over questions."""

    def __call__(self, logits, target):
        with torch.no_grad():
            pick = logits.detach().float().argmax(dim=1, keepdim=True)
            score = target.detach().float().gather(1, pick).sum()
            n = torch.tensor(float(logits.shape[0]), device=score.device)
            self._update(score, n)
            self.last = score / n
        return self.last


def set_metrics(pl_module):
    for split in ["train", "val"]:
        for k, v in pl_module.hparams.config["loss_names"].items():
            if v <= 0:
                continue
            if k == "vqa":                     # fiber_utils.py:19-21
                setattr(pl_module, f"{split}_vqa_score", VQAScore())
                setattr(pl_module, f"{split}_{k}_loss", Scalar())
                continue
            if k == "itc":                     # fiber_utils.py:36-40
                setattr(pl_module, f"{split}_{k}_i2t_accuracy", Accuracy())
                setattr(pl_module, f"{split}_{k}_t2i_accuracy", Accuracy())
                setattr(pl_module, f"{split}_{k}_loss", Scalar())
                setattr(pl_module, f"{split}_{k}_logit_scale", Scalar())
                continue
            setattr(pl_module, f"{split}_{k}_accuracy", Accuracy())
            setattr(pl_module, f"{split}_{k}_loss", Scalar())


# epoch-level metric(s) of a task: (attribute suffix, logged tag); the LAST one is what enters `the_metric`
_EPOCH_VALUES = {"vqa": (("score", "score"),), "itc": (("i2t_accuracy", "i2t_accuracy"), ("t2i_accuracy", "t2i_accuracy"))}


def epoch_wrapup(pl_module):
    """End-of-epoch bookkeeping (reference fiber_utils.py:44-140): per active task log `<task>/<phase>/<metric>_epoch` and
    `<task>/<phase>/loss_epoch`, reset the running metrics, and log their sum as `<phase>/the_metric` -- the quantity
    run.py:29-35's ModelCheckpoint monitors.  Retrieval recall (compute_itc_recall / compute_itm_recall) is outside the
    fused-backbone path built here."""
    cfg = pl_module.hparams.config
    phase = "train" if pl_module.training else "val"
    if cfg.get("get_recall_metric") and not pl_module.training:
        raise NotImplementedError("retrieval-recall evaluation (objectives.py:266-499) is outside the hot path built here")
    the_metric = 0
    for task, weight in cfg["loss_names"].items():
        if weight <= 0:
            continue
        value = 0
        for attr, tag in _EPOCH_VALUES.get(task, (("accuracy", "accuracy"),)):
            metric = getattr(pl_module, f"{phase}_{task}_{attr}", None)
            if metric is None:
                continue
            value = metric.compute()
            pl_module.log(f"{task}/{phase}/{tag}_epoch", value)
            metric.reset()
        running_loss = getattr(pl_module, f"{phase}_{task}_loss")
        pl_module.log(f"{task}/{phase}/loss_epoch", running_loss.compute())
        running_loss.reset()
        the_metric = the_metric + value
    pl_module.log(f"{phase}/the_metric", the_metric)


def set_task(pl_module):
    pl_module.current_tasks = [k for k, v in pl_module.hparams.config["loss_names"].items() if v > 0]


NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight"]
HEAD_NAMES = ["vqa_classifier", "nlvr2_classifier", "mlm_score", "itm_score", "snli_classifier"]
CROSS_MODAL_NAMES = ["cross_modal", "i2t", "t2i"]


def param_group_index(name):
    """0..5: (plain|head|cross_modal) x (decay|no_decay) in the reference's order (fiber_utils.py:179-245)."""
    nd = any(s in name for s in NO_DECAY)
    head = any(s in name for s in HEAD_NAMES)
    cross = any(s in name for s in CROSS_MODAL_NAMES)
    if head and cross:
        return None                           # matches no reference group -> not optimised (as in the reference)
    return (4 if cross else (2 if head else 0)) + (1 if nd else 0)


def poly_decay_lambda(step, warmup, total, lr_init, lr_end, power):
    if step < warmup:
        return float(step) / float(max(1, warmup))
    if step > total:
        return lr_end / lr_init
    rem = 1 - (step - warmup) / (total - warmup)
    return ((lr_init - lr_end) * rem ** power + lr_end) / lr_init


def set_schedule(pl_module):
    cfg = pl_module.hparams.config
    lr, wd = cfg["learning_rate"], cfg["weight_decay"]
    mult = [1, 1, cfg["lr_mult_head"], cfg["lr_mult_head"], cfg["lr_mult_cross_modal"], cfg["lr_mult_cross_modal"]]
    groups = [{"params": [], "weight_decay": wd if i % 2 == 0 else 0.0, "lr": lr * mult[i]} for i in range(6)]
    for n, p in pl_module.named_parameters():
        gi = param_group_index(n)
        if gi is not None and p.requires_grad:
            groups[gi]["params"].append(p)
    own = None
    if cfg["optim_type"] == "adamw":
        # transformers 4.6.0 AdamW(correct_bias=True): eps on the un-corrected sqrt(v), weight decay after the update -- NOT
        # torch.optim.AdamW's form of the rule (fiber_amd/optim.py).  On a HIP device: one kernel per parameter group that
        # also rewrites the bf16 working copies; host tensors (the CPU wiring tests) take the plain-torch statement of the
        # same rule.
        from ..optim import FiberAdamW, HFAdamW
        if any(p.is_cuda for g in groups for p in g["params"]) and not os.environ.get("FIBER_TORCH_ADAMW"):
            optimizer = own = FiberAdamW(groups, lr=lr, eps=1e-8, betas=(0.9, 0.98))
        else:
            optimizer = HFAdamW(groups, lr=lr, eps=1e-8, betas=(0.9, 0.98))
    elif cfg["optim_type"] == "adam":
        optimizer = torch.optim.Adam(groups, lr=lr)
    else:
        optimizer = torch.optim.SGD(groups, lr=lr, momentum=0.9)
    from .. import ops
    if own is None:
        optimizer.register_step_post_hook(lambda *a, **k: ops.mark_weights_dirty())   # bf16 working copies follow the update
    tr = getattr(pl_module, "trainer", None)
    max_steps = getattr(tr, "max_steps", None) if tr is not None else None
    if max_steps is None and tr is not None and getattr(tr, "datamodule", None) is not None and getattr(tr, "max_epochs", None):
        # epoch-bounded runs (fine-tuning configs set max_steps=None): fiber_utils.py:254-259
        max_steps = (len(tr.datamodule.train_dataloader()) * tr.max_epochs // getattr(tr, "accumulate_grad_batches", 1))
    if max_steps is None:
        max_steps = cfg["max_steps"]
    warmup = cfg["warmup_steps"]
    if isinstance(warmup, float):
        warmup = int(max_steps * warmup)
    if cfg["decay_power"] == "cosine":
        lam = lambda s: s / max(1, warmup) if s < warmup else max(0.0, 0.5 * (1 + math.cos(math.pi * (s - warmup) / max(1, max_steps - warmup))))
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lam)
    else:
        # HF uses optimizer.defaults["lr"] (the base lr) as lr_init for every group
        lam = lambda s: poly_decay_lambda(s, warmup, max_steps, lr, cfg["end_lr"], cfg["decay_power"])
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lam)
    return [optimizer], [{"scheduler": scheduler, "interval": "step"}]
