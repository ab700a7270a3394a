"""CPU: the Lightning-free Trainer stand-in (gradient accumulation, scheduler stepping, max_steps) on a toy module, and
the optimizer parameter-group rules / LR schedule of fiber_utils.set_schedule against the reference's rules."""
import torch
import torch.nn as nn

from fiber_amd.lightning import LightningModule
from fiber_amd.modules import fiber_utils
from fiber_amd.trainer import Trainer


class Toy(LightningModule):
    def __init__(self, config):
        super().__init__()
        self.save_hyperparameters()
        self.lin = nn.Linear(4, 1)
        self.current_tasks = ["toy"]

    def forward(self, batch):
        return {"toy_loss": (self.lin(batch["x"]) - batch["y"]).pow(2).mean()}

    def configure_optimizers(self):
        return fiber_utils.set_schedule(self)


def _cfg(**kw):
    c = dict(learning_rate=0.1, weight_decay=0.0, lr_mult_head=5, lr_mult_cross_modal=5, end_lr=0, decay_power=1,
             optim_type="sgd", warmup_steps=2, max_steps=10)
    c.update(kw)
    return c


def test_trainer_accumulates_and_steps():
    torch.manual_seed(0)
    m = Toy(_cfg())
    assert m.hparams.config["learning_rate"] == 0.1
    w0 = m.lin.weight.detach().clone()
    data = [{"x": torch.randn(8, 4), "y": torch.randn(8, 1)} for _ in range(4)]
    tr = Trainer(max_steps=3, accumulate_grad_batches=2, log_every_n_steps=0)
    tr.fit(m, data, device=torch.device("cpu"))
    assert tr.global_step == 3 and m.global_step == 3
    assert not torch.equal(w0, m.lin.weight.detach())


def test_poly_schedule_matches_hf_formula():
    # transformers 4.6.0 get_polynomial_decay_schedule_with_warmup(lr_end=0, power=1): linear warm-up then linear decay
    f = lambda s: fiber_utils.poly_decay_lambda(s, 10, 100, 1e-5, 0.0, 1)
    assert f(0) == 0.0 and abs(f(5) - 0.5) < 1e-12 and abs(f(10) - 1.0) < 1e-12
    assert abs(f(55) - 0.5) < 1e-12 and f(100) == 0.0 and f(150) == 0.0


def test_param_group_rules():
    """fiber_utils.py:160-245: (plain|head|cross_modal) x (decay|no_decay); names containing cross_modal / i2t / t2i get
    lr x lr_mult_cross_modal, heads lr x lr_mult_head, the no_decay list gets weight_decay 0."""
    g = fiber_utils.param_group_index
    assert g("vit_model.layers.0.blocks.0.attn.qkv.weight") == 0
    assert g("vit_model.layers.0.blocks.0.attn.qkv.bias") == 1
    assert g("vit_model.layers.0.blocks.0.norm1.weight") == 1
    assert g("text_transformer.encoder.layer.0.attention.output.LayerNorm.weight") == 1
    assert g("mlm_score.decoder.weight") == 2 and g("itm_score.fc.bias") == 3
    assert g("vit_model.layers.2.blocks.15.attn.qkv_i2t.weight") == 4
    assert g("vit_model.layers.2.blocks.15.attn.alpha_i2t") == 4
    assert g("text_transformer.encoder.layer.7.crossattention_t2i.self.key.bias") == 5
    assert g("cross_modal_text_transform.weight") == 4 and g("cross_modal_image_pooler.dense.bias") == 5
    # norm_i2t_i.weight: "norm.weight"/"norm1.weight" are not substrings -> decays, cross-modal lr (as in the reference)
    assert g("vit_model.layers.3.blocks.0.attn.norm_i2t_i.weight") == 4
    # "bias" is a substring of relative_position_bias_table -> the reference puts it in the no-decay group
    assert g("vit_model.layers.0.blocks.0.attn.relative_position_bias_table") == 1


def test_set_schedule_matches_reference_groups_and_lr(golden):
    """G8: optimizer group membership, group lr / weight decay and the per-step learning rates of
    fiber_utils.set_schedule against the reference's own set_schedule (run by oracle/gen_golden.py) -- for the step-bounded
    pre-training config and the epoch-bounded fine-tuning config (max_steps=None -> dataloader length x epochs // accumulation)."""
    import types
    import numpy as np
    import torch
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils
    from oracle import cases
    gold = golden("schedule")
    for tag, over, trainer in (
            ("pretrain", dict(loss_names={"mlm": 1, "itm": 1}, learning_rate=1e-5, lr_mult_head=5, lr_mult_cross_modal=5,
                              warmup_steps=0.1), dict(max_steps=200, max_epochs=None, accumulate_grad_batches=1, n_batches=0)),
            ("vqa", dict(loss_names={"vqa": 1}, learning_rate=2e-5, lr_mult_head=50, lr_mult_cross_modal=5, warmup_steps=0.1,
                         vqav2_label_size=17), dict(max_steps=None, max_epochs=10, accumulate_grad_batches=2, n_batches=37))):
        cfg = make_config(**dict(cases.TINY, weight_decay=0.01, end_lr=0, decay_power=1, optim_type="adamw", **over))
        m = FIBERTransformerSS(cfg)
        assert sorted(n for n, _ in m.named_parameters()) == gold[f"{tag}/all_names"].tolist()
        dl = list(range(trainer["n_batches"]))
        m.trainer = types.SimpleNamespace(max_steps=trainer["max_steps"], max_epochs=trainer["max_epochs"],
                                          accumulate_grad_batches=trainer["accumulate_grad_batches"],
                                          datamodule=types.SimpleNamespace(train_dataloader=lambda dl=dl: dl))
        (opt,), (sch,) = fiber_utils.set_schedule(m)
        names = {id(p): n for n, p in m.named_parameters()}
        assert len(opt.param_groups) == 6
        for gi, g in enumerate(opt.param_groups):
            assert sorted(names[id(p)] for p in g["params"]) == gold[f"{tag}/group{gi}/names"].tolist(), (tag, gi)
            assert abs(g["initial_lr"] - float(gold[f"{tag}/group{gi}/lr"])) < 1e-12
            assert g["weight_decay"] == float(gold[f"{tag}/group{gi}/wd"])
        want = gold[f"{tag}/lrs"]
        for step in range(want.shape[0]):
            got = np.array([g["lr"] for g in opt.param_groups])
            np.testing.assert_allclose(got, want[step], rtol=1e-9, atol=1e-15, err_msg=f"{tag} step {step}")
            opt.step()
            sch["scheduler"].step()


# ---- boundary completeness: import surface, epoch hooks, the_metric, Trainer epochs / checkpoints ----------

def test_fiber_import_surface_alias():
    """run.py:7-9 does `from fiber.modules import FIBERTransformerSS`: the alias package serves the MI355X-native class and the
    sub-module names callers use."""
    import importlib
    import fiber_amd.modules as native
    from fiber.modules import FIBERTransformerSS
    assert FIBERTransformerSS is native.FIBERTransformerSS
    for sub in ("fiber_module", "objectives", "fiber_utils", "heads", "swin_transformer", "roberta", "swin_helpers"):
        assert importlib.import_module(f"fiber.modules.{sub}") is getattr(native, sub)


def _tiny_module(loss_names):
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS
    from oracle import cases
    return FIBERTransformerSS(make_config(**dict(cases.TINY, loss_names=loss_names, vqav2_label_size=17)))


def test_epoch_hooks_log_the_metric():
    """fiber_module.py:480-489 / fiber_utils.py:44-140: *_epoch_end -> epoch_wrapup logs `<task>/<phase>/accuracy_epoch`,
    `.../loss_epoch` and their sum `<phase>/the_metric` (what run.py:29-35's ModelCheckpoint monitors), from running metrics
    with the reference's semantics (my_metrics.py: correct / total over the epoch, ignoring -100; mean of the losses)."""
    m = _tiny_module({"mlm": 1, "itm": 1})
    for hook in ("training_epoch_end", "validation_step", "validation_epoch_end", "test_step", "test_epoch_end"):
        assert callable(getattr(m, hook))
    m.train()
    logits = torch.tensor([[[2.0, 0.0, 0.0], [0.0, 3.0, 0.0]], [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]]])     # argmax 0,1 / 2,0
    assert abs(float(m.train_mlm_accuracy(logits, torch.tensor([[0, 2], [-100, 0]]))) - 2 / 3) < 1e-6   # batch value (Metric.forward)
    m.train_mlm_accuracy(logits, torch.tensor([[-100, -100], [2, 1]]))                         # 1 of 2
    m.train_mlm_loss(torch.tensor(2.0)); m.train_mlm_loss(torch.tensor(4.0))
    m.train_itm_accuracy(torch.tensor([[0.0, 1.0], [1.0, 0.0]]), torch.tensor([1, 1]))         # 1 of 2
    m.train_itm_loss(torch.tensor(0.5))
    m.training_epoch_end([])
    assert abs(float(m.logged["mlm/train/accuracy_epoch"]) - 3 / 5) < 1e-6
    assert abs(float(m.logged["mlm/train/loss_epoch"]) - 3.0) < 1e-6
    assert abs(float(m.logged["itm/train/accuracy_epoch"]) - 0.5) < 1e-6
    assert abs(float(m.logged["train/the_metric"]) - (3 / 5 + 0.5)) < 1e-6
    assert m.train_mlm_accuracy.num is None                                                     # reset for the next epoch
    m.eval()
    m.val_mlm_accuracy(logits, torch.tensor([[0, 1], [2, 0]])); m.val_mlm_loss(torch.tensor(1.0))
    m.val_itm_accuracy(torch.tensor([[0.0, 1.0]]), torch.tensor([1])); m.val_itm_loss(torch.tensor(1.0))
    m.validation_epoch_end([])
    assert abs(float(m.logged["val/the_metric"]) - 2.0) < 1e-6


def test_vqa_epoch_metric_and_test_wrapup(tmp_path, monkeypatch):
    import json
    import types
    from fiber_amd.modules import objectives
    m = _tiny_module({"vqa": 1})
    m.eval()
    tgt = torch.zeros(2, 17); tgt[0, 3] = 0.9; tgt[1, 5] = 0.3
    lg = torch.zeros(2, 17); lg[0, 3] = 5.0; lg[1, 4] = 5.0                                     # second question answered wrongly
    m.val_vqa_score(lg, tgt); m.val_vqa_loss(torch.tensor(7.0))
    m.validation_epoch_end([])
    assert abs(float(m.logged["vqa/val/score_epoch"]) - 0.45) < 1e-6 and abs(float(m.logged["val/the_metric"]) - 0.45) < 1e-6
    # test_step record + merged submission file (objectives.py:513-557)
    dm = types.SimpleNamespace(dm_dicts={"vqa": types.SimpleNamespace(id2answer={i: f"ans{i}" for i in range(17)})})
    m.trainer = types.SimpleNamespace(datamodule=dm)
    rec = objectives.vqa_test_step(m, {"qid": [11, 12]}, {"vqa_logits": lg})
    assert rec == {"qids": [11, 12], "preds": ["ans3", "ans4"]}
    monkeypatch.chdir(tmp_path)
    objectives.vqa_test_wrapup([rec], "fiber_x")
    assert json.load(open(tmp_path / "result" / "vqa_submit_fiber_x.json")) == [
        {"question_id": 11, "answer": "ans3"}, {"question_id": 12, "answer": "ans4"}]
    assert not list(tmp_path.glob("vqa_submit_*.json"))


class ToyVal(Toy):
    def training_epoch_end(self, outs):
        self.epochs_ended = getattr(self, "epochs_ended", 0) + 1

    def validation_step(self, batch, batch_idx):
        return self(batch)["toy_loss"]

    def validation_epoch_end(self, outs):
        self.log("val/the_metric", -torch.stack(outs).mean())


def test_trainer_epochs_validation_checkpoint_resume(tmp_path):
    """max_steps=None (the epoch-bounded fine-tuning configs, config.py:134-150) runs max_epochs epochs; every epoch ends with
    training_epoch_end + a validation pass; last.ckpt / best.ckpt carry optimizer and scheduler state and
    resume_from_checkpoint continues the step count and the LR schedule; an empty loader is an error, not a spin."""
    import pytest
    torch.manual_seed(0)
    data = [{"x": torch.randn(8, 4), "y": torch.randn(8, 1)} for _ in range(4)]
    m = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    tr = Trainer(max_steps=None, max_epochs=2, accumulate_grad_batches=1, log_every_n_steps=0, default_root_dir=str(tmp_path), seed=3)
    tr.fit(m, data, val_dataloader=data[:2], device=torch.device("cpu"))
    assert tr.global_step == 8 and tr.current_epoch == 2 and m.epochs_ended == 2 and tr.best_metric is not None
    ck = torch.load(tmp_path / "last.ckpt", weights_only=False)
    assert ck["global_step"] == 8 and ck["optimizer_states"][0]["state"] and ck["lr_schedulers"][0]["last_epoch"] == 8
    assert (tmp_path / "best.ckpt").exists()
    m2 = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    tr2 = Trainer(max_steps=10, log_every_n_steps=0, resume_from_checkpoint=str(tmp_path / "last.ckpt"))
    tr2.fit(m2, data, device=torch.device("cpu"))
    assert tr2.global_step == 10 and m2.global_step == 10
    with pytest.raises(ValueError):
        Trainer(max_steps=2, log_every_n_steps=0).fit(ToyVal(_cfg()), [], device=torch.device("cpu"))


def test_hf_adamw_rule():
    """fiber_amd.optim.HFAdamW = transformers 4.6.0 AdamW(correct_bias=True): eps on the UN-corrected sqrt(v), weight decay
    after the update -- checked against the formula in fp64, and shown to differ from torch.optim.AdamW at step 1."""
    from fiber_amd.optim import HFAdamW
    p0, g = torch.tensor([0.5, -1.0, 2.0], dtype=torch.float64), torch.tensor([1e-9, 0.3, -2.0], dtype=torch.float64)
    p = p0.clone().requires_grad_(True)
    opt = HFAdamW([p], lr=0.1, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    m = v = torch.zeros(3, dtype=torch.float64)
    want = p0.clone()
    for t in range(1, 4):
        p.grad = g.clone()
        opt.step()
        m = 0.9 * m + 0.1 * g
        v = 0.98 * v + 0.02 * g * g
        want = want - 0.1 * (1 - 0.98 ** t) ** 0.5 / (1 - 0.9 ** t) * m / (v.sqrt() + 1e-8)
        want = want - 0.1 * 0.01 * want
        assert torch.allclose(p.detach(), want, rtol=1e-12, atol=0)
    q = p0.clone().requires_grad_(True)
    q.grad = g.clone()
    torch.optim.AdamW([q], lr=0.1, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01).step()
    r = p0.clone().requires_grad_(True)
    r.grad = g.clone()
    HFAdamW([r], lr=0.1, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01).step()
    assert abs(float(q[0] - r[0])) > 1e-3          # tiny gradient: the two forms of eps differ by 1/sqrt(1 - b2)


def test_rng_stream_seeding_and_weak_weight_cache():
    """the dropout / DropPath keys follow config seed (+rank) and the optimizer step; the derived-weight cache
    holds parameters weakly (a deleted module's copies disappear)."""
    import gc
    from fiber_amd import ops
    from fiber_amd.lightning import seed_everything
    seed_everything(7); ops.set_rng_step(5); a = [ops.next_seed() for _ in range(3)]
    seed_everything(7); ops.set_rng_step(5); assert a == [ops.next_seed() for _ in range(3)]
    ops.set_rng_step(6); b = ops.next_seed()
    seed_everything(8); ops.set_rng_step(5); c = ops.next_seed()
    assert b not in a and c not in a and len(set(a)) == 3
    ops.set_rng_step(5 + 4096); assert ops.next_seed() != c                       # the window index wraps, the key does not
    w = torch.nn.Parameter(torch.randn(4, 4))
    n0 = len(ops._wcache)
    ops._cache_put(("T", id(w)), ops._stamp(w), torch.zeros(1), w)
    assert len(ops._wcache) == n0 + 1 and ops._cache_get(("T", id(w)), w) is not None
    del w
    gc.collect()
    assert len(ops._wcache) == n0


def test_val_check_interval_int_and_fraction(tmp_path):
    """val_check_interval is honoured the way pl.Trainer reads it (run.py passes config["val_check_interval"]):
    an int = every N training batches, a float < 1 = that fraction of the epoch; 1.0 = once per epoch; bad values raise."""
    import pytest
    torch.manual_seed(0)
    data = [{"x": torch.randn(8, 4), "y": torch.randn(8, 1)} for _ in range(6)]

    class Counting(ToyVal):
        def validation_epoch_end(self, outs):
            self.val_runs = getattr(self, "val_runs", 0) + 1
            super().validation_epoch_end(outs)

    def runs(vci):
        m = Counting(_cfg(optim_type="adamw", max_steps=12))
        Trainer(max_steps=None, max_epochs=2, log_every_n_steps=0, val_check_interval=vci).fit(
            m, data, val_dataloader=data[:2], device=torch.device("cpu"))
        return m.val_runs

    assert runs(1.0) == 2                      # once per epoch
    assert runs(2) == 6                        # after batches 2, 4, 6 of each epoch
    assert runs(0.5) == 4                      # after batches 3 and 6
    assert runs(4) == 2                        # after batch 4 of each epoch; pl.Trainer runs no extra epoch-end check for an interval
    for bad in (0.0, 1.5, 0):
        with pytest.raises(ValueError):
            Trainer(val_check_interval=bad)


def test_collate_seed_advances_in_graph_rng_mode():
    """the MLM-masking key of data.device_collate must not be the dropout stream's by-value counter (which restarts
    every step and, in graph-RNG mode, IS the whole by-value key): consecutive batches get different keys in either mode, and
    the keys are reproducible from the seed."""
    from fiber_amd import ops
    ops.manual_seed(11)
    a = [ops.collate_seed() for _ in range(3)]
    ops._seed_state["base_dev"] = torch.zeros((), dtype=torch.int64)          # what enable_graph_rng() installs (host tensor here)
    try:
        def new_step(i):                                                      # set_rng_step() minus the device write
            ops._seed_state["step"], ops._seed_state["ctr"] = i, 0
        new_step(0); k0 = ops.collate_seed()
        new_step(1); k1 = ops.collate_seed()
        new_step(1); k2 = ops.collate_seed()
    finally:
        ops._seed_state["base_dev"] = None
    assert len({*a, k0, k1, k2}) == 6
    ops.manual_seed(11)
    assert a == [ops.collate_seed() for _ in range(3)]
    ops.manual_seed(12)
    assert ops.collate_seed() not in a


def test_resume_from_a_lightning_shaped_checkpoint(tmp_path):
    """a reference (PyTorch-Lightning 1.3) `.ckpt` pickles `hyper_parameters` / `callbacks` through classes that
    `weights_only=True` rejects (and that are not installed here).  `load_checkpoint` must still resume from it: those objects
    become inert placeholders, only state_dict / optimizer_states / lr_schedulers / global_step / epoch are kept, and nothing
    from the file is imported or executed."""
    import sys
    import types
    import pytest
    from fiber_amd.trainer import load_checkpoint
    torch.manual_seed(0)
    data = [{"x": torch.randn(8, 4), "y": torch.randn(8, 1)} for _ in range(4)]
    m = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    tr = Trainer(max_steps=4, log_every_n_steps=0, default_root_dir=str(tmp_path))
    tr.fit(m, data, device=torch.device("cpu"))
    own = torch.load(tmp_path / "last.ckpt", weights_only=True)
    # a module that exists only while the file is written, like pytorch_lightning on the machine that trained the reference
    fake = types.ModuleType("pl_absent_here")

    class AttributeDict(dict):
        pass

    class ModelCheckpoint:
        executed = False

        def __reduce__(self):                                  # a callable the loader must never call
            return (_boom, ())

    AttributeDict.__module__ = ModelCheckpoint.__module__ = "pl_absent_here"
    AttributeDict.__qualname__, ModelCheckpoint.__qualname__ = "AttributeDict", "ModelCheckpoint"
    fake.AttributeDict, fake.ModelCheckpoint, fake._boom = AttributeDict, ModelCheckpoint, _boom
    _boom.__module__ = "pl_absent_here"
    sys.modules["pl_absent_here"] = fake
    try:
        ck = dict(own, hyper_parameters=AttributeDict(config={"vit": "swin_base"}), callbacks={ModelCheckpoint: {"best": 1.0}},
                  extra=ModelCheckpoint(), **{"pytorch-lightning_version": "1.3.2"})
        ck.pop("collate_ctr", None)
        torch.save(ck, tmp_path / "pl.ckpt")
    finally:
        del sys.modules["pl_absent_here"]
    with pytest.raises(Exception):
        torch.load(tmp_path / "pl.ckpt", weights_only=True)
    got = load_checkpoint(str(tmp_path / "pl.ckpt"))
    assert not _boom.called
    assert set(got) == {"state_dict", "optimizer_states", "lr_schedulers", "global_step", "epoch", "best_metric"}
    assert got["global_step"] == 4 and all(torch.equal(got["state_dict"][k], v) for k, v in own["state_dict"].items())
    m2 = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    tr2 = Trainer(max_steps=6, log_every_n_steps=0, resume_from_checkpoint=str(tmp_path / "pl.ckpt"))
    tr2.fit(m2, data, device=torch.device("cpu"))
    assert tr2.global_step == 6
    with pytest.raises(ValueError):                             # not a training checkpoint
        torch.save({"weights": torch.zeros(2)}, tmp_path / "w.pt")
        load_checkpoint(str(tmp_path / "w.pt"))


def test_load_checkpoint_never_executes_a_legacy_format_pickle(tmp_path):
    """a non-zip file whose first pickle carries a `__reduce__` payload.  torch's legacy reader goes through
    `pickle_module.load` for the header -- every entry point of the restricted module must be the restricted Unpickler, and the
    permissive fallback refuses non-zip files before reading them."""
    import io
    import pickle
    import pytest
    from fiber_amd.trainer import _restricted_pickle, load_checkpoint

    class Evil:
        def __reduce__(self):
            return (_boom2, ())

    _boom2.called = False
    blob = pickle.dumps(Evil())
    (tmp_path / "evil.ckpt").write_bytes(blob)
    with pytest.raises(Exception):
        load_checkpoint(str(tmp_path / "evil.ckpt"))
    assert not _boom2.called
    # the module-level entry points torch.load may call, directly: the payload's callable resolves to a placeholder
    _restricted_pickle.load(io.BytesIO(blob))
    _restricted_pickle.loads(blob)
    assert not _boom2.called
    # a legacy-format (non-zip) torch file with the payload inside is refused as well
    torch.save({"state_dict": {}, "x": Evil()}, tmp_path / "legacy.ckpt", _use_new_zipfile_serialization=False)
    with pytest.raises(Exception):
        load_checkpoint(str(tmp_path / "legacy.ckpt"))
    assert not _boom2.called


def _boom2():
    _boom2.called = True
    raise RuntimeError("code from a legacy-format checkpoint was executed")


def _boom():
    _boom.called = True
    raise RuntimeError("code from a checkpoint was executed")


_boom.called = False


def test_interval_checkpoints_finish_the_epoch_and_collate_stream_resumes(tmp_path):
    """(a) with an interval check on the last batch the epoch-end save used to be skipped, so last.ckpt recorded the
    finished epoch as unfinished; (b) the collate (MLM-masking) key stream restarts at 0 on manual_seed -- its position is now
    part of the checkpoint and a resumed run continues it."""
    from fiber_amd import ops
    torch.manual_seed(0)
    data = [{"x": torch.randn(8, 4), "y": torch.randn(8, 1)} for _ in range(4)]
    m = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    ops.manual_seed(5)
    keys = [ops.collate_seed() for _ in range(4)]              # four batches collated by the first run
    tr = Trainer(max_steps=None, max_epochs=1, log_every_n_steps=0, default_root_dir=str(tmp_path), val_check_interval=2)
    tr.fit(m, data, val_dataloader=data[:2], device=torch.device("cpu"))
    ck = torch.load(tmp_path / "last.ckpt", weights_only=True)
    assert ck["epoch"] == 1 and ck["global_step"] == 4 and ck["collate_ctr"] == 4
    ops.manual_seed(5)                                          # a fresh process seeds again: the stream is back at 0 ...
    m2 = ToyVal(_cfg(optim_type="adamw", max_steps=8))
    tr2 = Trainer(max_steps=5, log_every_n_steps=0, resume_from_checkpoint=str(tmp_path / "last.ckpt"))
    tr2.fit(m2, data, device=torch.device("cpu"))
    assert tr2.current_epoch >= 1
    nxt = ops.collate_seed()                                    # ... and the resume moved it past the first run's batches
    assert nxt not in keys and (nxt & 0xFFFFFFFF) == 5
