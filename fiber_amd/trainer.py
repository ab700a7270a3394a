"""Minimal stand-in for the `pl.Trainer` kwargs the reference's run.py uses (coarse_grained/run.py:50-75) when
pytorch_lightning is not installed: max_steps / max_epochs, accumulate_grad_batches, DDP over one process per GPU,
step-interval LR scheduler, the Lightning hook order (training_step -> training_epoch_end, validation_step ->
validation_epoch_end, test_step -> test_epoch_end), `last.ckpt` / best-`val/the_metric` checkpoints carrying
`state_dict` (key-compatible with the reference's `ckpt["state_dict"]`), optimizer and scheduler state, and
`resume_from_checkpoint`.  With Lightning present, use `pl.Trainer(accelerator="ddp", ...)` directly -- the module is a
LightningModule."""
import os
import time

import torch

from . import parallel
from .lightning import seed_everything


_CKPT_KEYS = ("state_dict", "optimizer_states", "lr_schedulers", "global_step", "epoch", "best_metric", "collate_ctr")


class _Opaque:
    """Placeholder for every object of a checkpoint whose class is not plain tensor / container data (Lightning's
    `hyper_parameters` AttributeDict, `callbacks` keyed by callback classes, ...): built, never executed, dropped."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Opaque()

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def __hash__(self):
        return id(self)

    def append(self, *a):
        pass

    def extend(self, *a):
        pass

    def update(self, *a, **k):
        pass


class _restricted_pickle:
    """`pickle_module` for torch.load: resolves the globals a tensor checkpoint needs and maps EVERYTHING else to _Opaque --
    no class or function from the file is imported or called, so a foreign checkpoint cannot run code here."""
    import pickle as _p
    __name__ = "fiber_amd.trainer._restricted_pickle"
    _ALLOWED = {
        ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "set"), ("builtins", "frozenset"),
        ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"),
        ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "complex"), ("builtins", "slice"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
        ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
        ("torch.nn.parameter", "Parameter"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
        ("torch.serialization", "_get_layout"),
    }

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _restricted_pickle._ALLOWED:
                return super().find_class(module, name)
            if module == "torch" and (name.endswith("Storage") or name in _restricted_pickle._DTYPES):
                return getattr(torch, name)
            return _Opaque

    _DTYPES = {n for n in dir(torch) if isinstance(getattr(torch, n), torch.dtype)}

    @staticmethod
    def load(f, **kw):                                          # every entry point goes through the restricted Unpickler
        return _restricted_pickle.Unpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        import io
        return _restricted_pickle.Unpickler(io.BytesIO(b), **kw).load()

    UnpicklingError = _p.UnpicklingError
    Pickler = _p.Pickler


def load_checkpoint(path, map_location=None):
    """Read a training checkpoint: this Trainer's own files, or a reference (PyTorch-Lightning 1.3) `.ckpt`
    (coarse_grained/run.py:29-35 ModelCheckpoint, `resume_from_checkpoint` run.py:66).  Own files are tensors / numbers /
    containers and load under `weights_only=True`.  A Lightning file also pickles `hyper_parameters`, `callbacks`, ... through
    classes that are neither allow-listed nor (here) installed: those are read through a restricted unpickler that turns every
    such object into an inert placeholder, and only the entries a resume needs are kept: state_dict, optimizer_states,
    lr_schedulers, global_step, epoch (+ best_metric, collate_ctr of own files)."""
    import pickle
    import zipfile
    try:
        ck = torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as first:
        # Only the zip container (torch >= 1.6, what Lightning 1.3 writes) is read permissively; a legacy-format file
        # would have its header pickles read before any tensor and is refused outright.
        if not zipfile.is_zipfile(path):
            raise ValueError(f"{path}: not a zip-format torch checkpoint; refusing the permissive reader") from first
        try:
            ck = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_restricted_pickle)
        except Exception as second:
            raise second from first
    if not isinstance(ck, dict) or "state_dict" not in ck:
        raise ValueError(f"{path}: not a training checkpoint (no 'state_dict' entry; found {sorted(ck) if isinstance(ck, dict) else type(ck).__name__})")

    def plain(o):                                               # an entry we keep must not contain placeholders
        if isinstance(o, _Opaque) or o is _Opaque:
            return False
        if isinstance(o, dict):
            return all(plain(k) and plain(v) for k, v in o.items())
        if isinstance(o, (list, tuple)):
            return all(plain(v) for v in o)
        return True

    out = {}
    for k in _CKPT_KEYS:
        if k in ck:
            if not plain(ck[k]):
                raise ValueError(f"{path}: checkpoint entry '{k}' holds objects of classes this loader does not read")
            out[k] = ck[k]
    return out


def _to_device(batch, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else [t.to(device) for t in v]
                if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v) for k, v in batch.items()}


class Trainer:
    def __init__(self, max_steps=None, max_epochs=None, accumulate_grad_batches=1, log_every_n_steps=10, default_root_dir=None,
                 resume_from_checkpoint=None, val_check_interval=1.0, seed=None, datamodule=None, **unused):
        if max_steps is None and max_epochs is None:
            max_epochs = 1000                                   # run.py:59 passes 1000 epochs when it bounds the run by steps
        self.max_steps = max_steps
        self.max_epochs = max_epochs
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.log_every_n_steps = log_every_n_steps
        self.default_root_dir = default_root_dir
        self.resume_from_checkpoint = resume_from_checkpoint
        self.datamodule = datamodule
        self.seed = seed
        # pl.Trainer semantics (run.py:69 passes config["val_check_interval"]): float in (0, 1] = fraction of the training
        # epoch between validation runs, int = number of training batches between them
        if isinstance(val_check_interval, float) and not (0.0 < val_check_interval <= 1.0):
            raise ValueError(f"val_check_interval={val_check_interval}: a float must lie in (0, 1]")
        if isinstance(val_check_interval, int) and not isinstance(val_check_interval, bool) and val_check_interval < 1:
            raise ValueError(f"val_check_interval={val_check_interval}: an int must be >= 1")
        self.val_check_interval = val_check_interval
        self.global_step = 0
        self.current_epoch = 0
        self.best_metric = None

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _attach(self, model):
        try:
            model.trainer = self                                # the stand-in LightningModule; a real one exposes a property
        except AttributeError:
            object.__setattr__(model, "_fiber_trainer", self)

    def _done(self):
        return self.max_steps is not None and self.global_step >= self.max_steps

    def _save(self, model, opt, sched, name):
        if not self.default_root_dir:
            return
        os.makedirs(self.default_root_dir, exist_ok=True)
        from . import ops
        torch.save({"state_dict": model.state_dict(), "global_step": self.global_step, "epoch": self.current_epoch,
                    "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched["scheduler"].state_dict()],
                    "best_metric": self.best_metric, "collate_ctr": ops.collate_counter()},
                   os.path.join(self.default_root_dir, name))

    def _resume(self, model, opt, sched, device):
        ck = load_checkpoint(self.resume_from_checkpoint, map_location=device)
        model.load_state_dict(ck["state_dict"], strict=False)
        if ck.get("optimizer_states"):
            opt.load_state_dict(ck["optimizer_states"][0])      # FiberAdamW drops its cached device tables here
        if ck.get("lr_schedulers"):
            sched["scheduler"].load_state_dict(ck["lr_schedulers"][0])
        self.global_step = int(ck.get("global_step", 0))
        self.current_epoch = int(ck.get("epoch", 0))
        self.best_metric = ck.get("best_metric")
        model.global_step = self.global_step
        from . import ops
        ops.mark_weights_dirty()                                # cached bf16 working copies belong to the old weights
        # the MLM-masking key stream of data.device_collate: continue where the saved run stopped (a reference checkpoint has no
        # such entry: one collated batch per micro-batch so far)
        self._resume_collate_ctr = int(ck.get("collate_ctr", self.global_step * self.accumulate_grad_batches))

    # ---- loops ---------------------------------------------------------------------------------------------------
    def fit(self, model, train_dataloader, val_dataloader=None, device=None):
        """`train_dataloader` / `val_dataloader`: iterables of batch dicts (schema of BaseDataset.collate,
        base_dataset.py:172-245).  Returns the last training loss."""
        rank, local, world = parallel.init_distributed()
        self._attach(model)
        if self.seed is not None:
            seed_everything(self.seed)
        if device is None:
            device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
        if hasattr(model, "unused_parameter_names"):
            parallel.freeze_unused(model, model.unused_parameter_names())
        model.to(device).train()
        (opt,), (sched,) = model.configure_optimizers()
        if self.resume_from_checkpoint:
            self._resume(model, opt, sched, device)
            from . import ops
            ops.set_collate_counter(self._resume_collate_ctr)
        net = parallel.wrap_ddp(model, device)
        if device.type == "cuda":
            from . import ops as _ops
            # one launch folds the split weight gradients at the end of backward(): valid when every backward starts from .grad = None
            # in one process (ops.set_fold_defer)
            _ops.set_fold_defer(world == 1 and self.accumulate_grad_batches == 1)
        opt.zero_grad(set_to_none=True)
        micro, t0, last, step0 = 0, time.time(), None, self.global_step

        def run_validation():
            if val_dataloader is None:
                return
            metric = self.validate(model, val_dataloader, device=device)
            if rank == 0 and metric is not None and (self.best_metric is None or metric > self.best_metric):
                self.best_metric = metric                       # ModelCheckpoint(monitor="val/the_metric", mode="max", save_top_k=1)
                self._save(model, opt, sched, "best.ckpt")
            if rank == 0:
                self._save(model, opt, sched, "last.ckpt")      # save_last=True

        every = None                                            # validate every `every` training batches inside the epoch
        vci = self.val_check_interval
        if isinstance(vci, int) and not isinstance(vci, bool):
            every = vci
        elif vci < 1.0:
            if not hasattr(train_dataloader, "__len__"):
                raise ValueError("a fractional val_check_interval needs a training dataloader with a length")
            every = max(1, int(len(train_dataloader) * vci))
        while not self._done() and (self.max_epochs is None or self.current_epoch < self.max_epochs):
            seen = 0
            for batch_idx, batch in enumerate(train_dataloader):
                seen += 1
                batch = _to_device(batch, device)
                sync = (micro + 1) % self.accumulate_grad_batches == 0
                ctx = net.no_sync() if (not sync and hasattr(net, "no_sync")) else _null()
                with ctx:
                    if net is model and hasattr(model, "training_step"):
                        loss = model.training_step(batch, batch_idx)
                    else:                                       # DDP must see forward() itself to arm its reducer
                        if "loss_names" in getattr(getattr(model, "hparams", None), "config", {}):
                            from .modules import fiber_utils
                            fiber_utils.set_task(model)
                        from . import ops
                        ops.set_rng_step(self.global_step)
                        out = net(batch)
                        loss = sum(v for k, v in out.items() if "loss" in k)
                    loss = loss / self.accumulate_grad_batches
                    loss.backward()
                micro += 1
                if sync:
                    opt.step()
                    sched["scheduler"].step()
                    opt.zero_grad(set_to_none=True)
                    self.global_step += 1
                    model.global_step = self.global_step
                    last = loss.detach()
                    if rank == 0 and self.log_every_n_steps and self.global_step % self.log_every_n_steps == 0:
                        print(f"step {self.global_step}: loss {float(last) * self.accumulate_grad_batches:.4f} "
                              f"({(time.time() - t0) / max(1, self.global_step - step0):.3f} s/step)", flush=True)
                    if self._done():
                        break
                if every is not None and seen % every == 0:
                    run_validation()
            if seen == 0:
                raise ValueError("Trainer.fit: the training dataloader yielded no batch")
            if hasattr(model, "training_epoch_end"):
                model.training_epoch_end([])
            self.current_epoch += 1
            if every is None:                                   # val_check_interval = 1.0: the check at the end of the epoch
                run_validation()
            if rank == 0 and (every is not None or val_dataloader is None):
                # interval checks (pl.Trainer runs no additional epoch-end validation for them): the checkpoints written inside
                # the epoch carry the epoch as unfinished -- re-save so that a resume does not repeat it; without a validation
                # loader this is the epoch's only `last.ckpt`
                self._save(model, opt, sched, "last.ckpt")
        return last

    @torch.no_grad()
    def validate(self, model, dataloader, device=None):
        """validation_step over the loader, then validation_epoch_end; returns `val/the_metric` if the module logged it."""
        self._attach(model)
        device = device or next(model.parameters()).device
        was_training = model.training
        model.eval()
        outs = [model.validation_step(_to_device(b, device), i) for i, b in enumerate(dataloader)]
        model.validation_epoch_end(outs)
        model.train(was_training)
        value = getattr(model, "logged", {}).get("val/the_metric")
        return float(value) if value is not None else None

    @torch.no_grad()
    def test(self, model, dataloader, device=None):
        """test_step over the loader, then test_epoch_end (run.py:75)."""
        self._attach(model)
        device = device or next(model.parameters()).device
        model.eval()
        outs = [model.test_step(_to_device(b, device), i) for i, b in enumerate(dataloader)]
        model.test_epoch_end(outs)
        return outs


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
