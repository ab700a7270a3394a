"""Minimal stand-in for the `pl.Trainer` kwargs the reference's run.py uses (coarse_grained/run.py:50-70) when
pytorch_lightning is not installed: max_steps, accumulate_grad_batches, DDP over one process per GPU, step-interval LR
scheduler, optional checkpoint of `state_dict` (key-compatible with the reference's `ckpt["state_dict"]`).
With Lightning present, use `pl.Trainer(accelerator="ddp", ...)` directly -- the module is a LightningModule."""
import time

import torch

from . import parallel


class Trainer:
    def __init__(self, max_steps=100, accumulate_grad_batches=1, log_every_n_steps=10, default_root_dir=None, **unused):
        self.max_steps = max_steps
        self.max_epochs = None
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.log_every_n_steps = log_every_n_steps
        self.default_root_dir = default_root_dir
        self.datamodule = None
        self.global_step = 0

    def fit(self, model, train_dataloader, device=None):
        """`train_dataloader`: iterable of batch dicts (schema of BaseDataset.collate, base_dataset.py:172-245)."""
        rank, local, world = parallel.init_distributed()
        model.trainer = self
        if device is None:
            device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
        if hasattr(model, "unused_parameter_names"):
            parallel.freeze_unused(model, model.unused_parameter_names())
        model.to(device).train()
        (opt,), (sched,) = model.configure_optimizers()
        net = parallel.wrap_ddp(model, device)
        opt.zero_grad(set_to_none=True)
        micro, t0, last = 0, time.time(), None
        while self.global_step < self.max_steps:
            for batch in train_dataloader:
                batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else [t.to(device) for t in v]
                             if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v) for k, v in batch.items()}
                sync = (micro + 1) % self.accumulate_grad_batches == 0
                ctx = net.no_sync() if (not sync and hasattr(net, "no_sync")) else _null()
                with ctx:
                    if hasattr(model, "current_tasks") and not model.current_tasks:
                        from .modules import fiber_utils
                        fiber_utils.set_task(model)
                    out = net(batch)
                    loss = sum(v for k, v in out.items() if "loss" in k) / self.accumulate_grad_batches
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
                              f"({(time.time() - t0) / self.global_step:.3f} s/step)", flush=True)
                    if self.global_step >= self.max_steps:
                        break
        if rank == 0 and self.default_root_dir:
            torch.save({"state_dict": model.state_dict(), "global_step": self.global_step}, f"{self.default_root_dir}/last.ckpt")
        return last


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
