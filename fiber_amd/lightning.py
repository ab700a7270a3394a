"""pytorch_lightning surface used by the hot path (fiber_module.py:26-29, objectives.py:38-39, fiber_utils.py:254-261).

If pytorch_lightning is importable it is used unchanged; otherwise a minimal stand-in provides exactly the members the
path touches: save_hyperparameters() -> self.hparams.config, log(), device, trainer, global_step.
"""
import inspect
import types

import torch
import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_PL = True
except Exception:  # noqa: BLE001
    HAVE_PL = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = types.SimpleNamespace()
            self.trainer = None
            self.global_step = 0
            self.logged = {}

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            for name in args.args:
                if name != "self":
                    setattr(self.hparams, name, args.locals[name])

        def log(self, name, value, **kw):
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        @property
        def device(self):
            return next(self.parameters()).device
