"""pytorch_lightning surface used by the hot path (fiber_module.py:26-29, 480-520, objectives.py:38-39, fiber_utils.py:44-140,
254-261; run.py:21).

If pytorch_lightning is importable it is used unchanged; otherwise a minimal stand-in provides exactly the members the path
touches: save_hyperparameters() -> self.hparams.config, log() / log_dict(), device, trainer, global_step, logger, and
seed_everything().
"""
import inspect
import random
import types

import torch
import torch.nn as nn


def _seed_path_rng(seed):
    """The dropout / DropPath streams of the HIP kernels are counter based (fiber_amd/ops.py): seed them with the run's seed
    (+ rank, as DistributedSampler-style per-rank streams) so that `seed` in the config actually selects the masks."""
    import os
    from . import ops
    ops.manual_seed(int(seed) + int(os.environ.get("RANK", "0")))


try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_PL = True

    def seed_everything(seed):
        pl.seed_everything(seed)
        _seed_path_rng(seed)
        return seed
except Exception:  # noqa: BLE001
    HAVE_PL = False

    def seed_everything(seed):
        """pl.seed_everything (run.py:21) + the path's own counter-based streams."""
        seed = int(seed)
        random.seed(seed)
        try:
            import numpy as np
            np.random.seed(seed % (2 ** 32))
        except ImportError:
            pass
        torch.manual_seed(seed)
        _seed_path_rng(seed)
        return seed

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = types.SimpleNamespace()
            self.trainer = None
            self.logger = None
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

        def log_dict(self, dictionary, **kw):
            for k, v in dictionary.items():
                self.log(k, v, **kw)

        @property
        def device(self):
            return next(self.parameters()).device
