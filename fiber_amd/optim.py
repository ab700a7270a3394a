"""AdamW for the fused path on the HIP kernel of csrc/optim.hip: one launch per parameter group, bf16 working copies of
the weights refreshed in the same pass.  The update rule is transformers 4.6.0 `AdamW(correct_bias=True)` (reference
fiber_utils.py:248-252) in ITS form -- eps added to the un-corrected sqrt(v), weight decay applied after the Adam update --
which differs from torch.optim.AdamW for small gradients / early steps.  `HFAdamW` is the same rule in plain torch for host
tensors (CPU wiring tests).  Parameter groups, `lr` scheduling through LambdaLR, state_dict / load_state_dict work as for
any torch optimizer.

Host side: the pointer / size / chunk tables of a group live on the device and are rebuilt only when something moved (a
gradient was re-allocated, a bf16 copy appeared); the per-step work is one pass over the parameters comparing addresses."""
import torch

from . import lib, ops


class HFAdamW(torch.optim.Optimizer):
    """transformers 4.6.0 AdamW(correct_bias=True), restated (optimization.py of that release): used where the HIP kernel
    cannot run (host tensors)."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                g = p.grad
                st["exp_avg"].mul_(b1).add_(g, alpha=1.0 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                step_size = group["lr"] * (1.0 - b2 ** st["step"]) ** 0.5 / (1.0 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], st["exp_avg_sq"].sqrt().add_(group["eps"]), value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])
        return loss


class FiberAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._chunk = None
        self._tables = {}                       # group index -> cached device tables
        self.rebuilds = 0                       # how often a table had to be rebuilt (diagnostics)
        self._hyper = None                      # graph mode: device float [groups, 2] = {lr, bias-corrected step size}

    # ---- hipGraph support ---------------------------------------------------------------------------------------------
    # A captured step bakes every by-value kernel argument.  In graph mode the two arguments that change per step -- the
    # group's learning rate and its bias-corrected step size -- live in device memory; `prepare_replay()` advances the step
    # counters on the host, recomputes them from the (scheduler-updated) param_groups and uploads them ahead of the replay.
    def enable_graph_mode(self):
        if not self._tables:
            raise lib.FiberHipError("FiberAdamW.enable_graph_mode: take one eager step first (device tables not built yet)")
        dev = next(p for g in self.param_groups for p in g["params"]).device
        n = len(self.param_groups)
        self._hyper = torch.zeros((n, 2), dtype=torch.float32, device=dev)

    def disable_graph_mode(self):
        self._hyper = None

    def prepare_replay(self):
        # The host runs ahead of the GPU (replays are asynchronous), so the upload must not read host memory that a LATER
        # prepare_replay() may already have overwritten: a FRESH pageable host tensor per call -- the runtime stages a pageable
        # source before copy_() returns (as for the pointer tables in step()), so step k's graph always sees step k's values.
        dev_t = self._hyper
        host_t = torch.zeros(tuple(dev_t.shape), dtype=torch.float32)
        for gi, group in enumerate(self.param_groups):
            tab = self._tables.get(gi)
            if tab is None:
                continue
            step = tab["states"][0]["step"] + 1
            for st in tab["states"]:
                st["step"] = step
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            host_t[gi, 0] = lr
            host_t[gi, 1] = lr * (1.0 - b2 ** step) ** 0.5 / (1.0 - b1 ** step)
        dev_t.copy_(host_t, non_blocking=True)

    def load_state_dict(self, state_dict):
        """Loaded moments and step counters live in NEW tensors / dicts: drop every cached device table."""
        super().load_state_dict(state_dict)
        self._tables = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}
        self.__dict__.setdefault("_chunk", None)
        self.__dict__.setdefault("rebuilds", 0)
        self.__dict__.setdefault("_hyper", None)

    def _state_of(self, p):
        st = self.state[p]
        if st and isinstance(st.get("step"), torch.Tensor):     # a state dict saved by torch.optim.AdamW keeps tensor steps
            st["step"] = int(st["step"].item())
        if not st:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise lib.FiberHipError("FiberAdamW needs contiguous fp32 parameters on a HIP device")
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops.flush_folds()                                    # (no-op unless a backward pass died before its end-of-backward callback)
        if self._chunk is None:
            self._chunk = lib.plain("fiber_adamw_chunk")
        cached = ops.bf16_copy_if_cached
        bumped = False
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            grads = []
            for p in plist:
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise lib.FiberHipError("FiberAdamW needs contiguous fp32 gradients")
                grads.append(g.data_ptr())
            copies = [cached(p) for p in plist]
            # what the cached device table was built from: the parameters AND their storage AND the state dicts (a
            # load_state_dict() swaps self.state's dicts and tensors, model.to() moves storage under an unchanged id)
            members = (tuple(map(id, plist)), tuple(p.data_ptr() for p in plist), tuple(id(self.state[p]) for p in plist))
            key = (members, tuple(grads), tuple(0 if c is None else c.data_ptr() for c in copies))
            tab = self._tables.get(gi)
            if tab is None or tab["key"] is None or tab["key"][0] != key[0]:   # membership / storage / state changed: rebuild
                self.rebuilds += 1
                states = [self._state_of(p) for p in plist]
                sizes = [p.numel() for p in plist]
                chunks = [(i, c) for i, n in enumerate(sizes) for c in range(-(-n // self._chunk))]
                dev = plist[0].device
                fixed = torch.tensor([(p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), 0)
                                      for p, st in zip(plist, states)], dtype=torch.int64)
                tab = {"key": None, "n": len(chunks), "states": states, "fixed": fixed,
                       "table": torch.empty_like(fixed, device=dev),
                       "numel": torch.tensor(sizes, dtype=torch.int64).to(dev),
                       "chunks": torch.tensor(chunks, dtype=torch.int32).to(dev)}
                self._tables[gi] = tab
            if tab["key"] != key:
                # gradients are re-allocated every step under zero_grad(set_to_none=True): re-send the (27 KB) pointer table.
                # A fresh pageable host tensor per upload: the runtime stages it before returning, so nothing the CPU does
                # later (it runs up to a step ahead of the GPU) can touch the bytes in flight.
                host = tab["fixed"].clone()
                host[:, 1] = torch.tensor(key[1], dtype=torch.int64)
                host[:, 4] = torch.tensor(key[2], dtype=torch.int64)
                tab["table"].copy_(host, non_blocking=True)
                tab["key"] = key
            b1, b2 = group["betas"]
            if self._hyper is None:
                step = tab["states"][0]["step"] + 1
                for st in tab["states"]:
                    st["step"] = step
                hyper = None
            else:                                  # graph mode: prepare_replay() owns the counters and the device scalars
                step = max(1, tab["states"][0]["step"])
                hyper = self._hyper[gi].data_ptr()
            lib.call("fiber_adamw_multi_f32", lib.ptr(tab["table"]), lib.ptr(tab["numel"]), lib.ptr(tab["chunks"]), tab["n"],
                     float(group["lr"]), float(group["weight_decay"]), float(b1), float(b2), float(group["eps"]), int(step), hyper)
            ops.restamp_bf16_copies(plist, bump=not bumped)
            bumped = True
        if bumped:
            ops.refresh_transposed_copies()            # every W^T working copy in one launch (233 strided copies per step before)
            ops.refresh_head_major_copies()            # ... and the permuted qkv copies of the window blocks (120 ATen launches per step before)
        return loss
