"""AdamW for the fused path on the HIP kernel of csrc/optim.hip: one launch per parameter group, bf16 working copies of
the weights refreshed in the same pass.  Same update rule as torch.optim.AdamW / transformers 4.6.0 AdamW(correct_bias=True)
(reference fiber_utils.py:248-252); parameter groups, `lr` scheduling through LambdaLR and state_dict work as for any
torch optimizer.

Host side: the pointer / size / chunk tables of a group live on the device and are rebuilt only when something moved (a
gradient was re-allocated, a bf16 copy appeared); the per-step work is one pass over the parameters comparing addresses."""
import torch

from . import lib, ops


class FiberAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._chunk = None
        self._tables = {}                       # group index -> cached device tables
        self.rebuilds = 0                       # how often a table had to be rebuilt (diagnostics)

    def _state_of(self, p):
        st = self.state[p]
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
            key = (tuple(map(id, plist)), tuple(grads), tuple(0 if c is None else c.data_ptr() for c in copies))
            tab = self._tables.get(gi)
            if tab is None or tab["key"][0] != key[0]:            # membership changed: rebuild everything for this group
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
            step = tab["states"][0]["step"] + 1
            for st in tab["states"]:
                st["step"] = step
            b1, b2 = group["betas"]
            lib.call("fiber_adamw_multi_f32", lib.ptr(tab["table"]), lib.ptr(tab["numel"]), lib.ptr(tab["chunks"]), tab["n"],
                     float(group["lr"]), float(group["weight_decay"]), float(b1), float(b2), float(group["eps"]), int(step))
            ops.restamp_bf16_copies(plist, bump=not bumped)
            bumped = True
        return loss
