"""One training step (forward, backward, AdamW) as a captured hipGraph -- for the small per-GPU batches the reference's
README trains with (per_gpu_batchsize 8, coarse_grained/README.md:35), where the ~2,000 kernel launches of a step are bound
by the host (Python autograd nodes + ctypes launches, ~15-20 us each) and not by the GPU.

Everything a replay must see change lives in device memory:
  * dropout / DropPath keys -- per-step base in an int64 scalar (ops.enable_graph_rng), call-site part baked by value;
  * learning rate and bias-corrected step size of every parameter group (FiberAdamW.enable_graph_mode);
  * the batch -- static tensors the caller overwrites in place (`GraphedTrainStep.batch`).
Gradients are persistent (zeroed inside the graph, accumulated into by autograd) so that the optimizer's device pointer tables
stay valid; the bf16 working / transposed weight copies are rewritten by kernels that are part of the captured step.
Single process only: under DDP the reducer's bucket logic runs on the host, so N > 1 runs stay eager.
"""
import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, scheduler, batch, warmup=2):
        """`batch`: dict of device tensors (the static inputs); must contain `itm_labels_override` when ITM is active (the
        reference's torch.randperm runs on the CPU generator and cannot be captured -- redraw it on the device per step)."""
        self.model, self.opt, self.sched, self.batch = model, optimizer, scheduler, batch
        dev = next(model.parameters()).device
        self.step_index = int(getattr(model, "global_step", 0) or 0)
        if not getattr(optimizer, "_tables", None):          # the optimizer's device tables are built by its first step
            self.model.global_step = self.step_index
            ops.set_rng_step(self.step_index)
            optimizer.zero_grad(set_to_none=False)
            model.training_step(batch, self.step_index).backward()
            optimizer.step()
            self._post()
        ops.enable_graph_rng(dev)
        optimizer.enable_graph_mode()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                        # eager warm-up on a side stream (torch.cuda.graph requirement)
            for _ in range(max(1, warmup)):
                self._pre()
                self._one()
                self._post()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._pre()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._one()
        self.graph.replay()                                  # capture records, it does not execute: run the prepared step
        self._post()

    def _one(self):
        self.opt.zero_grad(set_to_none=False)                # persistent gradients: stable addresses for the optimizer tables
        loss = self.model.training_step(self.batch, self.step_index)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _pre(self):
        self.model.global_step = self.step_index
        ops.set_rng_step(self.step_index)                    # rewrites the device-resident key base
        self.opt.prepare_replay()                            # step counters, lr and step size -> device

    def _post(self):
        self.sched["scheduler"].step() if isinstance(self.sched, dict) else self.sched.step()
        self.step_index += 1

    def __call__(self):
        self._pre()
        self.graph.replay()
        self._post()
        return self.loss

    def close(self):
        self.opt.disable_graph_mode()
        ops.disable_graph_rng()
