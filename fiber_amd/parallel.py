"""Data-parallel wiring for the fused path: one process per GPU, RCCL (backend "nccl") over xGMI.

The reference trains with Lightning DDP (`pl.Trainer(accelerator="ddp")`, coarse_grained/run.py:50-54), i.e. bucketed
gradient all-reduce overlapped with backward.  The only collective on this path is that all-reduce.
"""
import os
from datetime import timedelta

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
            # one node: rendezvous and gloo pairs over the loopback interface -- the container hostname may resolve to an
            # address no local process can reach, and gloo then waits for its full mesh until the group timeout
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        # bounded rendezvous and collectives (torch's default is 30 minutes: one unlucky rendezvous must not eat a whole run)
        timeout = timedelta(seconds=int(os.environ.get("FIBER_DIST_TIMEOUT", "120")))
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout, **kw)
    return rank, local, world


def freeze_unused(model, names):
    """requires_grad=False for parameters that never get a gradient on the fused path, so DDP's reducer is static
    (find_unused_parameters=False) and no bucket waits for a gradient that never comes."""
    names = set(names)
    n = 0
    for k, p in model.named_parameters():
        if k in names:
            p.requires_grad_(False)
            n += 1
    return n


def wrap_ddp(model, device=None, bucket_cap_mb=64, bf16_grads=None):
    """DistributedDataParallel with bucket views (gradients are written straight into the all-reduce buckets).
    Bucket ORDER: with find_unused_parameters=False (the never-used parameters are frozen, freeze_unused) torch's reducer
    rebuilds its buckets after the first iteration in the order the gradients actually became ready, i.e. reverse execution
    order of the interleaved image / text stacks (heads and Swin stage 3 / text layers 10-11 first, patch embedding last) --
    the definition order of `model.parameters()` only matters for step 0.
    bf16_grads (default OFF: the reference trains with an fp32 all-reduce; env FIBER_DDP_BF16=1 or the argument turn it on):
    all-reduce the buckets in bf16 (`bf16_compress_hook`: 0.56 GB instead of 1.13 GB per step over xGMI)
    -- the sum is then formed in bf16 on the wire and decompressed into the fp32 gradient views.  The fp32 ring moves
    2*(7/8)*1.13 GB over one ~153 GB/s xGMI link = ~13 ms per step, hidden under a 320-ms backward at the bench batch."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False,
                                                    gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
    if bf16_grads is None:
        env = os.environ.get("FIBER_DDP_BF16")
        bf16_grads = env == "1"
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    inner = default_hooks.bf16_compress_hook if bf16_grads else default_hooks.allreduce_hook
    on_gpu = device is not None and device.type == "cuda"

    def hook(state, bucket):
        # The text stack runs (forward AND backward) on the module's second HIP stream, so one bucket can hold gradients written
        # on two streams, while the reducer orders the all-reduce only behind the stream of the LAST gradient that arrived.
        # Every gradient of a ready bucket has been enqueued by now: wait for both streams' tails before the collective.
        if on_gpu:
            from . import ops
            ops.join_wgrad_stream()                          # (weight gradients on their own stream, FIBER_WGRAD_STREAM=1)
            cur = torch.cuda.current_stream()
            for st in (getattr(model, "_side_stream", None), getattr(model, "_main_stream", None)):
                if st is not None and st != cur:
                    cur.wait_stream(st)
        return inner(state, bucket)

    ddp.register_comm_hook(state=None, hook=hook)
    return ddp
