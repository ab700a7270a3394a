"""CPU, world_size 2, gloo: the data-parallel wiring used by bench.py (fiber_amd/parallel.py).

The HIP path cannot run without a GPU, so the DP logic is exercised with a small plain-torch module that has the same
structure hazards as the real one: a parameter that is never used (must be frozen for a static reducer) and a parameter
used twice per step.  Checks: averaged gradients == single-process gradients over the concatenated batch, and weights
stay identical across ranks after optimizer steps."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

from mp_util import free_port, run_bounded, spawn_bounded


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Linear(4, 4)          # like vit_model.norm / text pooler on the fused path

    def forward(self, batch):
        h = torch.tanh(self.a(batch["x"]))
        h2 = torch.tanh(self.a(batch["x2"]))   # shared weights used twice (MLM + ITM passes)
        return {"loss": (self.b(h) - batch["y"]).pow(2).mean() + self.b(h2).pow(2).mean()}


def _data(n, seed):
    g = torch.Generator().manual_seed(seed)
    return {"x": torch.randn(n, 8, generator=g), "x2": torch.randn(n, 8, generator=g), "y": torch.randn(n, 4, generator=g)}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fiber_amd import parallel
    r, _, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    m = Toy()
    assert parallel.freeze_unused(m, ["unused.weight", "unused.bias"]) == 2
    net = parallel.wrap_ddp(m)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.1)
    full = _data(8, 1)
    shard = {k: v[rank * 4:(rank + 1) * 4] for k, v in full.items()}
    net(shard)["loss"].backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    opt.step()
    flat = torch.cat([p.detach().flatten() for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"grads": grads, "same": bool(torch.equal(gathered[0], gathered[1]))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_match_single_process(tmp_path):
    port = free_port()
    out = str(tmp_path / "r0.pt")
    spawn_bounded(_worker, (2, port, out), nprocs=2, deadline_s=150)
    res = torch.load(out)
    assert res["same"], "weights diverged across ranks"
    ref = Toy()
    ref(_data(8, 1))["loss"].backward()
    for k, p in ref.named_parameters():
        if k.startswith("unused"):
            assert k not in res["grads"]
            continue
        torch.testing.assert_close(res["grads"][k], p.grad, rtol=1e-5, atol=1e-6)


def _worker_collectives(rank, world, port, out):
    """World-2 behaviour of the path's OTHER collectives: ITC's concat_all_gather + _dequeue_and_enqueue
    (fiber_module.py:12-24,181-222), the metric-state reduction of epoch_wrapup (my_metrics.py dist_reduce_fx="sum"), DDP with
    the bf16 gradient-compression hook and no_sync gradient accumulation."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fiber_amd import parallel
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_module, fiber_utils
    from oracle import cases
    parallel.init_distributed("gloo")
    res = {}
    # 1. concat_all_gather: rank-major concatenation, no gradient
    t = torch.full((2, 3), float(rank + 1))
    g = fiber_module.concat_all_gather(t)
    res["gather"] = g.clone()
    # 2. ITC queues: every rank enqueues the GLOBAL batch (4 samples from 2 ranks), pointer / total advance by the global size
    m = FIBERTransformerSS(make_config(**dict(cases.TINY, loss_names={"mlm": 1, "itm": 1, "itc": 1}, itc_queue_size=6)))
    D = m.image_queue.shape[0]
    imgs = torch.full((2, 3, cases.TINY["image_size"], cases.TINY["image_size"]), float(10 + rank))
    ids = torch.full((2, cases.TINY["max_text_len"]), 100 + rank, dtype=torch.long)
    for rnd in range(2):                                      # second round wraps the 6-slot queue
        m._dequeue_and_enqueue(torch.full((2, D), float(rank + 1 + 10 * rnd)), torch.full((2, D), -float(rank + 1 + 10 * rnd)),
                               imgs + rnd, ids + rnd, torch.ones_like(ids))
    res["queue_ptr"], res["queue_total"] = int(m.queue_ptr), int(m.queue_total)
    res["image_queue_row0"] = m.image_queue[0].clone()
    res["text_ids_col0"] = m.text_input_queue[:, 0].clone()
    # 3. metric state summed over ranks at epoch end
    acc = fiber_utils.Accuracy()
    acc(torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]]), torch.tensor([0, 1, 1]) if rank == 0 else torch.tensor([1, 0, -100]))
    res["acc"] = float(acc.compute())                         # rank 0: 2 of 3, rank 1: 0 of 2 -> 2 / 5
    # 4. DDP + bf16 compression hook + no_sync accumulation == single process on the concatenated micro-batches
    toy = Toy()
    parallel.freeze_unused(toy, ["unused.weight", "unused.bias"])
    net = parallel.wrap_ddp(toy, bf16_grads=True)
    full = [_data(8, 1), _data(8, 2)]
    with net.no_sync():
        (net({k: v[rank * 4:(rank + 1) * 4] for k, v in full[0].items()})["loss"] / 2).backward()
    (net({k: v[rank * 4:(rank + 1) * 4] for k, v in full[1].items()})["loss"] / 2).backward()
    res["grads"] = {k: p.grad.clone() for k, p in toy.named_parameters() if p.grad is not None}
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gather_queue_metrics_bf16_hook_no_sync(tmp_path):
    port = free_port()
    out = str(tmp_path / "r0.pt")
    spawn_bounded(_worker_collectives, (2, port, out), nprocs=2, deadline_s=150)
    res = torch.load(out)
    assert torch.equal(res["gather"], torch.tensor([[1.0] * 3] * 2 + [[2.0] * 3] * 2))
    # 8 samples through a 6-slot queue: pointer (0 + 4 + 4) % 6 = 2, total 8; slots 4,5,0,1 hold round 2 (11,11,12,12), 2,3 round 1 (2,2)
    assert res["queue_ptr"] == 2 and res["queue_total"] == 8
    assert res["image_queue_row0"].tolist() == [12.0, 12.0, 2.0, 2.0, 11.0, 11.0]
    assert res["text_ids_col0"].tolist() == [102, 102, 101, 101, 101, 101]
    assert abs(res["acc"] - 2 / 5) < 1e-6
    ref = Toy()
    for d in (_data(8, 1), _data(8, 2)):
        (ref(d)["loss"] / 2).backward()
    for k, p in ref.named_parameters():
        if not k.startswith("unused"):
            torch.testing.assert_close(res["grads"][k], p.grad, rtol=2e-2, atol=2e-3)       # bf16 on the wire


def test_unused_parameter_list_is_consistent_with_reference_golden(golden):
    """unused_parameter_names() == the parameters that received no gradient in the REFERENCE run (golden fixture)."""
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS
    from oracle import cases
    for name in ("path_tiny", "path_swin_t"):
        pc, gold = cases.PATH_CASES[name], golden(name)
        m = FIBERTransformerSS(make_config(**pc["config"]))
        mine = {n for n in m.unused_parameter_names() if not n.startswith("rank_output.")}
        ref = {n for n in gold["unused_params"].tolist() if not n.startswith("rank_output.")}
        assert mine == ref, (sorted(mine - ref)[:5], sorted(ref - mine)[:5])


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads on a CPU-only host and exports every function include/fiber_hip.h declares."""
    import re
    from fiber_amd import lib
    if not os.path.isfile(lib.LIB_PATH):
        import shutil
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("libfiber_hip.so not built and no hipcc here")
        import __graft_entry__
        __graft_entry__.build()          # cross-compiles for gfx950 without a GPU
    l = lib.load()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fiber_hip.h")).read()
    declared = set(re.findall(r"\b(?:int|long)\s+(fiber_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(l, name), f"{name} declared in fiber_hip.h but not exported"
    assert declared == set(lib.exported_symbols()), declared ^ set(lib.exported_symbols())


def _queue_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fiber_amd import parallel
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS
    from oracle import cases
    parallel.init_distributed("gloo")
    cfg = dict(cases.TINY, loss_names={"mlm": 1, "itm": 1, "itc": 1}, itc_queue_size=10, image_size=32)
    torch.manual_seed(0)
    m = FIBERTransformerSS(make_config(**cfg))
    hs, S, B = cfg["hidden_size"], cfg["max_text_len"], 3
    for step in range(2):                               # 2 steps x 2 ranks x 3 samples = 12 > queue_size 10: wraps
        g = torch.Generator().manual_seed(10 * step + rank)
        m._dequeue_and_enqueue(torch.randn(B, hs, generator=g), torch.randn(B, hs, generator=g),
                               torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 99, (B, S), generator=g),
                               torch.randint(0, 2, (B, S), generator=g))
    state = {k: getattr(m, k).clone() for k in ("image_queue", "text_queue", "image_input_queue", "text_input_queue",
                                                "text_input_mask_queue", "queue_ptr", "queue_total")}
    torch.save(state, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_itc_queue_all_gather_world2(tmp_path):
    """fiber_module.py:181-222 under 2 ranks: every rank enqueues the all-gathered batch in rank order, so the queues stay
    identical across ranks; queue_ptr wraps, queue_total keeps counting."""
    port = free_port()
    out = str(tmp_path / "q")
    spawn_bounded(_queue_worker, (2, port, out), nprocs=2, deadline_s=150)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a["queue_ptr"]) == 2 and int(a["queue_total"]) == 12
    # slot contents: step 0 fills slots 0..5 (rank 0 then rank 1), step 1 fills 6..9 and wraps into 0..1
    hs, B = 64, 3
    feats = {}
    for step in range(2):
        for rank in range(2):
            g = torch.Generator().manual_seed(10 * step + rank)
            feats[(step, rank)] = torch.randn(B, hs, generator=g)
    assert torch.equal(a["image_queue"][:, 6:9], feats[(1, 0)].T)          # step 1, rank 0 -> slots 6, 7, 8
    assert torch.equal(a["image_queue"][:, 9], feats[(1, 1)][0])           # step 1, rank 1 -> slots 9, 0, 1
    assert torch.equal(a["image_queue"][:, 0:2], feats[(1, 1)][1:].T)
    assert torch.equal(a["image_queue"][:, 2:6], torch.cat([feats[(0, 0)], feats[(0, 1)]])[2:6].T)   # survivors of step 0


def _run_bench(args, extra_env=None, timeout=240):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return run_bounded([sys.executable, os.path.join(root, "bench.py")] + args, timeout, cwd=root, env=env)


def test_bench_gpus_n_spawns_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher must itself start 2 ranks (one per GPU; reference: Lightning DDP spawns them,
    coarse_grained/run.py:50-54) and form a 2-rank process group.  --dry-run stops after the rendezvous + one all-reduce, which
    is all a machine without GPUs can execute; tests/test_hip_ddp.py runs the full self-spawned bench on a GPU."""
    import json
    res = _run_bench(["--gpus", "2", "--dry-run"], {"FIBER_DIST_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]               # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["allreduce_of_ones"] == 2.0


def test_bench_gpus_disagreeing_with_the_launcher_fails_loudly():
    res = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "--gpus 2" in res.stderr and "WORLD_SIZE=4" in res.stderr, res.stderr[-500:]
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
