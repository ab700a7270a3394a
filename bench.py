"""FIBER-Base (Swin-B 384^2 + RoBERTa-base, 40 tokens) MLM+ITM train-step benchmark on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one optimizer step on a synthetic per-GPU batch: two fused-backbone passes (MLM on masked text, ITM on
true/false image pairs), losses, backward, gradient all-reduce (DDP over RCCL, N>1), AdamW + poly LR schedule.
Prints ONE JSON line on rank 0 (metric = whole-job images/s).  Data is synthetic and already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL across processes)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FWD_PAIR = 111.07e9          # fused fwd per image-text pair incl. transforms/poolers
FLOP_MLM_HEAD = 3.14e9
FLOP_STEP_PER_IMAGE = 3 * (2 * FLOP_FWD_PAIR + FLOP_MLM_HEAD)   # = 675.8 GFLOP (fwd + 2x bwd, two passes)
PEAK_BF16_TFLOPS = 2500.0         # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0            # HBM3E spec, MI355X_MICROARCH.md (6.3 TB/s is what a float4 copy achieves)
# Other task configurations of the same path (extras; the default line stays on BASELINE.json's metric).  Algorithmic
# GFLOP per image per optimizer step, from SURVEY.md section 8(d): image-only Swin 94.16, text-only RoBERTa 6.85, fused 111.07
# per pair at 384^2 / 40 tokens; 256.51 per pair at 576^2 / 50 tokens.
TASKS = {
    "mlm_itm": dict(named="task_pretrain_mlm_itm", flop=FLOP_STEP_PER_IMAGE,
                    workload="FIBER-Base (Swin-B 384^2 + RoBERTa-base S=40) MLM+ITM pretrain step, fused backbone "
                             "fwd+bwd x2 + heads + AdamW"),
    "mlm_itm_itc": dict(named="task_pretrain_mlm_itm_itc", flop=3 * ((111.07e9 + FLOP_MLM_HEAD) + (94.16e9 + 6.85e9 + 0.27e9) + 3 * 111.07e9),
                        workload="FIBER-Base 384^2 S=40 MLM + ITC (4096-deep queues) + hard-negative ITM (3B pairs) pretrain "
                                 "step: 1 + 3 fused passes, image-only and text-only passes, heads, AdamW"),
    "vqa": dict(named="task_finetune_vqa", flop=3 * 256.51e9,
                workload="FIBER-Base VQAv2 fine-tune step, Swin-B 576^2 (18x18 windows) + RoBERTa-base S=50, BCE over 3129 "
                         "answers, AdamW"),
}


def synth_batch(B, image_size, S, vocab, device, seed):
    """SURVEY.md section 8(d) synthetic batch, generated on device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    img = torch.randn(B, 3, image_size, image_size, device=device, generator=g)
    fimg = torch.randn(B, 3, image_size, image_size, device=device, generator=g)
    lens = torch.randint(8, S + 1, (B,), device=device, generator=g)
    pos = torch.arange(S, device=device)[None]
    ids = torch.randint(3, vocab - 2, (B, S), device=device, generator=g)
    ids = torch.where(pos == 0, torch.zeros_like(ids), ids)
    ids = torch.where(pos == lens[:, None] - 1, torch.full_like(ids, 2), ids)
    ids = torch.where(pos >= lens[:, None], torch.ones_like(ids), ids)
    masks = (ids != 1).long()
    special = (ids == 0) | (ids == 1) | (ids == 2)
    pick = (torch.rand(B, S, device=device, generator=g) < 0.15) & ~special
    pick[:, 1] |= ~pick.any(1)
    labels_mlm = torch.where(pick, ids, torch.full_like(ids, -100))
    r = torch.rand(B, S, device=device, generator=g)
    ids_mlm = torch.where(pick & (r < 0.8), torch.full_like(ids, vocab - 1), ids)
    rnd = torch.randint(3, vocab - 2, (B, S), device=device, generator=g)
    ids_mlm = torch.where(pick & (r >= 0.8) & (r < 0.9), rnd, ids_mlm)
    return {"image": [img], "false_image_0": [fimg], "text": ["x"] * B, "text_ids": ids, "text_masks": masks,
            "text_labels": torch.full_like(ids, -100), "text_ids_mlm": ids_mlm, "text_labels_mlm": labels_mlm}


_CPU_SNIPPET = r"""
import json, os, sys, time, torch
sys.path.insert(0, {root!r})
from oracle import cases, detgen, fiber_ref
torch.set_num_threads({threads})
LEGS = {legs!r}


def run(tag, cfg, size, B):
    m = fiber_ref.FiberRef(dict(cfg, text_dropout=0.1, drop_path_rate=0.1)).train()
    for n, p in m.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)
    b = detgen.synth_batch(B, size, 40, 50265, seed=0)
    if "train" in LEGS[tag]:
        for i in range({warm} + {timed}):                     # SURVEY.md 8d: warm-ups + timed steps, MLM + ITM forward + backward
            t = time.time()
            m.zero_grad(set_to_none=True)
            m.training_loss(b, b["itm_labels"]).backward()
            print(json.dumps({{"leg": tag + "/train", "B": B, "step": i, "sec": time.time() - t}}), flush=True)
    if "fwd" in LEGS[tag]:
        m.eval()
        with torch.no_grad():
            for i in range({warm} + {timed}):                 # the fused-backbone forward of one image-text batch (fiber_module.py:224-367)
                t = time.time()
                m.infer(b)
                print(json.dumps({{"leg": tag + "/fwd", "B": B, "step": i, "sec": time.time() - t}}), flush=True)


if "base" in LEGS:
    run("base", cases.SWIN_B, 384, {batch})
if "swin_t" in LEGS:
    run("swin_t", cases.SWIN_T, 224, 4)
"""


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _cpu_run(threads, budget_s, batch, legs, warm=2, timed=5):
    """One child process (killed by PID at its deadline): {leg: (median seconds per iteration, timed iterations, batch)}."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    proc = subprocess.Popen([sys.executable, "-c", _CPU_SNIPPET.format(root=ROOT, threads=threads, batch=batch, legs=legs, warm=warm, timed=timed)],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
    try:
        out, _ = proc.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        proc.kill()                                           # the exact child we started
        out, _ = proc.communicate()
    by_leg = {}
    for line in out.splitlines():
        if line.startswith("{"):
            r = json.loads(line)
            by_leg.setdefault(r["leg"], []).append((r["sec"], r["B"]))
    res = {}
    for leg, rows in by_leg.items():
        secs = [r[0] for r in rows]
        done = sorted(secs[warm:])                            # warm-ups dropped, median of the timed steps that finished
        res[leg] = (done[len(done) // 2] if done else None, len(done), rows[0][1])
    return res


def launches_per_step(batch, budget_s=150.0):
    """Device launches of one steady-state step (kernels of this library, ATen, the BLAS library; copy / fill commands), counted by
    tools/probes/launch_count.py with torch.profiler in a CHILD process under a deadline (killed by PID): a profiler that misbehaves cannot
    take the bench line with it.  The launch structure does not depend on the batch; the child uses a small one to stay short."""
    import subprocess
    torch.cuda.empty_cache()
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "probes", "launch_count.py"), str(batch)],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        out, _ = proc.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        proc.kill()                                           # the exact child we started
        proc.communicate()
        return {"error": f"launch count did not finish in {budget_s:.0f} s"}
    for line in reversed(out.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": f"launch count child exited with {proc.returncode}"}


def cpu_baseline(budget_s=280.0, batch=2):
    """The oracle (CPU restatement of the reference path, fp32 PyTorch) timed on this host's cores, the legs BASELINE.md section 4b
    names (2 warm-ups, median of 5 each), every child process under a hard wall-clock budget (killed by PID at the deadline) so that
    the default bench finishes in minutes:
      8 threads (the survey's container reference point): FIBER-Base 384^2 S=40 B=2 MLM+ITM forward + backward (images/s) and the
        fused-backbone forward alone (ms / image); Swin-T 224^2 + RoBERTa-base B=4 (BASELINE.json configs[0]) forward + backward;
      every physical core, run FIRST with a budget of its own (200 s): the FIBER-Base train leg at B=8, 1 warm-up + 2 timed steps.  On the
        round-6 box (2 x EPYC 9575F, 128 cores; tools/probes/cpu_threads_probe.py) one such step takes 47 s on 128 threads, 21 s on 64,
        12 s on 32 -- the oracle's many small ops get SLOWER with more threads (8 threads at B=2: 3 s per step = the best rate) -- so
        this leg is 3 x 47 s + 40 s of set-up; rounds 4-5 gave it 49 s and it never finished.  Reported when both timed steps finished.
    `value` is the best FIBER-Base train-step rate, `cores` its thread count."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys = avail
    try:                                                      # logical -> physical cores (SMT siblings share a core)
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        per_core = len(sib.replace("-", ",").split(",")) if "," in sib or "-" in sib else 1
        phys = max(1, avail // max(1, per_core))
    except OSError:
        pass
    t8 = min(8, avail)
    all_budget = min(200.0, budget_s * 0.72)
    legs_all = _cpu_run(phys, all_budget, 8, {"base": ["train"]}, warm=1, timed=2) if phys > t8 else {}
    legs8 = _cpu_run(t8, budget_s - (all_budget if phys > t8 else 0.0), batch, {"base": ["train", "fwd"], "swin_t": ["train"]})

    def rate(entry):                                          # images/s of a (seconds, n, B) entry
        return entry[2] / entry[0] if entry and entry[0] else None

    parts, cands = [], []
    e = legs8.get("base/train")
    if rate(e):
        parts.append(f"{t8} threads: FIBER-Base 384^2 B={e[2]} train {rate(e):.3f} images/s (median of {e[1]})")
        cands.append((rate(e), t8))
    else:
        parts.append(f"{t8} threads: FIBER-Base train leg did not finish a timed step")
    e = legs8.get("base/fwd")
    parts.append(f"fused-backbone forward {1e3 * e[0] / e[2]:.0f} ms/image (median of {e[1]})" if e and e[0] else "forward leg did not finish")
    e = legs8.get("swin_t/train")
    parts.append(f"Swin-T 224^2 B=4 (configs[0]) train {rate(e):.2f} images/s (median of {e[1]})" if rate(e) else "Swin-T leg did not finish")
    if legs_all or phys > t8:
        e = legs_all.get("base/train")
        if e and e[0] and e[1] >= 2:
            parts.append(f"{phys} threads: FIBER-Base 384^2 B={e[2]} train {rate(e):.3f} images/s (median of {e[1]})")
            cands.append((rate(e), phys))
        else:
            parts.append(f"{phys} threads, B=8: did not finish 1 warm-up + 2 timed steps inside its {all_budget:.0f} s")
    base = {"unit": "images/s", "kind": "port",
            "sample": f"oracle/fiber_ref.py MLM+ITM, fp32, on {_cpu_model()} ({avail} logical / {phys} physical cores visible), "
                      f"{budget_s:.0f} s budget, 2 warm-ups + median of up to 5: " + "; ".join(parts)}
    legs = {k: {"sec_per_iter": v[0], "timed": v[1], "B": v[2], "threads": t8} for k, v in legs8.items()}
    legs.update({k + f"@{phys}t": {"sec_per_iter": v[0], "timed": v[1], "B": v[2], "threads": phys} for k, v in legs_all.items()})
    base["legs"] = legs
    if not cands:
        return dict(base, value=None, cores=t8)
    v, t = max(cands)
    return dict(base, value=round(v, 4), cores=t)


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one (path, file name) -- (None, None) when there is none."""
    import glob
    import re
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")):
        n = int(re.match(r"r(\d\d)_", os.path.basename(p)).group(1))
        if best is None or n > best[0]:
            best = (n, p)
    return (best[1], "profiles/" + os.path.basename(best[1])) if best else (None, None)


def _dominant_from_profile(B):
    """The (entry point, shape) with the largest share of the step in the committed per-shape profile of this bench
    (profiles/rNN_shape_breakdown.json of the latest round, written by tools/shape_breakdown.py at the bench batch); None when the file
    is missing or was taken at another batch."""
    try:
        prof = json.load(open(_latest_profile("shape_breakdown.json")[0]))
    except (OSError, ValueError, TypeError):
        return None
    if prof.get("batch") != B:
        return None
    rows = [r for r in prof["rows"] if "kind" in r]
    return (rows[0], prof["single_stream_step_ms"]) if rows else None


def time_dominant_kernel(B, device):
    """Live HIP-event timing of the dominant hand-written kernel.  WHICH kernel that is comes from the committed per-shape
    profile (largest ms per step over all (entry point, shape) pairs); without a profile for this batch it falls back to the
    stage-2 MLP fc1 GEMM (bias + GELU + pre-activation store).  Reports the achieved fraction of BOTH rooflines (algorithmic
    bytes / time against HBM, algorithmic FLOPs / time against dense bf16 MFMA); `bound` names the tighter one, i.e. the resource
    whose floor is higher for this shape."""
    from fiber_amd import ops
    dom = _dominant_from_profile(B)
    if dom is None:
        row = {"entry": "fiber_gemm_nt_bf16", "kind": "NT gelu+pre", "shape": [2 * B * 576, 2048, 512], "ms_per_step": None}
        step_ms = None
    else:
        row, step_ms = dom
    M, N, K = row["shape"]
    kind = row["kind"]
    bf = torch.bfloat16
    rs = torch.ones(M // 576 if M % 576 == 0 else 1, device=device)
    rps = M // rs.numel()
    if row["entry"] == "fiber_gemm_tn_bf16":
        dy = torch.randn(M, N, device=device).to(bf)
        x = torch.randn(M, K, device=device).to(bf)
        fn = lambda: ops.wgrad(dy, x, want_bias=True)
        alg_bytes = 2 * M * (N + K) + 4 * N * K
        name = "gemm_tn_kernel<256,64,2,true> + tn_fold_kernel (weight + bias gradient)"
    else:
        x = torch.randn(M, K, device=device).to(bf)
        w = (torch.randn(N, K, device=device) * K ** -0.5).to(bf)
        bias = torch.randn(N, device=device)
        alg_bytes = 2 * (M * K + N * K + M * N)
        if kind.startswith("NT gelu'"):
            aux = torch.randn(M, N, device=device).to(bf)
            dp = "droppath" in kind
            fn = lambda: ops.gemm_nt(x, w, None, None, 2, False, rs if dp else None, rps if dp else 0, aux=aux)
            alg_bytes += 2 * M * N
            name = "gemm_nt_q8_kernel<2,...> ((dY.W2^T) * gelu'(H) fused MLP backward)"
        elif kind == "NT gelu+pre":
            fn = lambda: ops.gemm_nt(x, w, bias, None, 1, True)
            alg_bytes += 2 * M * N
            name = "gemm_nt_q8_kernel<1,false,false> (fc1: bias + GELU + pre-activation store)"
        elif kind.startswith("NT residual"):
            r = torch.randn(M, N, device=device).to(bf)
            dp = "droppath" in kind
            fn = lambda: ops.gemm_nt(x, w, bias, r, 0, False, rs if dp else None, rps if dp else 0)
            alg_bytes += 2 * M * N
            name = "gemm_nt_q8_kernel<0,true,...> (bias + DropPath scale + residual)"
        else:
            fn = lambda: ops.gemm_nt(x, w, bias)
            name = "gemm_nt_q8_kernel<0,false,false> (plain / bias)"
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * M * N * K
    tf = flops / (us * 1e-6) / 1e12
    gbps = alg_bytes / us / 1e3
    # which roofline bounds this shape: the one with the higher floor (time at peak)
    hbm_floor_us, mfma_floor_us = alg_bytes / PEAK_HBM_GBPS / 1e3, flops / PEAK_BF16_TFLOPS / 1e6
    bound = "hbm" if hbm_floor_us >= mfma_floor_us else "mfma"
    out = {"kernel": name, "entry": row["entry"], "kind": kind, "shape": [M, N, K], "us": round(us, 2),
           "share_of_step": (round(row["ms_per_step"] / step_ms, 4) if step_ms else None),
           "selected_from": f"{_latest_profile('shape_breakdown.json')[1]} (largest ms per step)" if dom else "default (no profile at this batch)",
           "bound": bound, "flop_per_byte": round(flops / alg_bytes, 1),
           "achieved": round(gbps if bound == "hbm" else tf, 1), "peak": PEAK_HBM_GBPS if bound == "hbm" else PEAK_BF16_TFLOPS,
           "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round((gbps / PEAK_HBM_GBPS) if bound == "hbm" else (tf / PEAK_BF16_TFLOPS), 4),
           "hbm_GBps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4), "mfma_TFLOPs": round(tf, 1),
           "mfma_frac": round(tf / PEAK_BF16_TFLOPS, 4), "algorithmic_bytes": alg_bytes, "traffic": None}
    try:   # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs on this shape)
        pmc_path, pmc_name = _latest_profile("pmc_kernels.json")
        pmc = json.load(open(pmc_path))["dominant"]
        if pmc["algorithmic_bytes_per_launch"] == alg_bytes and pmc.get("shape") == [M, N, K]:
            out["traffic"] = int(pmc["traffic_bytes_per_launch"])
            out["traffic_from"] = pmc_name
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return out


def time_forward(model, batch, B, device):
    """fwd ms/img: the fused-backbone forward (`infer`, B image-text pairs, no dropout, no autograd graph) on the batch that
    was just trained on; HIP events around 5 passes after 2 warm-ups."""
    was_training = model.training
    model.eval()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for _ in range(2):
            model.infer(batch)
        e0.record()
        for _ in range(5):
            model.infer(batch)
        e1.record()
    torch.cuda.synchronize(device)
    model.train(was_training)
    return e0.elapsed_time(e1) / 5 / B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # warm-up 10, >= 50 timed steps, median + p10/p90
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("FIBER_BENCH_BATCH", "0")),
                    help="per-GPU batch (default: 256 for the headline task, 96 for mlm_itm_itc whose 1+3 fused passes hold "
                         "more activations, 160 for vqa at 576^2)")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="on = capture the whole step (fwd + bwd + AdamW) in a hipGraph and replay it; auto = off = eager.  Measured: "
                         "single-stream eager 40.2 ms vs replay 38.4 at B=8 (50.4 vs 48.1 at B=16: small batches are bound by ~2500 "
                         "short kernels on the GPU, not by the host, and graph mode's persistent gradients add ~300 accumulate "
                         "kernels), but eager with the text stack on its second stream -- the default, not capturable -- is 36.3 ms; "
                         "N > 1 always runs eager (the DDP reducer is host code)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only and dominant-kernel timings (clean rocprof runs)")
    ap.add_argument("--mlm-head", choices=("full", "labelled"), default="full",
                    help="full (default) = the reference's MLM head over every text position; labelled = config mlm_compact_rows: the head "
                         "only on the rows that carry a label (same loss and gradients; -2.8 ms per step at B=256; the head's FLOPs are then "
                         "charged by the labelled fraction)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous check only: start the ranks, form the process group (gloo when no GPU is visible), "
                         "all-reduce one tensor, print {n_gpus, rccl_ranks} and exit -- no model, no timing (tests/test_dp_gloo.py)")
    ap.add_argument("--task", choices=sorted(TASKS), default="mlm_itm",
                    help="default = BASELINE.json's metric; the others are extra configurations of the same path")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = {"mlm_itm": 256, "mlm_itm_itc": 96, "vqa": 160}[args.task]

    # `--gpus N` is the contract: one rank per GPU.  Under a launcher (torchrun sets WORLD_SIZE) the two must agree; without one,
    # N > 1 re-executes this very command line under torch.distributed.run on a free local port (reference: Lightning spawns one
    # process per GPU itself, coarse_grained/run.py:50-54).
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                     f"pass --gpus {os.environ['WORLD_SIZE']} (or launch {args.gpus} ranks)")
    elif args.gpus > 1:
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without a launcher: spawning {args.gpus} ranks ({' '.join(cmd[1:8])} ...)", file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        import torch.distributed as dist
        from fiber_amd import parallel
        parallel.init_distributed(os.environ.get("FIBER_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
        ranks = dist.get_world_size() if dist.is_initialized() else 1
        assert ranks == args.gpus == world, f"process group has {ranks} ranks, --gpus {args.gpus}, WORLD_SIZE {world}"
        t = torch.ones(1, device=torch.device("cuda", local % torch.cuda.device_count()) if torch.cuda.is_available() else "cpu")
        if ranks > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "allreduce_of_ones": t.item()}), flush=True)
        if ranks > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X devices (there is no CPU fallback for the product path)"
    local %= torch.cuda.device_count()                      # (only matters for the 2-ranks-on-1-GPU gloo wiring test)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    from fiber_amd import parallel
    backend = os.environ.get("FIBER_DIST_BACKEND", "nccl")                    # RCCL; "gloo" only for 1-GPU wiring tests
    # the group timeout bounds the rendezvous too: eight ranks paging torch in on a fresh box may arrive a minute or two apart
    # (the tests pin 120 s through the environment; a benchmark run gets the room)
    os.environ.setdefault("FIBER_DIST_TIMEOUT", "600")
    parallel.init_distributed(backend)
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    assert ranks == args.gpus == world, f"process group has {ranks} ranks, --gpus {args.gpus}, WORLD_SIZE {world}"

    from fiber_amd import lib, ops
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils
    lib.load()
    torch.manual_seed(0)
    ops.manual_seed(rank)
    from fiber_amd.config import named_config
    task = TASKS[args.task]
    cfg = named_config(task["named"], per_gpu_batchsize=args.batch, num_gpus=world, max_steps=100000, warmup_steps=10000,
                       draw_false_image=1 if args.task == "mlm_itm" else 0, mlm_compact_rows=args.mlm_head == "labelled")
    model = FIBERTransformerSS(cfg)
    for n, p in model.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)          # reference init is 0: fusion branches would carry no signal
    parallel.freeze_unused(model, model.unused_parameter_names())   # never touched on the fused MLM+ITM path
    model.to(device).train()
    ops.set_fold_defer(world == 1)                          # slabs of the split weight-gradient GEMMs folded in one launch at the end of backward()
    ops.enable_wgrad_stream(model)                          # weight-gradient GEMMs + gradient accumulation on a second stream (before DDP)
    if os.environ.get("FIBER_WGRAD_STREAM") == "2" and world == 1:
        ops.set_wgrad_stream(True)                          # (experiment: the GEMMs alone, gradients taken over by autograd without a copy)
    fiber_utils.set_task(model)
    (opt,), (sched,) = model.configure_optimizers()
    net = parallel.wrap_ddp(model, device)
    batch = synth_batch(args.batch, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], device, seed=rank)
    if args.task == "vqa":                                   # 1-3 soft answers per question (vqav2_dataset schema)
        gq = torch.Generator().manual_seed(rank)
        ks = torch.randint(1, 4, (args.batch,), generator=gq).tolist()
        batch["vqa_labels"] = [torch.randperm(cfg["vqav2_label_size"], generator=gq)[:k].tolist() for k in ks]
        batch["vqa_scores"] = [[(0.3, 0.6, 0.9, 1.0)[int(torch.randint(0, 4, (1,), generator=gq))] for _ in range(k)] for k in ks]

    def eager_step():
        ops.set_rng_step(model.global_step)
        out = net(batch)
        loss = sum(v for k, v in out.items() if "loss" in k)
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        opt.zero_grad(set_to_none=True)
        model.global_step += 1
        return loss

    use_graph = args.graph == "on"          # "auto" = eager: with the text stack on its own stream eager beats replay at every batch (36.3 vs 37.5 ms at B=8)
    use_graph = use_graph and world == 1 and args.task == "mlm_itm" and hasattr(opt, "enable_graph_mode")
    step = eager_step
    if use_graph:
        from fiber_amd.graph import GraphedTrainStep
        # the reference draws the ITM true/false permutation with torch.randperm on the CPU generator (objectives.py:47-48);
        # a captured step needs it on the device: a static label tensor, re-permuted in place ahead of every replay
        base_lab = torch.cat([torch.ones(args.batch // 2), torch.zeros(args.batch - args.batch // 2)]).to(device)
        batch["itm_labels_override"] = base_lab.clone()
        eager_step()                                          # builds the optimizer's device tables
        gstep = GraphedTrainStep(model, opt, sched, batch, warmup=2)

        def step():
            batch["itm_labels_override"].copy_(base_lab[torch.randperm(args.batch, device=device)])
            return gstep()

    for _ in range(args.warmup):
        step()
    # per-step HIP events (recorded on the compute stream, read after the timed region: no extra synchronisation inside it)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_local = dt = time.perf_counter() - t0
    per_rank_ms = [dt_local / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [x.item() / args.steps * 1e3 for x in allt]
        dt = max(x.item() for x in allt)                     # MAX over ranks
    ms = dt / args.steps * 1e3
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))]
    exposed_allreduce_ms = None
    if world > 1 and not args.no_extras:
        # the same step without the gradient all-reduce (DDP no_sync): the difference is the communication time that backward
        # did NOT hide.  Outside the timed region; gradients of these steps are rank-local and are thrown away with the run.
        n_ns = max(3, min(10, args.steps))
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(n_ns):
            with net.no_sync():
                eager_step()
        torch.cuda.synchronize()
        ms_ns = torch.tensor([(time.perf_counter() - t1) / n_ns * 1e3], device=device, dtype=torch.float64)
        dist.all_reduce(ms_ns, op=dist.ReduceOp.MAX)
        exposed_allreduce_ms = round(ms - ms_ns.item(), 3)
    if rank == 0:
        print(f"[bench] {ms:.2f} ms/step, {args.batch * world * args.steps / dt:.1f} images/s, peak HBM "
              f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB allocated / {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB reserved",
              file=sys.stderr, flush=True)
    ips = args.batch * world * args.steps / dt
    lossv = loss.item()
    if rank == 0 and hasattr(opt, "rebuilds"):
        print(f"[bench] optimizer table rebuilds: {opt.rebuilds} over {args.steps + args.warmup} steps x {len(opt.param_groups)} groups",
              file=sys.stderr, flush=True)

    if rank == 0:
        # FLOPs charged per image: with `mlm_compact_rows` the 50265-way MLM head runs on the rows that carry a label only (same loss
        # and gradients, fiber_amd/modules/objectives.py:_mlm_head), so only that fraction of the head is counted
        flop_img, head_rows = task["flop"], "all"
        if args.task in ("mlm_itm", "mlm_itm_itc") and cfg.get("mlm_compact_rows", False) and not use_graph:
            lab = batch["text_labels_mlm"]
            frac = float((lab != -100).float().mean())
            flop_img = task["flop"] - 3 * FLOP_MLM_HEAD * (1.0 - frac)
            head_rows = f"labelled rows only ({frac * 100:.1f} % of {lab.numel()})"
        tf_per_gpu = ips / world * flop_img / 1e12
        res = {
            "metric": "train-step images/sec (384^2, seq40) FIBER-Base" if args.task == "mlm_itm" else
                      f"train-step images/sec FIBER-Base, task {args.task}", "value": round(ips, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": task["workload"], "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "rccl_ranks": ranks,
                       "dist_backend": (backend + (" (RCCL over xGMI)" if backend == "nccl" else "")) if world > 1 else None,
                       "dropout": "reference defaults (text 0.1, DropPath linspace 0..0.1)",
                       "residual_dtype": "fp32" if ops.residual_fp32() else "bf16", "mlm_head_rows": head_rows,
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager (one launch per kernel)"},
            "loss": round(lossv, 4),
            "step_ms": {"p10": round(pct(0.1), 3), "median": round(pct(0.5), 3), "p90": round(pct(0.9), 3),
                        "per_rank_mean": [round(x, 3) for x in per_rank_ms], "allreduce_exposed": exposed_allreduce_ms,
                        "grad_allreduce": (("bf16 buckets (bf16_compress_hook)" if os.environ.get("FIBER_DDP_BF16") == "1" else "fp32 buckets")
                                           + ", 64 MB, reverse execution order after step 0" if world > 1 else None)},
            "roofline": {"bound": "mfma", "achieved": round(tf_per_gpu, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf_per_gpu / PEAK_BF16_TFLOPS, 4), "traffic": None,
                         "basis": f"{flop_img / 1e9:.1f} GFLOP algorithmic per image per step (BASELINE.md section 3 / SURVEY.md "
                                  f"section 8d: {task['flop'] / 1e9:.1f} with the MLM head on every row) / measured step time, per GPU"},
        }
        if not args.no_extras:
            res["fwd_ms_per_image"] = round(time_forward(model, batch, args.batch, device), 4)   # BASELINE.json metric, second half
        if args.task == "mlm_itm" and not args.no_extras:
            res["roofline"]["dominant_kernel"] = time_dominant_kernel(args.batch, device)
            res["roofline"]["traffic"] = res["roofline"]["dominant_kernel"]["traffic"]
        if world == 1 and args.task == "mlm_itm" and not args.no_extras and not use_graph:
            # SURVEY.md section 8d names per-GPU batches 8 (the reference README's example), 16 and 32: the same model and optimizer,
            # 8 warm-up + 10 timed eager steps each, after (and outside) the headline's timed region
            sweep = {}
            batch = None                                     # (release the headline batch)
            for b in (8, 16, 32):
                torch.cuda.empty_cache()                     # (the headline's 170-GB block cache makes small-batch steps 30 % slower)
                if b >= args.batch:
                    continue
                sb = synth_batch(b, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], device, seed=1000 + b)

                def sstep():
                    ops.set_rng_step(model.global_step)
                    out = net(sb)
                    sum(v for k, v in out.items() if "loss" in k).backward()
                    opt.step()
                    sched["scheduler"].step()
                    opt.zero_grad(set_to_none=True)
                    model.global_step += 1
                for _ in range(8):
                    sstep()
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(10):
                    sstep()
                torch.cuda.synchronize()
                sec = (time.perf_counter() - ts) / 10
                sweep[str(b)] = {"images_per_s": round(b / sec, 1), "ms_per_step": round(sec * 1e3, 2),
                                 "frac_of_mfma_peak": round(b / sec * flop_img / 1e12 / PEAK_BF16_TFLOPS, 4)}
            res["extra"] = {"batch_sweep": sweep, "batch_sweep_note": "per-GPU batches of SURVEY.md 8d (8 = coarse_grained/README.md:35), "
                            "same process after torch.cuda.empty_cache(), 8 warm-up + 10 timed eager steps each, wall clock around a device synchronise"}
            res["extra"]["launches_per_step"] = launches_per_step(32)
        if world == 1 and not args.no_cpu_baseline and args.task == "mlm_itm":
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
