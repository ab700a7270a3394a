"""How much of the bf16 path's deviation on the scalar fusion gates (alpha_i2t, alpha_t2i) is forced by bf16 STORAGE of the two tensors their
gradient is a dot product of?  TEST / ANALYSIS INFRASTRUCTURE ONLY (CPU, oracle only).

d alpha = <dOut, branch> over every element of a [B, L, C] tensor (swin_transformer.py:259, roberta.py:483).  The HIP path accumulates that dot
product in fp32 (csrc/elementwise.hip stream_add_bwd_kernel) but reads dOut and branch as the bf16 tensors they are in HBM.  Here the oracle's own
fp32 tensors at every gate are captured, and the dot product is re-evaluated in fp64 with (a) both operands exact, (b) both rounded to bf16
(fresh rounding of exact values: the FLOOR of any path that stores them in bf16), (c) only the gradient, (d) only the branch rounded, next to the
cancellation ratio |sum| / sum|terms|.  Several batches per configuration give the spread.

    python -m oracle.gate_grad_study [--config tiny|swin_t] [--batches 4]      -> profiles/r05_gate_grad_study_<config>.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases, detgen  # noqa: E402
from oracle import fiber_ref as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny", choices=["tiny", "swin_t"])
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--B", type=int, default=2)
    a = ap.parse_args()
    cfg = dict(cases.TINY if a.config == "tiny" else cases.SWIN_T)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = detgen.fill_(R.FiberRef(cfg).eval())
    c = ref.config
    gates = {}                                            # parameter name -> the module whose output is the gated branch
    for n, m in ref.named_modules():
        if hasattr(m, "alpha_i2t") and hasattr(m, "proj_i2t"):
            gates[n + ".alpha_i2t"] = m.proj_i2t
        if hasattr(m, "alpha_t2i") and hasattr(m, "crossattention_t2i"):
            gates[n + ".alpha_t2i"] = m.crossattention_t2i
    params = dict(ref.named_parameters())
    cap = {}

    def hook_for(name):
        def fwd(_m, _i, out):
            t = out[0] if isinstance(out, tuple) else out
            cap.setdefault(name, []).append({"b": t.detach().clone()})
            rec = cap[name][-1]
            t.register_hook(lambda g: rec.__setitem__("gb", g.detach().clone()))
        return fwd
    for name, mod in gates.items():
        mod.register_forward_hook(hook_for(name))
    r16 = lambda t: t.to(torch.bfloat16).double()
    rows = {n: [] for n in gates}
    for s in range(a.batches):
        b = detgen.synth_batch(a.B, c["image_size"], c["max_text_len"], c["vocab_size"], seed=1 + s, min_len=min(8, c["max_text_len"] // 2))
        cap.clear()
        ref.zero_grad(set_to_none=True)
        ref.training_loss(b, b["itm_labels"]).backward()
        for n in gates:
            alpha = float(params[n])
            exact = both = gonly = bonly = sabs = 0.0
            for rec in cap.get(n, []):                   # the MLM pass and the ITM pass each visit the gate
                if "gb" not in rec:
                    continue
                br, d = rec["b"].double(), rec["gb"].double() / alpha          # dOut = d(loss) / d(alpha * branch)
                exact += float((d * br).sum()); sabs += float((d * br).abs().sum())
                both += float((r16(d.float()) * r16(br.float())).sum())
                gonly += float((r16(d.float()) * br).sum()); bonly += float((d * r16(br.float())).sum())
            ag = float(params[n].grad)
            rows[n].append({"autograd": ag, "exact_dot": exact, "cancel": abs(exact) / max(sabs, 1e-300), "both_bf16": both / exact - 1,
                            "grad_bf16": gonly / exact - 1, "branch_bf16": bonly / exact - 1})
    out = {"what": "relative deviation of d alpha = <dOut, branch> when the oracle's exact fp32 operands are rounded to bf16 before an exact dot product",
           "config": a.config, "B": a.B, "batches": a.batches, "gates": rows}
    worst = max(abs(r["both_bf16"]) for v in rows.values() for r in v)
    out["worst_both_bf16"] = worst
    for n, v in rows.items():
        print(n.split("encoder.")[-1][-44:].rjust(44), " cancel %.1e" % v[0]["cancel"], " both_bf16:", " ".join("%+.3f" % r["both_bf16"] for r in v),
              " | grad only:", " ".join("%+.3f" % r["grad_bf16"] for r in v))
    print("worst |both_bf16|: %.3f" % worst)
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(f"profiles/r05_gate_grad_study_{a.config}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
