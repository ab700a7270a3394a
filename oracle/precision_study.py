"""Where does the bf16 path's loss gap come from?  TEST / ANALYSIS INFRASTRUCTURE ONLY (CPU, oracle only).

Runs oracle/fiber_ref.py (fp32) next to copies of itself in which chosen tensors are rounded to bf16 exactly where the HIP
path stores them in bf16, plus the oracle under torch.autocast(bfloat16) -- the analogue of the reference's own mixed-precision
training (`precision=16`, reference config.py:92; fp16 has no CPU GEMMs, bf16 autocast is the closest runnable form).
It answers two questions the round-2 review asked:
  * what does the REFERENCE's mixed precision do to the MLM+ITM loss curve on these batches (gap to its fp32 run), and
  * which rounding sites of the HIP data flow carry the gap (residual stream / LayerNorm outputs / GEMM outputs / GEMM operands).

Rounding sites (flags of `Sites`):
  gemm_in   activations entering a GEMM (they are bf16 tensors in HBM)          w        bf16 working copies of the weights
  gemm_out  GEMM results stored in bf16 (qkv, fc1 pre-activation, branch outputs)   ln_out   LayerNorm outputs stored in bf16
  resid     the residual stream itself: x + branch stored in bf16 (Swin: swin_transformer.py:388-391; RoBERTa: LN input
            a + h and dense + input, roberta.py:485,417-423)                       attn     softmax probabilities / attention outputs in bf16
  act_grad  activation gradients rounded to bf16 at the same sites (the cast's backward)

    python -m oracle.precision_study step0   [--config swin_t|tiny|swin_b] [--batches 6]
    python -m oracle.precision_study curve   [--config swin_t] [--steps 20]
Writes profiles/r03_precision_study_<what>_<config>.json.
"""
import argparse
import contextlib
import dataclasses
import json
import math
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases, detgen  # noqa: E402
from oracle import fiber_ref as R  # noqa: E402


@dataclasses.dataclass
class Sites:
    gemm_in: bool = False
    w: bool = False
    gemm_out: bool = False
    ln_out: bool = False
    resid: bool = False
    attn: bool = False
    act_grad: bool = True
    branch16: bool = False        # with an fp32 stream: the branch is rounded to bf16 BEFORE it is added (bf16 epilogue staging)


class _RoundSTE(torch.autograd.Function):
    """bf16 rounding forward, identity backward (activation gradients kept in fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


_S = Sites()
_state = {"on": False, "keep_out": 0}


def _r(x, flag):
    if not (_state["on"] and flag):
        return x
    return x.to(torch.bfloat16).to(torch.float32) if _S.act_grad else _RoundSTE.apply(x)


@contextlib.contextmanager
def _fp32_out():
    """The next linears feed a residual add whose epilogue works on the fp32 accumulator: no output rounding when the residual
    stream itself is fp32 (with a bf16 stream the SUM is rounded, by the `resid` site)."""
    if _S.branch16:                                          # the branch goes through a bf16 staging slab first
        yield
        return
    _state["keep_out"] += 1
    try:
        yield
    finally:
        _state["keep_out"] -= 1


def _linear_forward(self, x):
    w = _r(self.weight, _S.w)
    y = F.linear(_r(x, _S.gemm_in), w, self.bias)
    return y if _state["keep_out"] else _r(y, _S.gemm_out)


def _conv_forward(self, x):
    y = F.conv2d(_r(x, _S.gemm_in), _r(self.weight, _S.w), self.bias, self.stride, self.padding)
    return _r(y, _S.gemm_out)


def _ln_forward(self, x):
    return _r(F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps), _S.ln_out)


def _mlp_forward(self, x):
    with _fp32_out():                                        # GELU runs on the fp32 accumulator in the fc1 epilogue
        h = self.fc1(x)
    g = _r(F.gelu(h), _S.gemm_out)
    with _fp32_out():
        return self.fc2(g)


def _winattn_forward(self, xw, mask=None, y=None, y_mask=None):
    Bw, N, C = xw.shape
    h, d = self.heads, C // self.heads
    q, k, v = self.qkv(xw).view(Bw, N, 3, h, d).permute(2, 0, 3, 1, 4)
    a = (q * d ** -0.5) @ k.transpose(-1, -2)
    bias = self.relative_position_bias_table[self.relative_position_index.reshape(-1)]
    a = a + bias.view(N, N, h).permute(2, 0, 1)[None]
    if mask is not None:
        nW = mask.shape[0]
        a = (a.view(Bw // nW, nW, h, N, N) + mask[None, :, None]).view(Bw, h, N, N)
    o = _r((_r(a.softmax(-1), _S.attn) @ v).transpose(1, 2).reshape(Bw, N, C), _S.attn)
    if y is None:
        with _fp32_out():
            return self.proj(o)
    out = self.proj(o)
    B, S, _ = y.shape
    nW = Bw // B
    kt, vt = self.qkv_text_i2t(y).view(B, S, 2, h, d).permute(2, 0, 3, 1, 4)
    kt = kt.repeat_interleave(nW, 0)
    vt = vt.repeat_interleave(nW, 0)
    qi = self.qkv_i2t(self.norm_i2t_i(out)).view(Bw, N, h, d).transpose(1, 2) * d ** -0.5
    ai = qi @ kt.transpose(-1, -2)
    if y_mask is not None:
        ai = ai + y_mask.view(B, 1, 1, S).repeat_interleave(nW, 0)
    yi = _r((_r(ai.softmax(-1), _S.attn) @ vt).transpose(1, 2).reshape(Bw, N, C), _S.attn)
    return _r(out + self.alpha_i2t * self.proj_i2t(yi), _S.gemm_out)      # the branch (a bf16 tensor), not the stream


def _block_forward(self, x, y=None, y_mask=None):
    H, W = self.res
    B, L, C = x.shape
    u = self.norm1(x).view(B, H, W, C)
    if self.shift:
        u = torch.roll(u, (-self.shift, -self.shift), (1, 2))
    aw = self.attn(R.to_windows(u, self.ws), self.attn_mask, y, y_mask)
    u = R.from_windows(aw, self.ws, H, W)
    if self.shift:
        u = torch.roll(u, (self.shift, self.shift), (1, 2))
    x = _r(x + self.drop_path(u.reshape(B, L, C)), _S.resid)
    return _r(x + self.drop_path(self.mlp(self.norm2(x))), _S.resid)


def _rsa_forward(self, hq, hkv, mask):
    B, S, Hd = hq.shape
    d = Hd // self.heads
    q = self.query(hq).view(B, S, self.heads, d).transpose(1, 2)
    k = self.key(hkv).view(B, -1, self.heads, d).transpose(1, 2)
    v = self.value(hkv).view(B, -1, self.heads, d).transpose(1, 2)
    a = q @ k.transpose(-1, -2) / math.sqrt(d)
    if mask is not None:
        a = a + mask
    p = self.dropout(_r(a.softmax(-1), _S.attn))
    return _r((p @ v).transpose(1, 2).reshape(B, S, Hd), _S.attn)


def _rout_forward(self, x, resid, last_norm=True):
    with _fp32_out():
        y = self.dense(x)
    x = _r(self.dropout(y) + resid, _S.resid)
    return self.LayerNorm(x) if last_norm else x


def _rlayer_forward(self, h, mask=None, encoder_hidden_states=None, last_norm=True):
    if encoder_hidden_states is None:
        with _fp32_out():
            a = self.attention(h, h, mask)
    else:
        a = self.attention(h, h, mask)
        a = _r(self.alpha_t2i * self.crossattention_t2i(a, encoder_hidden_states, None) + a, _S.gemm_out)
    a = self.attention.output.LayerNorm(_r(a + h, _S.resid))
    return (self.output(self.intermediate(a), a, last_norm),)


def _inter_forward(self, x):
    with _fp32_out():
        h = self.dense(x)
    return _r(F.gelu(h), _S.gemm_out)


_PATCHES = [(nn.Linear, _linear_forward), (nn.Conv2d, _conv_forward), (nn.LayerNorm, _ln_forward), (R.Mlp, _mlp_forward),
            (R.WindowAttention, _winattn_forward), (R.SwinTransformerBlock, _block_forward), (R.RobertaSelfAttention, _rsa_forward),
            (R.RobertaOutput, _rout_forward), (R.RobertaLayer, _rlayer_forward), (R.RobertaIntermediate, _inter_forward)]


@contextlib.contextmanager
def emulate(sites):
    """Run oracle modules with bf16 rounding at `sites`."""
    global _S
    saved = [(cls, cls.forward) for cls, _ in _PATCHES]
    prev = _S
    _S = sites
    for cls, fn in _PATCHES:
        cls.forward = fn
    _state["on"] = True
    try:
        yield
    finally:
        _state["on"] = False
        _S = prev
        for cls, fn in saved:
            cls.forward = fn


MODES = {
    # what the round-2 HIP path stores in bf16: everything
    "hip_bf16_stream": Sites(gemm_in=True, w=True, gemm_out=True, ln_out=True, resid=True, attn=True),
    # the same with an fp32 residual stream (Swin x, RoBERTa LN inputs / outputs on the residual path)
    "hip_fp32_stream": Sites(gemm_in=True, w=True, gemm_out=True, ln_out=False, resid=False, attn=True),
    # fp32 stream, but every branch is rounded to bf16 before the add (what a bf16-staged GEMM epilogue does)
    "hip_fp32_stream_bf16_branch": Sites(gemm_in=True, w=True, gemm_out=True, ln_out=False, resid=False, attn=True, branch16=True),
    # fp32 stream AND fp32 activation gradients (only the forward sites are bf16)
    "hip_fp32_stream_fp32_grads": Sites(gemm_in=True, w=True, gemm_out=True, ln_out=False, resid=False, attn=True, act_grad=False),
    # floor of "bf16 MFMA compute": only the GEMM operands are bf16, every stored tensor fp32
    "gemm_operands_only": Sites(gemm_in=True, w=True, attn=True),
    # single sites on top of fp32
    "only_resid": Sites(resid=True),
    "only_weights": Sites(w=True),
}


def build(config_name):
    cfg = dict({"tiny": cases.TINY, "swin_t": cases.SWIN_T, "swin_b": cases.SWIN_B}[config_name])
    size = cfg["image_size"]
    vocab = cfg.get("vocab_size", 50265)
    S = cfg.get("max_text_len", 40)
    return cfg, size, vocab, S


def make_model(cfg):
    torch.manual_seed(0)
    m = detgen.fill_(R.FiberRef(cfg).train())
    for n, p in m.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)
    return m


def loss_of(model, batch, mode):
    if mode == "fp32":
        return model.training_loss(batch, batch["itm_labels"])
    if mode == "autocast_bf16":
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = model.training_loss(batch, batch["itm_labels"])
        return out.float()
    with emulate(MODES[mode]):
        return model.training_loss(batch, batch["itm_labels"])


def cmd_step0(args):
    cfg, size, vocab, S = build(args.config)
    model = make_model(cfg).eval()
    modes = ["autocast_bf16"] + list(MODES)
    rows = []
    for i in range(args.batches):
        b = detgen.synth_batch(args.B, size, S, vocab, seed=100 + i, min_len=min(8, S))
        with torch.no_grad():
            ref = loss_of(model, b, "fp32").item()
            row = {"batch": i, "fp32": ref}
            for m in modes:
                row[m] = loss_of(model, b, m).item() - ref
        rows.append(row)
        print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
    summary = {m: {"max_abs": max(abs(r[m]) for r in rows), "mean_abs": sum(abs(r[m]) for r in rows) / len(rows)} for m in modes}
    out = {"what": "forward-only loss gap to the fp32 oracle (MLM + ITM, dropout off)", "config": args.config, "B": args.B,
           "rows": rows, "summary": summary}
    print(json.dumps(summary, indent=1))
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(f"profiles/r03_precision_study_step0_{args.config}.json", "w"), indent=1)


def cmd_curve(args):
    from fiber_amd.optim import HFAdamW
    from fiber_amd.modules import fiber_utils
    cfg, size, vocab, S = build(args.config)
    modes = args.modes.split(",")
    nb, warm, steps = args.nb, args.warmup, args.steps
    batches = [detgen.synth_batch(args.B, size, S, vocab, seed=100 + i, min_len=min(8, S)) for i in range(nb)]
    curves = {}
    for mode in ["fp32"] + modes:
        model = make_model(cfg)
        head = [p for n, p in model.named_parameters() if any(k in n for k in ("mlm_score", "itm_score", "pooler", "cross_modal"))]
        ids = {id(p) for p in head}
        body = [p for p in model.parameters() if id(p) not in ids]
        # two groups are enough here (the 6-group split of set_schedule only changes lr multipliers / decay, identically in
        # every mode): backbone at lr, heads and cross-modal at 5 x lr
        opt = HFAdamW([{"params": body, "lr": args.lr, "weight_decay": 0.01}, {"params": head, "lr": 5 * args.lr, "weight_decay": 0.01}],
                      lr=args.lr, eps=1e-8, betas=(0.9, 0.98))
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s_: fiber_utils.poly_decay_lambda(s_, warm, steps, args.lr, 0, 1))
        t0 = time.time()
        ls = []
        for step in range(steps):
            b = batches[step % nb]
            opt.zero_grad(set_to_none=True)
            loss = loss_of(model, b, mode)
            loss.backward()
            opt.step()
            sched.step()
            ls.append(loss.item())
        curves[mode] = ls
        print(mode, f"{time.time() - t0:.0f}s", [round(v, 4) for v in ls], flush=True)
    summary = {}
    for mode in modes:
        gaps = sorted(abs(a - b) for a, b in zip(curves[mode], curves["fp32"]))
        summary[mode] = {"gap_max": gaps[-1], "gap_median": gaps[len(gaps) // 2], "gap_p90": gaps[int(0.9 * (len(gaps) - 1))],
                         "steps_within_1e-3": sum(g <= 1e-3 for g in gaps), "step0": abs(curves[mode][0] - curves["fp32"][0])}
    print(json.dumps(summary, indent=1))
    out = {"what": "MLM+ITM loss curves, HF AdamW, warm-up + poly decay; gap of each precision mode to the fp32 oracle run",
           "config": args.config, "B": args.B, "steps": steps, "batches_cycled": nb, "lr": args.lr, "curves": curves, "summary": summary}
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(f"profiles/r03_precision_study_curve_{args.config}.json", "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["step0", "curve"])
    ap.add_argument("--config", default="swin_t")
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--nb", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--modes", default="autocast_bf16,hip_bf16_stream,hip_fp32_stream,gemm_operands_only")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
    {"step0": cmd_step0, "curve": cmd_curve}[a.cmd](a)
