"""GPU parity of the fp32 residual stream (config["residual_dtype"] = "fp32"; fiber_amd/ops.py "fp32 residual stream"): the
stream-aware kernels against plain PyTorch fp32 references of the same op, and the whole fused path against the oracle with the
tighter forward tolerance the mode buys (profiles/r03_precision_study_*: what each rounding site costs, measured on the oracle)."""
import pytest
import torch
import torch.nn.functional as F

from tests.hip_util import BF, DEV, assert_close, bf, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from fiber_amd import lib, ops
    lib.load()
    return ops


@pytest.fixture()
def fp32_stream(ops):
    ops.set_residual_dtype("fp32")
    yield
    ops.set_residual_dtype("bf16")


def rnd(*shape, std=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * std


# shapes that reach every kernel the residual epilogue lives in: register-staged fallback (K % 64 != 0), 64x64 / 128x128 /
# 256x128 LDS-DMA tiles, the wide kernel (>= 200 tiles), the persistent kernels v4 (with DropPath scale) and v3 (without)
@pytest.mark.parametrize("M,N,K", [(389, 136, 96), (300, 128, 64), (9216, 128, 512), (40000, 384, 128), (70001, 256, 1024),
                                   (140000, 512, 512), (33000 * 4, 1024, 256)])
@pytest.mark.parametrize("droppath", [False, True])
def test_gemm_fp32_residual_epilogue(ops, M, N, K, droppath):
    """Y32 = rowscale * (X W^T + b) + R32 with the branch rounded to bf16 once and the SUM kept in fp32; Y16 = bf16(Y32)."""
    x, wb, b = bf(rnd(M, K)), bf(rnd(N, K, std=K ** -0.5)), rnd(N, seed=1).to(DEV)
    r32 = (rnd(M, N, seed=2) * 3.0).to(DEV)                      # NOT bf16-representable: the payload really is fp32
    rps = 1000 if droppath else 0
    nsamp = -(-M // 1000)
    rs = (torch.rand(nsamp, generator=torch.Generator().manual_seed(5)) < 0.8).float().div(0.8).to(DEV) if droppath else None
    if droppath and M % 1000:
        pytest.skip("per-sample scale needs whole samples")
    y16, y32 = ops.gemm_nt(x, wb, b, None, 0, False, rs, rps, res32=r32)
    branch = x.float() @ wb.float().t() + b
    if droppath:
        branch = branch * rs.repeat_interleave(1000)[:, None]
    want = branch.to(BF).float() + r32
    # the only rounding is the bf16 branch: error relative to the BRANCH, far below bf16 resolution of the sum
    err = (y32 - want).abs().max().item()
    assert err <= 2.0 ** -7 * branch.abs().max().item(), err
    assert rel_l2(y32, branch + r32) < 2.5e-3
    assert torch.equal(y16, y32.to(BF)), "the shadow must be the bf16 rounding of the fp32 sum"
    y16b, y32b = ops.gemm_nt(x, wb, b, None, 0, False, rs, rps, res32=r32)
    assert torch.equal(y32, y32b) and torch.equal(y16, y16b)      # race screen


@pytest.mark.parametrize("rows,C", [(1000, 128), (513, 96), (1280, 768), (300, 2048), (9216, 256)])
def test_layernorm_on_fp32_stream(ops, rows, C):
    x32 = (rnd(rows, C) * 2.0 + 0.3).to(DEV)
    g, b = (1.0 + rnd(C, std=0.1, seed=1)).to(DEV).requires_grad_(True), rnd(C, std=0.1, seed=2).to(DEV).requires_grad_(True)
    x16 = x32.to(BF).requires_grad_(True)
    xs = ops.with_f32(x16, x32)
    y, res = ops.layernorm_res(xs, g, b, 1e-5)
    assert ops.f32_of(res) is x32
    xr = x32.clone().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-5)
    assert_close("y", y, yr, 3e-3)
    # the statistics come from the fp32 payload, not from its bf16 shadow
    y_shadow, _ = ops.layernorm_res(x16.detach(), g, b, 1e-5)
    assert rel_l2(y, yr) <= rel_l2(y_shadow, yr) + 1e-6
    dy, dres = bf(rnd(rows, C, seed=3)), bf(rnd(rows, C, seed=4))
    torch.autograd.backward([y, res], [dy, dres])
    yr.backward(dy.float())
    assert_close("dx", x16.grad, xr.grad + dres.float(), 6e-3)
    assert_close("dgamma", g.grad, gr.grad, 6e-3)
    assert_close("dbeta", b.grad, br.grad, 6e-3)
    # plain LayerNorm with an fp32 copy of the output (post-LN text stack)
    z = ops.layernorm(ops.with_f32(x32.to(BF), x32), g.detach(), b.detach(), 1e-5, want_f32=True)
    z32 = ops.f32_of(z)
    assert z32 is not None and torch.equal(z, z32.to(BF))
    assert rel_l2(z32, yr) < 1e-5


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (1, 24, 24, 512)])
def test_patch_merge_ln_on_fp32_stream(ops, B, H, W, C):
    x32 = (rnd(B, H * W, C) * 1.5).to(DEV)
    g, b = (1.0 + rnd(4 * C, std=0.1, seed=1)).to(DEV).requires_grad_(True), rnd(4 * C, std=0.1, seed=2).to(DEV).requires_grad_(True)
    x16 = x32.to(BF).requires_grad_(True)
    y = ops.patch_merge_ln(ops.with_f32(x16, x32), g, b, H, W, 1e-5)
    xr = x32.clone().requires_grad_(True)
    v = xr.view(B, H, W, C)
    z = torch.cat([v[:, 0::2, 0::2], v[:, 1::2, 0::2], v[:, 0::2, 1::2], v[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * C)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = F.layer_norm(z, (4 * C,), gr, br, 1e-5)
    assert_close("y", y, yr, 3e-3)
    dy = bf(rnd(B, H * W // 4, 4 * C, seed=3))
    y.backward(dy)
    yr.backward(dy.float())
    assert_close("dx", x16.grad, xr.grad, 6e-3)
    assert_close("dgamma", g.grad, gr.grad, 6e-3)


@pytest.mark.parametrize("res_kind", ["none", "bf16", "fp32"])
@pytest.mark.parametrize("with_b,rowscale,p_a,p_b", [(False, False, 0.0, 0.0), (True, True, 0.0, 0.0), (False, False, 0.1, 0.0),
                                                     (True, False, 0.0, 0.1), (True, True, 0.2, 0.3)])
def test_stream_add(ops, res_kind, with_b, rowscale, p_a, p_b):
    """out = res + rowscale * (drop(a) + alpha * drop(b)): values against torch where no dropout is active, the exact adjoint
    identity <out, dy> = <a, da> + <b, db> (+ residual term) where it is, dalpha = <rowscale * dy, drop(b)>."""
    Bn, L, C = 6, 200, 128
    ops.manual_seed(3)
    ops.set_rng_step(0)
    a = bf(rnd(Bn, L, C)).requires_grad_(True)
    b = bf(rnd(Bn, L, C, seed=1)).requires_grad_(True) if with_b else None
    alpha = torch.tensor([0.37], device=DEV, requires_grad=True) if with_b else None
    rs = torch.tensor([1.25, 0.0, 1.25, 1.25, 0.0, 1.25], device=DEV) if rowscale else None
    r32 = (rnd(Bn, L, C, seed=2) * 2).to(DEV)
    if res_kind == "fp32":
        res = ops.with_f32(r32.to(BF).requires_grad_(True), r32)
    elif res_kind == "bf16":
        res = r32.to(BF).requires_grad_(True)
    else:
        res = None
    out = ops.stream_add(res, a, b=b, alpha=alpha, rowscale=rs, p_a=p_a, p_b=p_b, training=True)
    o32 = ops.f32_of(out)
    assert (o32 is not None) == (res_kind == "fp32")
    val = o32 if o32 is not None else out.float()
    if p_a == 0.0 and p_b == 0.0:
        want = a.detach().float() + (alpha.detach() * b.detach().float() if with_b else 0.0)
        if rowscale:
            want = want * rs[:, None, None]
        if res is not None:
            want = want + (r32 if res_kind == "fp32" else r32.to(BF).float())
        tol = 1e-6 if o32 is not None else 4e-3
        assert rel_l2(val, want) <= tol
    if o32 is not None:
        assert torch.equal(out, o32.to(BF))
    dy = bf(rnd(Bn, L, C, seed=4))
    out.backward(dy)
    # adjoint identity in fp32 on the branch part (the map (a, b) -> out - res is linear for fixed masks)
    branch = val - (r32 if res_kind == "fp32" else (r32.to(BF).float() if res_kind == "bf16" else 0.0))
    lhs = (branch * dy.float()).sum().item()
    rhs = (a.detach().float() * a.grad.float()).sum().item()
    if with_b:
        rhs += (b.detach().float() * b.grad.float()).sum().item()
    # both sides are sums of N products of bf16-rounded factors (out / da / db are stored in bf16 unless the fp32 payload is
    # compared): independent 2^-9 relative errors on N terms -> 3 sigma of the sum
    n = branch.numel()
    bound = 3 * 2.0 ** -9 * branch.norm().item() * dy.float().norm().item() / n ** 0.5
    assert abs(lhs - rhs) < bound, (lhs, rhs, bound)
    if res is not None:
        assert torch.equal(res.grad, dy)
    if with_b:
        # dalpha = <rowscale * dy, drop(b)>: the map is linear in alpha, so out(alpha + 1) - out(alpha) = rowscale * drop(b) when the
        # same keys are drawn again (re-seeding restarts the key counter)
        ops.manual_seed(3)
        ops.set_rng_step(0)
        res2 = None if res is None else (ops.with_f32(res.detach(), r32) if res_kind == "fp32" else res.detach())
        out2 = ops.stream_add(res2, a.detach(), b=b.detach(), alpha=alpha.detach() + 1.0, rowscale=rs, p_a=p_a, p_b=p_b, training=True)
        v2 = ops.f32_of(out2) if o32 is not None else out2.float()
        want_dalpha = ((v2 - val) * dy.float()).sum().item()
        tol = (2e-2 if o32 is None else 5e-3) * max(1.0, abs(want_dalpha)) + (0.5 if o32 is None else 0.0)   # bf16 outputs: the difference of two rounded tensors
        assert abs(alpha.grad.item() - want_dalpha) <= tol, (alpha.grad.item(), want_dalpha)


_ORACLE_LOSS = {}            # (case, batch seed) -> the fp32 oracle's loss: the same for both residual dtypes, computed once


def _run_path(case, residual_dtype, n_batches=8):
    from fiber_amd import ops as O
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils, objectives
    from oracle import cases, detgen
    from oracle import fiber_ref as R
    from tests.hip_util import load_from_oracle
    cfg = dict(cases.PATH_CASES[case]["config"])
    B = cases.PATH_CASES[case]["B"]
    ref = detgen.fill_(R.FiberRef(cfg).eval())
    model = FIBERTransformerSS(make_config(**cfg, residual_dtype=residual_dtype)).eval()
    load_from_oracle(model, ref)
    model.to(DEV)
    c = ref.config
    losses = []
    for seed in range(100, 100 + n_batches):
        b = detgen.synth_batch(B, c["image_size"], c["max_text_len"], c["vocab_size"], seed=seed, min_len=min(8, c["max_text_len"]))
        bd = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else [t.to(DEV) for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else v)
              for k, v in b.items()}
        fiber_utils.set_task(model)
        with torch.no_grad():
            got = (objectives.compute_mlm(model, bd)["mlm_loss"] + objectives.compute_itm(model, bd, itm_labels=b["itm_labels"])["itm_loss"]).item()
            if (case, seed) not in _ORACLE_LOSS:
                _ORACLE_LOSS[(case, seed)] = ref.training_loss(b, b["itm_labels"]).item()
            want = _ORACLE_LOSS[(case, seed)]
        losses.append((got, want))
    O.set_residual_dtype("bf16")
    return losses


@pytest.mark.parametrize("case", ["path_tiny", "path_swin_t"])
def test_fused_path_forward_gap_fp32_stream(case):
    """MLM + ITM loss of the whole fused path against the fp32 oracle on three batches, bf16 stream vs fp32 stream.  The oracle-side
    study (profiles/r03_precision_study_step0_*.json) puts the fp32-stream forward gap at <= 1.7e-3 (Swin-T, max of 6 batches) with
    a floor of 1.3e-3 for bf16 GEMM operands alone and 1.7e-3 for the reference-style autocast run; the bf16 stream sits at 4.8e-3."""
    l32 = _run_path(case, "fp32")
    l16 = _run_path(case, "bf16")
    g32 = [abs(a - b) for a, b in l32]
    g16 = [abs(a - b) for a, b in l16]
    print(f"{case}: forward gap fp32 stream {['%.2e' % g for g in g32]}  bf16 stream {['%.2e' % g for g in g16]}")
    # measured (8 batches): path_tiny fp32 stream mean 2.9e-4 / max 5.9e-4, bf16 stream 8.7e-4 / 4.1e-3; path_swin_t fp32 stream
    # 8.8e-4 / 1.8e-3, bf16 stream 1.09e-3 / 1.9e-3 -- at Swin-T width the bf16 GEMM operands (study floor 6e-4 / 1.3e-3) dominate
    assert max(g32) < 3e-3 and sum(g32) / len(g32) < 1.5e-3, (g32, l32)
    assert max(g16) < 6e-3 and sum(g16) / len(g16) < 2.5e-3, (g16, l16)
    assert sum(g32) <= sum(g16) + 2e-3, "the fp32 stream must not be further from the oracle than the bf16 stream"


def test_weight_gradients_on_their_own_stream_are_the_same_gradients(monkeypatch):
    """ops.enable_wgrad_stream(model): the TN weight-gradient GEMMs (and whatever the callers derive from their results) run on a second
    stream, autograd accumulates every gradient of the model on that stream, and backward() joins it.  Same kernels, same inputs: every parameter gradient must be BITWISE the
    one of the single-stream run (a consumer that read a gradient before the side stream had written it would show up here), three
    times in a row, on a path that has every kind of caller (head-major qkv permutation, packed projections, patch embedding,
    the fused LayerNorm-Mlp unfolding at C = 128)."""
    from fiber_amd import ops
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils
    from oracle import cases, detgen
    cfg = dict(cases.SWIN_B)
    torch.manual_seed(0)
    model = FIBERTransformerSS(make_config(**cfg)).eval()           # eval: no dropout / DropPath draws, both runs compute the same function
    for n, p in model.named_parameters():
        if "alpha_" in n:
            p.data.fill_(0.5)
    model.to("cuda")
    fiber_utils.set_task(model)
    b = detgen.synth_batch(8, 384, 40, 50265, seed=5, min_len=8)
    bd = {k: (v.to("cuda") if isinstance(v, torch.Tensor) else [t.to("cuda") for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else v)
          for k, v in b.items()}
    bd["itm_labels_override"] = bd["itm_labels"]

    monkeypatch.setenv("FIBER_WGRAD_STREAM", "1")            # (the switch is an experiment: off unless the variable is set)
    ops.enable_wgrad_stream(model)                           # accumulator nodes on the weight-gradient stream (before the first forward)

    def grads(on):
        ops.set_wgrad_stream(on)
        try:
            model.zero_grad(set_to_none=True)
            out = model(bd)
            sum(v for k, v in out.items() if "loss" in k).backward()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            ops.set_wgrad_stream(False)
    ref = grads(False)
    for rep in range(3):
        got = grads(True)
        assert got.keys() == ref.keys()
        bad = [n for n in ref if not torch.equal(ref[n], got[n])]
        # (fp32 atomics: embedding rows and the gates are summed in an order that differs from run to run -- compare those by value)
        really = [n for n in bad if (ref[n].float() - got[n].float()).abs().max() > 1e-3 * (ref[n].float().abs().max() + 1e-6)]
        assert not really, (rep, really[:8])


def test_deferred_folds_give_the_same_gradients_bitwise(monkeypatch):
    """ops.set_fold_defer(True): the slabs of the split weight-gradient GEMMs are folded by ONE multi-tensor launch at the end of
    backward() instead of one small launch per GEMM -- the same per-element summation order, so every gradient must be bitwise the one
    of the immediate folds (fp32-atomic gradients -- embedding rows, gates -- compared by value), twice in a row, optimizer step included
    (the second run starts from weights the first run's deferred gradients produced)."""
    from fiber_amd import ops, parallel
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS, fiber_utils
    from oracle import cases, detgen
    monkeypatch.setenv("FIBER_TN_FOLD_DEFER", "1")           # (the switch is an experiment: off unless the variable is set)
    b = detgen.synth_batch(4, 224, 40, 50265, seed=6, min_len=8)
    bd = {k: (v.to("cuda") if isinstance(v, torch.Tensor) else [t.to("cuda") for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else v)
          for k, v in b.items()}
    bd["itm_labels_override"] = bd["itm_labels"]

    def run(defer):
        torch.manual_seed(0)
        model = FIBERTransformerSS(make_config(**dict(cases.SWIN_T), learning_rate=1e-4, warmup_steps=0, max_steps=100)).eval()
        for n, p in model.named_parameters():
            if "alpha_" in n:
                p.data.fill_(0.5)
        model.to("cuda")
        fiber_utils.set_task(model)
        parallel.freeze_unused(model, model.unused_parameter_names())
        (opt,), _ = model.configure_optimizers()
        ops.set_fold_defer(defer)
        try:
            out = []
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                o = model(bd)
                sum(v for k, v in o.items() if "loss" in k).backward()
                torch.cuda.synchronize()
                out.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
                opt.step()
            return out
        finally:
            ops.set_fold_defer(False)
    ref, got = run(False), run(True)
    for step in range(2):
        assert ref[step].keys() == got[step].keys()
        bad = [n for n in ref[step] if not torch.equal(ref[step][n], got[step][n])]
        really = [n for n in bad if (ref[step][n].float() - got[step][n].float()).abs().max() > 1e-3 * (ref[step][n].float().abs().max() + 1e-6)]
        assert not really, (step, really[:8])
