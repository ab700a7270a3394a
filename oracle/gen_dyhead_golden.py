"""Golden vectors for the DyHead tower of the fine-grained model (second half), produced by executing the
REFERENCE's own layers/dyhead.py (DyHead, DyConv, Conv3x3Norm) and layers/dyrelu.py (DYReLU, h_sigmoid) -- loaded by path,
CONTAINER-ONLY -- with ONE stand-in: `layers/deform_conv.py` imports the compiled CUDA extension (maskrcnn_benchmark._C, not
buildable here), so its `ModulatedDeformConv` is oracle/dcn_ref.py's restatement (pinned as that file's header says).  Everything
around the sampling -- the three-level mixing, stride-2 / up-sampled neighbours, GroupNorm, the scale-attention (AttnConv +
h_sigmoid), DYReLU -- is the reference's code.  Weights are drawn by parameter name (oracle/detgen.py); the offset convolution
gets a larger spread than the reference's init (std 0.01 would leave every offset ~0 and the test blind to the sampling).
Fixtures: inputs' names only (regenerated from the seed), outputs of every level, gradients of the inputs and the gradient
norm of every parameter for a fixed random projection.  Run: python -m oracle.gen_dyhead_golden"""
import os
import sys
import types

import numpy as np
import torch

from . import cases, dcn_ref, detgen, shim

LAYERS = os.path.join(shim.REF, "fine_grained", "maskrcnn_benchmark", "layers")
DY_CASES = {
    # channels, GroupNorm, DYReLU, scale attention, deformable, tower depth, batch, level sizes (fine -> coarse)
    "dyhead_dcn64": dict(C=64, gn=True, dyrelu=True, dyfuse=True, deform=True, convs=2, B=2, sizes=[(20, 28), (10, 14), (5, 7)]),
    "dyhead_plain64": dict(C=64, gn=True, dyrelu=False, dyfuse=True, deform=False, convs=1, B=1, sizes=[(12, 16), (6, 8), (3, 4)]),
}


class _Node:
    pass


def make_cfg(c):
    cfg = _Node()
    cfg.MODEL = _Node()
    cfg.MODEL.DYHEAD = _Node()
    d = cfg.MODEL.DYHEAD
    d.CHANNELS, d.USE_GN, d.USE_DYRELU, d.USE_DYFUSE, d.USE_DFCONV, d.NUM_CONVS = c["C"], c["gn"], c["dyrelu"], c["dyfuse"], c["deform"], c["convs"]
    return cfg


def load_reference():
    pkg = "_fiber_reference_layers"
    p = types.ModuleType(pkg)
    p.__path__ = [LAYERS]
    sys.modules[pkg] = p
    dc = types.ModuleType(pkg + ".deform_conv")
    dc.ModulatedDeformConv = dcn_ref.ModulatedDeformConv
    sys.modules[pkg + ".deform_conv"] = dc
    shim._load("dyrelu", os.path.join(LAYERS, "dyrelu.py"), pkg)
    return shim._load("dyhead", os.path.join(LAYERS, "dyhead.py"), pkg)


def set_weights(model, name):
    """deterministic values by parameter name; see the module docstring for the offset convolution's spread"""
    with torch.no_grad():
        for k, p in model.named_parameters():
            g = detgen._rng(f"dyhead:{name}:{k}", 0)
            if k.endswith("offset.weight"):
                std = 0.05
            elif k.endswith("offset.bias"):
                std = 0.3
            elif k.endswith(".bn.weight"):
                p.copy_(torch.from_numpy((1.0 + 0.1 * g.standard_normal(tuple(p.shape))).astype(np.float32)))
                continue
            elif k.endswith("bias"):
                std = 0.05
            else:
                std = 1.0 / (p[0].numel() ** 0.5) if p.dim() > 1 else 0.1
            p.copy_(torch.from_numpy((std * g.standard_normal(tuple(p.shape))).astype(np.float32)))
        for k, p in model.named_parameters():
            if k.endswith("conv.weight") or k.endswith("offset.weight"):
                p.copy_(p.to(torch.bfloat16).float())      # GEMM operands: bf16-exact, so both sides multiply the same numbers


def inputs(name):
    c = DY_CASES[name]
    return [cases.randn(f"{name}.x{i}", (c["B"], c["C"], h, w)).to(torch.bfloat16).float() for i, (h, w) in enumerate(c["sizes"])]


def projection(name, outs):
    tot = 0.0
    for i, o in enumerate(outs):
        tot = tot + (o * cases.randn(f"{name}.g{i}", tuple(o.shape))).sum() / o.numel() ** 0.5
    return tot


def main():
    dy = load_reference()
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for name, c in DY_CASES.items():
        torch.manual_seed(0)
        model = dy.DyHead(make_cfg(c), c["C"]).float()
        set_weights(model, name)
        xs = [x.requires_grad_() for x in inputs(name)]
        outs = model(xs)
        projection(name, outs).backward()
        rec = {"param_names": np.array([k for k, _ in model.named_parameters()])}
        for i, o in enumerate(outs):
            rec[f"out{i}"] = o.detach().numpy()
            rec[f"dx{i}"] = xs[i].grad.numpy()
        rec["grad_norms"] = np.array([float(p.grad.norm()) for _, p in model.named_parameters()], dtype=np.float64)
        for k, p in model.named_parameters():
            if k in ("dyhead_tower.0.DyConv.1.conv.weight", "dyhead_tower.0.offset.weight", "dyhead_tower.0.DyConv.1.conv.bias", "dyhead_tower.0.offset.bias"):
                rec["grad:" + k] = p.grad.numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, [tuple(o.shape) for o in outs], "params", len(rec["param_names"]), "out0 std", float(outs[0].std()))


if __name__ == "__main__":
    main()
