"""LayerNorm + Mlp + residual at the Swin stage-0 / stage-1 shapes (run on the GPU box): the fused kernels of csrc/mlp_rows.hip against
the separate-kernel path (layernorm_res + mlp), forward alone and forward + backward, with and without DropPath.
    python tools/lnmlp_bench.py [images=512]
Prints us per call and the rate of the ALGORITHMIC bytes of the fused form (forward: x in, y out; backward: x, dy in; dx, dh, g, xhat out)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops

dev, BF = "cuda", torch.bfloat16


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    lib.load()
    for name, L, C in (("s0", 9216, 128), ("s1", 2304, 256)):
        H = 4 * C
        M = B * L
        x = torch.randn(B, L, C, device=dev).to(BF).requires_grad_(True)
        g = torch.ones(C, device=dev, requires_grad=True)
        b = torch.zeros(C, device=dev, requires_grad=True)
        w1 = (torch.randn(H, C, device=dev) * C ** -0.5).requires_grad_(True)
        b1 = torch.zeros(H, device=dev, requires_grad=True)
        w2 = (torch.randn(C, H, device=dev) * H ** -0.5).requires_grad_(True)
        b2 = torch.zeros(C, device=dev, requires_grad=True)
        dy = torch.randn(B, L, C, device=dev).to(BF)
        for dp in (False, True):
            rs = (torch.rand(B, device=dev) > 0.05).float() * (1 / 0.95) if dp else None
            rv = 1 / 0.95 if dp else None

            def fused(bwd):
                y = ops._LnMlp.apply(x, g, b, 1e-5, w1, b1, w2, b2, rs, rv)       # (the kernels, whatever widths ops.ln_mlp routes to them)
                if bwd:
                    y.backward(dy)

            def separate(bwd):
                v, r = ops.layernorm_res(x, g, b, 1e-5)
                y = ops.mlp(v, w1, b1, w2, b2, residual=r, rowscale=rs, rowscale_value=rv)
                if bwd:
                    y.backward(dy)
            with torch.no_grad():
                tf, ts = timeit(lambda: fused(False)), timeit(lambda: separate(False))
            tfb, tsb = timeit(lambda: fused(True)), timeit(lambda: separate(True))
            fb = 2 * M * C * 2                                     # forward bytes of the fused form
            bb = (3 * M * C + M * C + 2 * M * H) * 2               # backward: x, dy in; dx, xhat, dh, g out
            fl = 2.0 * M * C * H * 2
            print(f"{name} C={C} M={M} droppath={int(dp)} | fwd fused {tf:8.1f} us ({fb / tf / 1e6:5.2f} TB/s, {fl / tf / 1e6:6.0f} TF/s) separate {ts:8.1f} us"
                  f" | fwd+bwd fused {tfb:8.1f} us separate {tsb:8.1f} us | bwd fused {tfb - tf:8.1f} us ({bb / max(tfb - tf, 1) / 1e6:5.2f} TB/s incl. wgrad) separate {tsb - ts:8.1f} us",
                  flush=True)


if __name__ == "__main__":
    main()
