"""Micro-benchmark of the hot-path kernels at FIBER-Base shapes (run on the GPU box): HIP GEMM vs library GEMM,
attention cores, LayerNorm.  Prints one line per case: us, TFLOP/s or GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def gemm_cases(B):
    out = []
    for name, L, C in (("s0", 9216, 128), ("s1", 2304, 256), ("s2", 576, 512), ("s3", 144, 1024)):
        M = B * L
        out += [(f"{name}.qkv", M, 3 * C, C, 0), (f"{name}.proj", M, C, C, 0), (f"{name}.fc1", M, 4 * C, C, 1), (f"{name}.fc2", M, C, 4 * C, 0)]
    M = B * 40
    out += [("txt.qkvo", M, 768, 768, 0), ("txt.fc1", M, 3072, 768, 1), ("txt.fc2", M, 768, 3072, 0), ("t2i.kv512", B * 576, 768, 512, 0)]
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    lib.load()
    if which in ("all", "gemm"):
        print(f"== GEMM (B={B}): hip fwd (epilogue) | lib fwd NT | lib dgrad NN | lib wgrad TN   [us / TFLOP/s]")
        for name, M, N, K, act in gemm_cases(B):
            x = torch.randn(M, K, device=dev).to(BF)
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
            bias = torch.randn(N, device=dev)
            dy = torch.randn(M, N, device=dev).to(BF)
            fl = 2.0 * M * N * K
            t_hip = timeit(lambda: ops.gemm_nt(x, w, bias, None, act, bool(act)))
            t_hip0 = timeit(lambda: ops.gemm_nt(x, w, None, None, 0, False))
            t_lib = timeit(lambda: torch.nn.functional.linear(x, w))
            t_dg = timeit(lambda: torch.matmul(dy, w))
            t_wg = timeit(lambda: torch.matmul(dy.t(), x))
            f = lambda t: f"{t:8.1f}us {fl / t / 1e6:7.1f}TF"
            print(f"{name:10s} M={M:7d} N={N:5d} K={K:5d} | hip+epi {f(t_hip)} | hip {f(t_hip0)} | lib {f(t_lib)} | dgrad {f(t_dg)} | wgrad {f(t_wg)}")
    if which in ("mlpbwd",):
        print("== fc2 dgrad + GELU' + bias-grad: fused hip GEMM epilogue vs lib GEMM + gelu_bwd + colsum  [us]")
        for name, L, C in (("s0", 9216, 128), ("s1", 2304, 256), ("s2", 576, 512), ("s3", 144, 1024)):
            M = B * L
            dy = torch.randn(M, C, device=dev).to(BF)
            w2 = (torch.randn(C, 4 * C, device=dev) * 0.02)
            h = torch.randn(M, 4 * C, device=dev).to(BF)
            wt, wb = ops.bf16_weight_t(w2), ops.bf16_weight(w2)
            t_f = timeit(lambda: ops.gemm_nt(dy, wt, None, None, 2, False, aux=h, want_colsum=True))
            t_p = timeit(lambda: ops.gemm_nt(dy, wt, None, None, 0, False))
            def unf():                                # the product path: library dgrad in NT form (ops._dgrad) + gelu' + column sums
                return ops.gelu_bwd_colsum(torch.nn.functional.linear(dy, wt), h)
            t_u = timeit(unf)
            t_l = timeit(lambda: torch.nn.functional.linear(dy, wt))
            print(f"{name} M={M} C={C}: fused {t_f:8.1f}us | hip plain {t_p:8.1f} | unfused total {t_u:8.1f} (lib gemm {t_l:8.1f})")
    if which in ("all", "attn"):
        print(f"== window attention (B={B})  fwd | bwd  [us, TFLOP/s on 4*L*N*C algorithmic flops fwd, x2.5 bwd]")
        for name, H, C, heads in (("s0", 96, 128, 4), ("s1", 48, 256, 8), ("s2", 24, 512, 16), ("s3", 12, 1024, 32)):
            L = H * H
            qkv = torch.randn(B, L, 3 * C, device=dev).to(BF).requires_grad_(True)
            tab = (torch.randn(529, heads, device=dev) * 0.5).requires_grad_(True)
            do = torch.randn(B, L, C, device=dev).to(BF)
            hm = os.environ.get("OPB_HEAD_MAJOR", "1") == "1"       # the layout the model uses (ops.linear_qkv_head_major)
            for shift in (0, 6 if H > 12 else 0):
                o = ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=hm)
                t_f = timeit(lambda: ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=hm))
                def bw():
                    o = ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=hm)
                    o.backward(do)
                t_fb = timeit(bw, reps=10)
                fl = 4.0 * B * L * 144 * C
                print(f"{name} shift={shift}: fwd {t_f:8.1f}us {fl / t_f / 1e6:6.1f}TF | fwd+bwd {t_fb:8.1f}us  (bwd ~{t_fb - t_f:8.1f}us)")
        print("== mha (text self / i2t / t2i)")
        for name, Lq, Lk, heads, D, masked in (("txt", 40, 40, 12, 64, True), ("i2t.s2", 576, 40, 16, 32, True), ("i2t.s3", 144, 40, 32, 32, True),
                                               ("t2i.s2", 40, 576, 12, 64, False), ("t2i.s3", 40, 144, 12, 64, False)):
            C = heads * D
            q = torch.randn(B * Lq, C, device=dev).to(BF).requires_grad_(True)
            k = torch.randn(B * Lk, C, device=dev).to(BF).requires_grad_(True)
            v = torch.randn(B * Lk, C, device=dev).to(BF).requires_grad_(True)
            km = torch.zeros(B, Lk, device=dev) if masked else None
            do = torch.randn(B * Lq, C, device=dev).to(BF)
            t_f = timeit(lambda: ops.mha(q, k, v, km, B, heads, D ** -0.5))
            def bw():
                ops.mha(q, k, v, km, B, heads, D ** -0.5).backward(do)
            t_fb = timeit(bw, reps=10)
            print(f"{name:7s}: fwd {t_f:8.1f}us | fwd+bwd {t_fb:8.1f}us")
    if which in ("all", "ln"):
        print("== layernorm fwd / fwd+bwd [us, GB/s]")
        for rows, C in ((B * 9216, 128), (B * 2304, 256), (B * 576, 512), (B * 144, 1024), (B * 40, 768)):
            x = torch.randn(rows, C, device=dev).to(BF).requires_grad_(True)
            g = torch.ones(C, device=dev, requires_grad=True)
            b = torch.zeros(C, device=dev, requires_grad=True)
            dy = torch.randn(rows, C, device=dev).to(BF)
            t_f = timeit(lambda: ops.layernorm(x, g, b))
            def bw():
                ops.layernorm(x, g, b).backward(dy)
            t_fb = timeit(bw, reps=10)
            by = rows * C * 2 * 2
            print(f"rows={rows:7d} C={C:5d}: fwd {t_f:7.1f}us {by / t_f / 1e3:7.1f}GB/s | fwd+bwd {t_fb:7.1f}us")


if __name__ == "__main__":
    main()
