"""A/B of the hand-written GEMM at the bench shapes (2*B*L rows): run once per build / env setting and diff the lines.
   python tools/gemm_ab.py [images=512]      env: FIBER_GEMM_PERSIST=0|1, FIBER_GEMM_TILE=..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops

BF = torch.bfloat16


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
    print(f"PERSIST={os.environ.get('FIBER_GEMM_PERSIST', '1')} TILE={os.environ.get('FIBER_GEMM_TILE', '-')}")
    shapes = [("s1.fc1", B * 2304, 1024, 256), ("s1.fc2", B * 2304, 256, 1024), ("s2.qkv", B * 576, 1536, 512),
              ("s2.proj", B * 576, 512, 512), ("s2.fc1", B * 576, 2048, 512), ("s2.fc2", B * 576, 512, 2048),
              ("s3.fc1", B * 144, 4096, 1024), ("s3.fc2", B * 144, 1024, 4096)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").to(BF)
        rs = torch.ones(M // 64, device="cuda")
        fl = 2.0 * M * N * K
        t = {
            "plain": timeit(lambda: ops.gemm_nt(x, w)),
            "bias": timeit(lambda: ops.gemm_nt(x, w, b)),
            "gelu+pre": timeit(lambda: ops.gemm_nt(x, w, b, None, 1, True)),
            "res+rs": timeit(lambda: ops.gemm_nt(x, w, b, r, 0, False, rs, 64)),
            "gelu'+cs": timeit(lambda: ops.gemm_nt(x, w, None, None, 2, False, None, 0, r, True)),
            "gelu'": timeit(lambda: ops.gemm_nt(x, w, None, None, 2, False, None, 0, r, False)),
            "lib": timeit(lambda: torch.nn.functional.linear(x, w)),
        }
        print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d} | " + " | ".join(f"{k} {v:7.0f}us {fl / v / 1e6:6.0f}TF" for k, v in t.items()), flush=True)


if __name__ == "__main__":
    main()
