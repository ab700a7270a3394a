"""LayerNorm forward / backward bandwidth at the bench shapes, for A/B builds of csrc/norm.hip alone:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DLN_UF=1 -DLN_UB=1 fiber_amd/csrc/norm.hip -o tools/bin/ln_u1.so
    python tools/ln_bench.py tools/bin/ln_u1.so fiber_amd/libfiber_hip.so
"""
import ctypes as C
import sys
import torch

P, I, F = C.c_void_p, C.c_int, C.c_float
SHAPES = [(256 * 9216, 128), (256 * 2304, 256), (256 * 576, 512), (256 * 144, 1024), (256 * 40, 768)]


def bench(path):
    lib = C.CDLL(path)
    lib.fiber_layernorm_fwd_bf16.argtypes = [P, P, P, P, P, P, I, I, F, P]
    lib.fiber_layernorm_bwd_bf16.argtypes = [P] * 10 + [I, I, P]
    lib.fiber_layernorm_bwd_grid.argtypes = [I]
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for rows, Cc in SHAPES:
        x = torch.randn(rows, Cc, device="cuda", dtype=torch.bfloat16)
        dy = torch.randn_like(x)
        dres = torch.randn_like(x)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
        ws = torch.empty(lib.fiber_layernorm_bwd_grid(rows) * 8 * Cc, device="cuda")
        p = lambda t: t.data_ptr()
        fwd = lambda: lib.fiber_layernorm_fwd_bf16(p(x), p(g), p(b), p(y), p(mean), p(rstd), rows, Cc, 1e-5, st)
        bwd = lambda r: lib.fiber_layernorm_bwd_bf16(p(dy), p(x), p(g), p(mean), p(rstd), p(dres) if r else None, p(dx), p(dg), p(db), p(ws), rows, Cc, st)
        res = []
        for fn, nb in ((fwd, 2), (lambda: bwd(False), 3), (lambda: bwd(True), 4)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.append((us, nb * rows * Cc * 2 / us / 1e6))
        out.append(res)
        ref = torch.nn.functional.layer_norm(x.float(), (Cc,), g, b, 1e-5)
        err = float((y.float() - ref).abs().max())
        print(f"  rows {rows:8d} C {Cc:5d}: fwd {res[0][0]:7.1f} us {res[0][1]:5.2f} TB/s | bwd {res[1][0]:7.1f} us {res[1][1]:5.2f} TB/s | bwd+dres {res[2][0]:7.1f} us {res[2][1]:5.2f} TB/s | fwd max err {err:.3g}")
    return out


for path in sys.argv[1:]:
    print(path)
    bench(path)
    bench(path) if len(sys.argv) == 2 else None
