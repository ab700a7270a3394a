"""DyHead / deformable-convolution timings at the detection model's FPN geometry (800 x 1344 input, 256 channels, five levels).
   python tools/dcn_bench.py [batch=4]
Per kernel: gather (cols write + one read of the map = the algorithmic HBM bytes), scatter (cols read + map read + fp32 map
read-modify-write), the MFMA GEMM between them; then one DyConv layer and the 6-layer tower forward + backward."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from fiber_amd.modules import dyhead as DH

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


class _N:
    pass


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lib.load()
    C = 256
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    print(f"batch {B}, C {C}, levels {sizes}")
    for (H, W) in sizes[:3]:
        M = B * H * W
        x = torch.randn(B, H, W, C, device="cuda").to(BF)
        off = torch.randn(M, 18, device="cuda") * float(os.environ.get("DCN_OFFSET_STD", "1.0"))
        msk = torch.rand(M, 9, device="cuda")
        cols = torch.empty(M, 9 * C, device="cuda", dtype=BF)
        dx = torch.zeros(B, H, W, C, device="cuda")
        doff, dm = torch.empty_like(off), torch.empty_like(msk)
        w = (torch.randn(C, 9 * C, device="cuda") * (9 * C) ** -0.5).to(BF)
        tg = timeit(lambda: lib.call("fiber_dcn_gather_bf16", lib.ptr(x), lib.ptr(off), lib.ptr(msk), lib.ptr(cols), B, H, W, C, H, W, 3, 3, 1, 1))
        ts = timeit(lambda: lib.call("fiber_dcn_scatter_bf16", lib.ptr(cols), lib.ptr(x), lib.ptr(off), lib.ptr(msk), lib.ptr(dx), lib.ptr(doff), lib.ptr(dm),
                                     B, H, W, C, H, W, 3, 3, 1, 1))
        tm = timeit(lambda: ops.gemm_nt(cols, w))
        tso = timeit(lambda: lib.call("fiber_dcn_scatter_bf16", lib.ptr(cols), lib.ptr(x), lib.ptr(off), lib.ptr(msk), None, lib.ptr(doff), lib.ptr(dm),
                                      B, H, W, C, H, W, 3, 3, 1, 1))
        n = lib.plain("fiber_dcn_dx_workspace", B, H, W, C, H, W, 1)
        ws = torch.zeros(n, device="cuda")  # (the far map stays zero: offsets within the halo)
        dxb = torch.empty(B, H, W, C, device="cuda", dtype=BF)
        ttd = timeit(lambda: lib.call("fiber_dcn_dx_bf16", lib.ptr(cols), lib.ptr(off), lib.ptr(msk), lib.ptr(dxb), lib.ptr(ws), B, H, W, C, H, W, 3, 3, 1, 1))
        print(f"level {H}x{W}: offset/mask gradients only {tso:7.1f} us | tiled input gradient (LDS windows + sum) {ttd:7.1f} us, workspace {n * 4 / 2 ** 20:.0f} MiB")
        gb_g = (cols.numel() * 2 + x.numel() * 2 + off.numel() * 4 + msk.numel() * 4) / 1e9
        gb_s = (cols.numel() * 2 + x.numel() * 2 + dx.numel() * 8 + 2 * (off.numel() + msk.numel()) * 4) / 1e9
        print(f"level {H}x{W} M={M}: gather {tg:7.1f} us {gb_g / tg * 1e6:6.0f} GB/s | scatter {ts:7.1f} us {gb_s / ts * 1e6:6.0f} GB/s "
              f"({M * 9 * C * 4 / ts / 1e3:5.1f} G atomics/s) | GEMM {tm:7.1f} us {2.0 * M * C * 9 * C / tm / 1e6:5.0f} TFLOP/s")
    cfg = _N(); cfg.MODEL = _N(); cfg.MODEL.DYHEAD = _N()
    d = cfg.MODEL.DYHEAD
    d.CHANNELS, d.USE_GN, d.USE_DYRELU, d.USE_DYFUSE, d.USE_DFCONV = C, True, True, True, True
    for convs in (1, 6):
        d.NUM_CONVS = convs
        model = DH.DyHead(cfg, C).cuda()
        xs = [torch.randn(B, C, h, w, device="cuda", requires_grad=True) for h, w in sizes]

        def step():
            outs = model(xs)
            sum(o.float().square().mean() for o in outs).backward()
        t = timeit(step, reps=5, warm=2)
        flops = sum(2.0 * B * h * w * 9 * C * (C * 3 + 27) for h, w in sizes) * convs * 3
        print(f"DyHead tower, {convs} DyConv layer(s), forward + backward: {t / 1e3:7.2f} ms  ({flops / t / 1e6:5.0f} TFLOP/s of convolution work, "
              f"peak mem {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB)")


if __name__ == "__main__":
    main()
