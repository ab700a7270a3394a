"""A/B of a window-attention forward switch read once per process (any switch of the forward; the round-6 FIBER_WIN_FWD_S0F experiment this was written for is removed): us per call at the four Swin-B stages of
512 images 384^2, outputs compared.    python tools/probes/win_fwd_ab.py [images] [ENV_NAME] [value_a] [value_b]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

STAGES = [("s0", 96, 128, 4), ("s1", 48, 256, 8), ("s2", 24, 512, 16), ("s3", 12, 1024, 32)]


def child(images, out_path):
    from fiber_amd import lib
    lib.load()
    dev, ws, res = "cuda", 12, {}
    for name, H, C, heads in STAGES:
        for shift in (0, 6):
            torch.manual_seed(1)
            rows = images * H * H
            qkv = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16)
            tab = torch.randn(529, heads, device=dev) * 0.5
            o = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
            lse = torch.empty(rows, heads, device=dev)

            def run():
                lib.call("fiber_window_attn_fwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o), lib.ptr(lse), images, H, H, C, heads, ws, shift, 1)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}_{shift}"] = (e0.elapsed_time(e1) * 50, o[:300000].float().cpu(), lse[:300000].cpu())
            del qkv, o, lse
    torch.save(res, out_path)


def main():
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 2 and sys.argv[2] == "--child":
        return child(images, sys.argv[3])
    name = sys.argv[2] if len(sys.argv) > 2 else "FIBER_WIN_FWD_S0F"
    va, vb = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("0", "1")
    out = {}
    for rep in range(2):
        for v in (va, vb):
            path = f"/tmp/win_fwd_ab_{v}.pt"
            subprocess.run([sys.executable, os.path.abspath(__file__), str(images), "--child", path], env=dict(os.environ, **{name: v}), check=True, timeout=600)
            out.setdefault(v, []).append(torch.load(path))
    print(f"== {name}: case | us {va} (two runs) | us {vb} (two runs) | max |d o| | max |d lse|")
    for k in out[va][0]:
        a, b = out[va][0][k], out[vb][0][k]
        print(f"{k:6s} {a[0]:8.1f} {out[va][1][k][0]:8.1f}   {b[0]:8.1f} {out[vb][1][k][0]:8.1f}   {(a[1] - b[1]).abs().max().item():.2e}  {(a[2] - b[2]).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
