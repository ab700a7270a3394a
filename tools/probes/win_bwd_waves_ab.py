"""A/B of the one-pass window-attention backward with 9 waves against 8 + 3 helper waves (FIBER_WIN_BWD_WAVES=11): same inputs in two child
processes (the switch is read once per process), outputs compared, us per call at the four Swin-B stages of 512 images 384^2.
    python tools/probes/win_bwd_waves_ab.py [images]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

STAGES = [("s0", 96, 128, 4), ("s1", 48, 256, 8), ("s2", 24, 512, 16), ("s3", 12, 1024, 32)]


def child(images, out_path, colsum):
    from fiber_amd import lib
    lib.load()
    dev, ws, res = "cuda", 12, {}
    for name, H, C, heads in STAGES:
        for shift in (0, 6):
            torch.manual_seed(1)
            rows, N = images * H * H, ws * ws
            qkv = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16)
            tab = torch.randn(529, heads, device=dev) * 0.5
            o = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
            lse = torch.empty(rows, heads, device=dev)
            lib.call("fiber_window_attn_fwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o), lib.ptr(lse), images, H, H, C, heads, ws, shift, 1)
            do = torch.randn(rows, C, device=dev).to(torch.bfloat16)
            dqkv = torch.empty_like(qkv)
            dtab = torch.empty_like(tab)
            delta = torch.zeros(rows, heads, device=dev)
            nz = lib.plain("fiber_window_attn_bwd_slices", rows // N, heads)
            part = torch.empty(nz * heads * N * N, device=dev)
            cs_rows = lib.plain("fiber_window_attn_colsum_rows", rows // N, heads, ws)
            csum = torch.zeros(3 * C, device=dev)
            cs_ws = torch.empty(cs_rows * 3 * C, device=dev)
            pc, pw = (lib.ptr(csum), lib.ptr(cs_ws)) if colsum else (None, None)

            def run():
                lib.call("fiber_window_attn_bwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o), lib.ptr(do), lib.ptr(lse), lib.ptr(dqkv),
                         lib.ptr(dtab), lib.ptr(delta), lib.ptr(part), pc, pw, images, H, H, C, heads, ws, shift, 1)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            sub = slice(0, 144 * 64)                     # the first 64 windows' rows of image 0 are enough for a comparison file
            res[f"{name}_{shift}"] = (us, dqkv.float().cpu() if rows <= 300000 else dqkv[:300000].float().cpu(), dtab.cpu(), csum.cpu())
            del qkv, o, do, dqkv, part
    torch.save(res, out_path)


def main():
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 3:
        return child(images, sys.argv[2], sys.argv[3] == "1")
    for colsum in ("0", "1"):
        out = {}
        for nwv in ("9", "11"):
            path = f"/tmp/win_ab_{nwv}.pt"
            env = dict(os.environ, FIBER_WIN_BWD_WAVES=nwv)
            subprocess.run([sys.executable, os.path.abspath(__file__), str(images), path, colsum], env=env, check=True, timeout=600)
            out[nwv] = torch.load(path)
        print(f"== colsum offer {colsum}: case | us 9 waves | us 8+3 waves | max |d dqkv| (rel to max) | max |d dtab| rel | max |d colsum| rel")
        for k in out["9"]:
            a, b = out["9"][k], out["11"][k]
            dq = (a[1] - b[1]).abs().max().item() / a[1].abs().max().item()
            dt = (a[2] - b[2]).abs().max().item() / a[2].abs().max().item()
            dc = (a[3] - b[3]).abs().max().item() / max(a[3].abs().max().item(), 1e-30)
            print(f"{k:6s} {a[0]:9.1f} {b[0]:9.1f}   {dq:.2e}  {dt:.2e}  {dc:.2e}")


if __name__ == "__main__":
    main()
