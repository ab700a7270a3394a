"""HBM read / write / copy bandwidth with plain torch kernels (GB/s): is a write-dominated kernel bound lower than a read one?"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
N = 2 * 1024 ** 3                      # 4 GiB of bf16
x = torch.empty(N, dtype=torch.bfloat16, device="cuda").normal_()
y = torch.empty_like(x)
gb = N * 2 / 1e9
print(f"write (fill_)        : {gb / t(lambda: y.fill_(1.0)):8.0f} GB/s")
print(f"read  (sum)          : {gb / t(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32)):8.0f} GB/s")
print(f"copy  (1R + 1W)      : {2 * gb / t(lambda: y.copy_(x)):8.0f} GB/s total")
z = torch.empty_like(x)
print(f"add   (2R + 1W)      : {3 * gb / t(lambda: torch.add(x, y, out=z)):8.0f} GB/s total")
h = N // 2
print(f"split (1R + 2W, chunk): {3 * (gb / 2) / t(lambda: (y[:h].copy_(x[:h]), z[:h].copy_(x[:h]))):8.0f} GB/s total (two launches)")
