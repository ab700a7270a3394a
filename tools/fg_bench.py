"""Fine-grained fused backbone (fiber_amd/modules/fusion_swin.py; BASELINE.json configs[4], backbone half) at detection
resolution: forward + backward + AdamW of FusionSwinTransformer on synthetic 800x1344 images (800x1333 padded to the 32-multiple
the detection pipeline uses) with 256 text tokens.   python tools/fg_bench.py [batch=2] [H=800] [W=1344] [S=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from fiber_amd.modules import fusion_swin as FS
from fiber_amd.optim import FiberAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 800
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1344
S = int(sys.argv[4]) if len(sys.argv) > 4 else 256
lib.load()
torch.manual_seed(0)
ops.manual_seed(0)
m = FS.FusionSwinTransformer(drop_path_rate=0.2)
for n, p in m.named_parameters():
    if "alpha_" in n:
        p.data.fill_(0.5)
m.cuda().train()
opt = FiberAdamW([p for p in m.parameters()], lr=1e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
img = torch.randn(B, 3, H, W, device="cuda")
ids = torch.randint(3, 50000, (B, S), device="cuda")
ids[:, 0] = 0
am = torch.ones_like(ids)


def flops():
    """algorithmic forward GFLOP per image (2 x MACs of the matmuls), the survey's counting rule"""
    f, h, w, C = 0.0, H // 4, W // 4, 128
    f += 2 * h * w * 48 * 128
    for i, depth in enumerate((2, 2, 18, 2)):
        L, Lp = h * w, (-(-h // 12) * 12) * (-(-w // 12) * 12)
        for j in range(depth):
            f += 24 * L * C * C + 2 * Lp * C * 3 * C - 6 * L * C * C + 4 * Lp * 144 * C      # qkv runs on the padded grid
            if (i == 2 and j >= 14) or i == 3:
                f += 4 * L * C * C + 4 * S * 768 * C + 4 * L * S * C                        # q_i2t, proj_i2t, kv_text, i2t attention
                f += 4 * L * C * 768 + 4 * 768 * 768 * S + 4 * S * L * 768                   # t2i k / v, q / o, attention
        if i < 3:
            f += 2 * (L // 4) * 4 * C * 2 * C
            h, w, C = h // 2, w // 2, 2 * C
    f += 12 * (24 * S * 768 * 768 + 4 * S * S * 768)
    return f / 1e9


def step():
    vis, lang, _ = m({"input_ids": ids, "attention_mask": am}, img)
    loss = sum(v.float().square().mean() for v in vis) + lang["hidden"].float().square().mean()
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.time()
n = 8
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.time() - t) / n
gf = flops()
print(f"fine-grained fused backbone {H}x{W}, S={S}, B={B}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} images/s; "
      f"{gf:.0f} GFLOP forward per image -> {3 * gf * B / dt / 1e3:.0f} TFLOP/s algorithmic (fwd + 2x bwd), "
      f"peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
