import sys; sys.path.insert(0, "/root/repo")
import torch
from fiber_amd.config import make_config
from fiber_amd.modules import FIBERTransformerSS
from fiber_amd.trainer import Trainer
from oracle import cases, detgen
torch.manual_seed(0)
m = FIBERTransformerSS(make_config(**cases.TINY, learning_rate=1e-4, warmup_steps=0, max_steps=6))
for n, p in m.named_parameters():
    if "alpha_" in n: p.data.fill_(0.5)
data = [detgen.synth_batch(4, 96, 12, 1000, seed=s, min_len=6) for s in range(3)]
for b in data: b["itm_labels_override"] = b["itm_labels"]
last = Trainer(max_steps=6, accumulate_grad_batches=2, log_every_n_steps=2).fit(m, data)
print("trainer ok, last loss", float(last))
