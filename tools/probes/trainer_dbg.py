import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cases, detgen
from tests.test_hip_trainer import _model
from fiber_amd import lib
from fiber_amd.trainer import Trainer, load_checkpoint
lib.load()
data = [detgen.synth_batch(4, 224, 40, 50265, seed=70 + s, min_len=8) for s in range(3)]
for b in data: b["itm_labels_override"] = b["itm_labels"]
val = data[:1]
tmp = tempfile.mkdtemp()
class Killed(RuntimeError): pass
class Dies(list):
    epochs = 0
    def __iter__(self):
        self.epochs += 1
        if self.epochs == 3: raise Killed()
        return super().__iter__()
def run(tag, max_steps, resume=None, total=6, kill=False):
    m = _model(total)
    losses = []
    orig = m.training_step
    def ts(batch, idx):
        l = orig(batch, idx); losses.append(float(l)); return l
    m.training_step = ts
    t = Trainer(max_steps=max_steps, accumulate_grad_batches=2, log_every_n_steps=0, default_root_dir=os.path.join(tmp, tag), resume_from_checkpoint=resume)
    last = None
    try:
        last = t.fit(m, Dies(data) if kill else data, val_dataloader=val)
    except Killed:
        pass
    print(tag, "last", None if last is None else float(last), "losses", [round(x, 6) for x in losses])
    return m
a = run("a", 6)
a2 = run("a2", 6)
b = run("b", 6, kill=True)
c = run("c", 6, resume=os.path.join(tmp, "b", "last.ckpt"))
sa, sc = a.state_dict(), c.state_dict()
worst = sorted(((float((sa[k].float() - sc[k].float()).abs().max()), k) for k in sa if sa[k].is_floating_point()), reverse=True)[:5]
print("worst a vs c", worst)
sa2 = a2.state_dict()
print("worst a vs a2", sorted(((float((sa[k].float() - sa2[k].float()).abs().max()), k) for k in sa if sa[k].is_floating_point()), reverse=True)[:3])
