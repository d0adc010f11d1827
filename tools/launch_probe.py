"""Where do the small launches of a train step come from?  torch.profiler over two steps, grouped by Python stack
(GPU box only).   python tools/launch_probe.py [op substring ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

dev = torch.device("cuda", 0)
trainer = DenoiserTrainStep(CFG3, device=dev, distributed=False, seed=0)
g = torch.Generator().manual_seed(1000)
X0, E0, mask = synthetic_graphs(256, 64, dev, g)
cond = torch.randn(256, 1024, generator=g).to(dev)
for _ in range(2):
    trainer.step(X0, E0, mask, cond)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    trainer.step(X0, E0, mask, cond)
    torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::mul"]
from collections import Counter
cnt = Counter()
for ev in prof.events():
    if ev.name in want:
        stack = [s for s in ev.stack if "mbvae_amd" in s or "optim" in s][:3]
        cnt[(ev.name, tuple(stack))] += 1
allc = Counter(ev.name for ev in prof.events())
for name, c in allc.most_common(45):
    print(f"{c:6d} {name}")
for (name, stack), c in cnt.most_common(40):
    print(f"{c:5d} {name}")
    for s in stack:
        print("        ", s[-110:])
