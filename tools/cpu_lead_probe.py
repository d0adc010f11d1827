"""Does the host run ahead of the GPU during a train step, or does something inside the step synchronise?
Prints the host-side duration of each step() call (no synchronisation between calls) next to the synchronised
per-step time, and the host<->device copies / synchronisations torch's profiler sees in one step.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

dev = torch.device("cuda", 0)
tr = DenoiserTrainStep(CFG3, device=dev, distributed=False, seed=0)
g = torch.Generator().manual_seed(1000)
X0, E0, mask = synthetic_graphs(256, 64, dev, g)
cond = torch.randn(256, 1024, generator=g).to(dev)
for _ in range(3):
    tr.step(X0, E0, mask, cond)
torch.cuda.synchronize()
t0 = time.perf_counter(); host = []
for _ in range(6):
    a = time.perf_counter(); tr.step(X0, E0, mask, cond); host.append(1e3 * (time.perf_counter() - a))
torch.cuda.synchronize()
print("host ms per step() call:", [round(h, 1) for h in host], " synchronised ms/step:", round(1e3 * (time.perf_counter() - t0) / 6, 2))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(X0, E0, mask, cond)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    n = ev.name
    if "Synchronize" in n or "Memcpy" in n or "memcpy" in n or n in ("aten::item", "aten::_local_scalar_dense", "aten::nonzero", "aten::is_nonzero"):
        c[n] += 1
print(dict(c))
