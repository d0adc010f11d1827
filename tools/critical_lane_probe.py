"""Which of the two streams of the training step does the step time wait for?  A one-thread spin kernel (torch.cuda._sleep)
is put into every layer's forward on one lane; the lane is on the critical path if the step gets slower by about
8 layers x delay, and has slack if it does not.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import graph_transformer as GT
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = DenoiserTrainStep(CFG3, device=dev, distributed=False, seed=0, side_stream=True, overlap_optimizer=True)
g = torch.Generator().manual_seed(1)
sizes = torch.randint(30, 65, (256,), generator=g); sizes[0] = 64
X0, E0, mask = synthetic_graphs(256, 64, dev, g, sizes=sizes)
cond = torch.randn(256, 1024, generator=g).to(dev)

# cycles per microsecond of the spin kernel's clock
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (a.elapsed_time(b) * 1e3)
print(f"spin clock: {cyc_per_us:.1f} cycles/us", flush=True)


def run(delay):
    GT.LANE_DELAY = delay
    for _ in range(3):
        tr.step(X0, E0, mask, cond)
    tr.wait_optimizer(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(X0, E0, mask, cond)
    tr.wait_optimizer(); torch.cuda.synchronize()
    GT.LANE_DELAY = None
    return (time.perf_counter() - t0) / 10 * 1e3


base = run(None)
print(f"no delay: {base:.2f} ms/step", flush=True)
for us in (100, 300):
    for lane in ("E", "N"):
        ms = run((lane, int(us * cyc_per_us)))
        print(f"{us} us per layer on lane {lane} (8 layers = {8 * us / 1e3:.1f} ms): {ms:.2f} ms/step, +{ms - base:.2f} ms", flush=True)
print(f"no delay again: {run(None):.2f} ms/step", flush=True)
