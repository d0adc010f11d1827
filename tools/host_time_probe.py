"""Is the train step bound by the host (Python issuing ~700 launches) or by the GPU?  Times, per step, how long the
host needs to ISSUE a step (no synchronisation inside) next to the wall time per step, one and two streams.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

dev = torch.device("cuda", 0)
for side in (True, False):
    torch.manual_seed(0)
    tr = DenoiserTrainStep(CFG3, device=dev, distributed=False, seed=0, side_stream=side, overlap_optimizer=True)
    g = torch.Generator().manual_seed(1)
    sizes = torch.randint(30, 65, (256,), generator=g); sizes[0] = 64
    X0, E0, mask = synthetic_graphs(256, 64, dev, g, sizes=sizes)
    cond = torch.randn(256, 1024, generator=g).to(dev)
    for _ in range(3):
        tr.step(X0, E0, mask, cond)
    tr.wait_optimizer(); torch.cuda.synchronize()
    K = 10
    host = []
    t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        tr.step(X0, E0, mask, cond)
        host.append(time.perf_counter() - a)
    issued = time.perf_counter() - t0
    tr.wait_optimizer(); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"second stream {side}: host issue time per step {sum(host) / K * 1e3:.2f} ms (min {min(host) * 1e3:.2f}, max {max(host) * 1e3:.2f}); "
          f"all {K} steps issued after {issued * 1e3:.1f} ms, finished after {wall * 1e3:.1f} ms = {wall / K * 1e3:.2f} ms/step", flush=True)
