"""What would a whole-step HIP graph of the denoiser train step buy?  Captures one step (noising, graph features,
forward, masked CE, backward, AdamW) with torch.cuda.graph and times replays next to eager steps, one and two streams.
The dropout seeds are kernel arguments, so a replay repeats the captured masks: good for timing, not for training.
GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs

dev = torch.device("cuda", 0)
K = int(os.environ.get("PROBE_STEPS", "20"))
modes = [m == "1" for m in os.environ.get("PROBE_SIDE", "1,0").split(",")]


def timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for side in modes:
    torch.manual_seed(0)
    tr = DenoiserTrainStep(CFG3, device=dev, distributed=False, seed=0, side_stream=side, overlap_optimizer=False)
    tr.opt = torch.optim.AdamW(tr.model.parameters(), lr=5e-4, weight_decay=1e-4, amsgrad=True, fused=True, capturable=True)
    g = torch.Generator().manual_seed(1)
    X0, E0, mask = synthetic_graphs(256, 64, dev, g)
    cond = torch.randn(256, 1024, generator=g).to(dev)
    step = lambda: tr.step(X0, E0, mask, cond)
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(warm)
    eager = timed(step, K)
    print(f"second stream {side}: eager {eager:.2f} ms/step (optimiser not overlapped with the next step)", flush=True)
    graph = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    try:
        with torch.cuda.graph(graph):
            loss = step()
    except Exception as e:                                   # report and go on with the next mode
        print(f"second stream {side}: capture failed: {type(e).__name__}: {str(e)[:600]}", flush=True)
        continue
    torch.cuda.synchronize()
    print(f"second stream {side}: captured and instantiated in {time.perf_counter() - t0:.1f} s", flush=True)
    graph.replay(); torch.cuda.synchronize()
    print(f"second stream {side}: loss of the first replay {float(loss):.5f}", flush=True)
    rep = timed(graph.replay, K)
    print(f"second stream {side}: replay {rep:.2f} ms/step against eager {eager:.2f} ms/step; loss after {K + 1} replays {float(loss):.5f}",
          flush=True)
    del graph
