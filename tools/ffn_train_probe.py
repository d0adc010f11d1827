"""Times the one-pass training form of the edge feed-forward block (mbvae_edge_ffn_fwd_train) and the two-kernel route
at the benchmark's row count (256 graphs x 64 x 64 pairs).  GPU box only; MBVAE_LIB selects a library build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import _lib, ops as O

M = int(os.environ.get("PROBE_ROWS", 256 * 64 * 64))
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 64, generator=g).cuda()
w1 = (torch.randn(128, 64, generator=g) / 8).cuda(); b1 = (torch.randn(128, generator=g) * 0.1).cuda()
w2 = (torch.randn(64, 128, generator=g) / 11).cuda(); b2 = (torch.randn(64, generator=g) * 0.1).cuda()
gamma = torch.ones(64).cuda(); beta = torch.zeros(64).cuda()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for p1, p2 in ((0.1, 0.1), (0.0, 0.0)):
    with torch.no_grad():
        for name, off in (("one kernel", 0), ("two kernels", 1)):
            with _lib.option("NO_EDGE_FFN_TRAIN", off):
                us = timed(lambda: O._FfnLn.apply(x, w1, b1, w2, b2, gamma, beta, 1e-5, p1, 7, p2, 9) if not off else
                           O.linear_ln(O.linear(x, w1, b1, act=1, drop_p=p1, seed=7), w2, b2, x, gamma, beta, 1e-5, p2, 9))
            traffic = M * 4 * (64 + 128 + 64 + 64) if not off else M * 4 * (64 + 128 + 128 + 64 + 64 + 64)
            print(f"dropout {p1}/{p2}  {name:12s} {us:8.1f} us   {traffic / us / 1e3:7.1f} GB/s of its own traffic", flush=True)
