import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd.trainstep import CFG3, DenoiserTrainStep, synthetic_graphs
g = torch.Generator().manual_seed(1000)
sizes = torch.randint(30, 65, (256,), generator=g); sizes[0] = 64
X0, E0, mask = synthetic_graphs(256, 64, "cuda", g, sizes=None if os.environ.get("CHECK_FULL") == "1" else sizes)
cond = torch.randn(256, 1024, generator=g).cuda()
runs = []
OVERLAP = os.environ.get("CHECK_OVERLAP", "1") == "1"
for side in (False, True, True):
    tr = DenoiserTrainStep(CFG3, device="cuda", seed=0, side_stream=side, overlap_optimizer=side and OVERLAP)
    torch.manual_seed(123)
    losses = torch.stack([tr.step(X0, E0, mask, cond) for _ in range(int(os.environ.get("CHECK_STEPS", "40")))])
    tr.wait_optimizer(); torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for p in tr.model.parameters()])
    runs.append((losses.cpu(), chk.cpu()))
    del tr
    torch.cuda.empty_cache()
def first_diff(a, b):
    d = (a != b).nonzero()
    return int(d[0]) if d.numel() else -1
for i, j in ((0, 1), (1, 2)):
    print(f"runs {i} vs {j}: first differing step {first_diff(runs[i][0], runs[j][0])}, weights equal {torch.equal(runs[i][1], runs[j][1])}")
