"""Which parameter gradients of ONE denoiser forward + backward at the benchmark size differ between the one-stream and
the two-stream run (ops.set_side_stream)?  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops as O
from mbvae_amd.graph_transformer import GraphTransformer
from mbvae_amd.trainstep import CFG3, synthetic_graphs
cfg = dict(CFG3); cfg["n_layers"] = int(os.environ.get("CHECK_LAYERS", "8"))
torch.manual_seed(4)
net = GraphTransformer(**cfg).cuda().train()
g = torch.Generator().manual_seed(6)
bs, n = int(os.environ.get("CHECK_BS", "256")), 64
X0, E0, mask = synthetic_graphs(bs, n, "cuda", g)
Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda()), 2)
yin = torch.randn(bs, 1037, generator=g).cuda()


def run(side):
    O.set_side_stream(side)
    net.zero_grad(set_to_none=True)
    torch.manual_seed(21)
    out = net(Xin, E0, yin, mask)
    lX, _ = O.masked_ce(out.X, X0); lE, _ = O.masked_ce(out.E, E0)
    loss = lX + 5 * lE
    loss.backward()
    torch.cuda.synchronize()
    O.set_side_stream(False)
    return loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


ref = run(False)
for trial in range(3):
    got = run(True)
    bad = [k for k in ref[1] if not torch.equal(ref[1][k], got[1][k])]
    print(f"trial {trial}: loss equal {torch.equal(ref[0], got[0])}; {len(bad)} of {len(ref[1])} gradients differ")
    good = [k for k in ref[1] if k not in bad]
    print("    identical:", " ".join(good))
again = run(False)
print("one-stream rerun identical:", all(torch.equal(ref[1][k], again[1][k]) for k in ref[1]))
