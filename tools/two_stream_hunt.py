"""Where does the second-stream mode (ops.set_side_stream) stop being bit-identical to the one-stream run?  GPU box only.

One denoiser forward + backward at the benchmark size is repeated under switches that each remove ONE possible cause
without removing the others:

  keepalive   every tensor any torch op or autograd node creates stays referenced until the step has been synchronised:
              the caching allocator can then never hand a block to a second owner inside a step.  If the differences
              vanish, the cause is a block reused while a kernel of the other stream still reads or writes it.
  serial[K]   the C-ABI calls named in K (and/or all torch ops) run alone on the GPU: the issuing stream first waits for
              the other one and the other one then waits for it.  Streams, pools and allocation order stay exactly as
              in the two-stream run; only the overlap on the device goes away.  If the differences vanish, the cause
              needs two kernels running at the same time.

`serialise only X` for every entry point X then names the calls that must overlap with something for the differences to
appear.
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_map_only

from mbvae_amd import _lib, ops as O
from mbvae_amd.graph_transformer import GraphTransformer
from mbvae_amd.trainstep import CFG3, synthetic_graphs

cfg = dict(CFG3); cfg["n_layers"] = int(os.environ.get("CHECK_LAYERS", "8"))
TRIALS = int(os.environ.get("CHECK_TRIALS", "3"))
torch.manual_seed(4)
net = GraphTransformer(**cfg).cuda().train()
g = torch.Generator().manual_seed(6)
bs, n = int(os.environ.get("CHECK_BS", "256")), 64
if os.environ.get("CHECK_RAGGED") == "1":
    sizes = torch.randint(30, 65, (bs,), generator=g); sizes[0] = 64
else:
    sizes = None
X0, E0, mask = synthetic_graphs(bs, n, "cuda", g, sizes=sizes)
Xin = torch.cat((X0, torch.randn(bs, n, 8, generator=g).cuda()), 2)
yin = torch.randn(bs, 1037, generator=g).cuda()

MAIN = torch.cuda.current_stream()
O.set_side_stream(True)
SIDE = O.side_stream(Xin.device)
O.set_side_stream(False)

SERIAL = set()            # entries: ("lib", name, "M" | "S"), ("torch",)  or the wildcards ("lib*",)
SEEN = collections.Counter()
KEEP = None               # list while the keep-alive switch is on


def _which(cur):
    return "S" if cur.cuda_stream == SIDE.cuda_stream else "M"


def _fence_before():
    cur = torch.cuda.current_stream()
    oth = MAIN if _which(cur) == "S" else SIDE
    cur.wait_stream(oth)
    return cur, oth


_orig_call = _lib.call


def _call(name, *args, **kw):
    key = ("lib", name, _which(torch.cuda.current_stream()))
    SEEN[key] += 1
    if ("lib*",) in SERIAL or key in SERIAL:
        cur, oth = _fence_before()
        r = _orig_call(name, *args, **kw)
        oth.wait_stream(cur)
        return r
    return _orig_call(name, *args, **kw)


_lib.call = _call


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        fence = ("torch",) in SERIAL
        if fence:
            cur, oth = _fence_before()
        out = func(*args, **(kwargs or {}))
        if fence:
            oth.wait_stream(cur)
        if KEEP is not None:
            tree_map_only(torch.Tensor, KEEP.append, out)
        return out


def run(side):
    global KEEP
    O.set_side_stream(side)
    net.zero_grad(set_to_none=True)
    torch.manual_seed(21)
    with Mode():
        out = net(Xin, E0, yin, mask)
        lX, _ = O.masked_ce(out.X, X0); lE, _ = O.masked_ce(out.E, E0)
        loss = lX + 5 * lE
        loss.backward()
    torch.cuda.synchronize()
    O.set_side_stream(False)
    res = loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    if KEEP is not None:
        KEEP.clear()
    return res


ref = run(False)
again = run(False)
print("one-stream rerun identical:", all(torch.equal(ref[1][k], again[1][k]) for k in ref[1]), flush=True)


def trial(label, serial=(), keep=False, trials=TRIALS):
    global KEEP
    SERIAL.clear(); SERIAL.update(serial)
    KEEP = [] if keep else None
    counts = []
    for _ in range(trials):
        got = run(True)
        bad = [k for k in ref[1] if not torch.equal(ref[1][k], got[1][k])]
        counts.append((len(bad), bool(torch.equal(ref[0], got[0]))))
    KEEP = None
    SERIAL.clear()
    print(f"{label:70s} differing gradients per trial (of {len(ref[1])}): " + " ".join(f"{c}{'' if ok else '!loss'}" for c, ok in counts), flush=True)
    return max(c for c, _ in counts)


trial("two streams, as is")
keys = sorted(SEEN)
print("entry points seen (stream M = main, S = second):", " ".join(f"{k[1][6:]}@{k[2]}x{SEEN[k]}" for k in keys), flush=True)
trial("keep-alive (no block reuse inside a step)", keep=True)
trial("every library call and torch op alone on the GPU", serial=[("lib*",), ("torch",)])
trial("every library call alone on the GPU", serial=[("lib*",)])
trial("every torch op alone on the GPU", serial=[("torch",)])
trial("keep-alive + every call alone", serial=[("lib*",), ("torch",)], keep=True)
print("--- serialise ONLY one entry point (differences vanish => that call must overlap with something)", flush=True)
for k in keys:
    trial(f"  only {k[1]}@{k[2]}", serial=[k], trials=2)
