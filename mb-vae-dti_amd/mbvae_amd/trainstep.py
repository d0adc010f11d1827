"""Denoiser train step on synthetic molecule graphs: noise -> extra features -> denoiser -> masked CE -> backward
-> (gradient all-reduce) -> AdamW.  Used by bench.py and the smoke test; mirrors the order of operations of
FullDTIModel._common_step / training_step for the decoder branch (mb_vae_dti/training/modules/full.py:816-952,
:1174-1211) with the optimiser of AbstractDTIModel.configure_optimizers (training/modules/utils.py:109-116).
"""
import os

import torch
import torch.nn.functional as F

from . import ops
from .ddp import BucketedGradSync
from .diffusion import PredefinedNoiseScheduleDiscrete, exp_draws
from .features import ExtraFeatures, ExtraMolecularFeatures
from .graph_transformer import GraphTransformer, refresh_stacked_weights

CFG3 = dict(  # the benchmark denoiser: 8 layers, 5 atom / 5 bond classes, reference widths
    n_layers=8, input_dims={"X": 13, "E": 5, "y": 1037}, output_dims={"X": 5, "E": 5, "y": 1024},
    hidden_mlp_dims={"X": 256, "E": 128, "y": 2048},
    hidden_dims={"dx": 256, "de": 64, "dy": 1024, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 1024})
MX = (0.72, 0.12, 0.10, 0.04, 0.02)
ME = (0.90, 0.05, 0.03, 0.01, 0.01)
VALENCIES = (4, 2, 3, 3, 2)
ATOM_WEIGHTS = (12.011, 15.999, 30.974, 14.007, 32.06)


def synthetic_graphs(bs, n, device, generator, sizes=None):
    """Clean graphs in the dense convention of the reference: X one-hot (all-zero on padding), E one-hot with
    class 0 = no bond on every off-diagonal pair (padded ones too), all-zero diagonal."""
    mx, me = torch.tensor(MX), torch.tensor(ME)
    Xc = torch.multinomial(mx, bs * n, True, generator=generator).reshape(bs, n)
    Ec = torch.multinomial(me, bs * n * n, True, generator=generator).reshape(bs, n, n)
    Ec = torch.triu(Ec, 1)
    Ec = Ec + Ec.transpose(1, 2)
    mask = torch.ones(bs, n, dtype=torch.bool) if sizes is None else torch.arange(n)[None, :] < sizes[:, None]
    Ec = Ec * (mask.unsqueeze(1) & mask.unsqueeze(2))
    X = F.one_hot(Xc, len(MX)).float() * mask.unsqueeze(-1)
    E = F.one_hot(Ec, len(ME)).float()
    E[:, torch.arange(n), torch.arange(n)] = 0
    return X.to(device), E.to(device), mask.to(device)


class DenoiserTrainStep:
    def __init__(self, cfg=CFG3, T=500, lr=5e-4, weight_decay=1e-4, lambdas=(1.0, 5.0), device="cuda", distributed=False,
                 seed=0, side_stream=None, overlap_optimizer=False):
        # side_stream: the node / global-feature remainder of every layer on a second HIP stream next to the long edge-tensor
        # kernels (ops.set_side_stream; bit-identical to one stream, tests/test_gpu_two_streams.py).  None = on for a GPU
        # model unless MBVAE_SIDE_STREAM=0; an explicit True / False is never overridden by the environment.  With
        # gradient-sync hooks the collectives order themselves behind both streams (ddp.BucketedGradSync).
        if side_stream is None:
            side_stream = torch.device(device).type == "cuda" and os.environ.get("MBVAE_SIDE_STREAM", "1") not in ("", "0")
        ops.set_side_stream(bool(side_stream))
        torch.manual_seed(seed)                      # identical initial weights on every rank
        self.device = torch.device(device)
        self.model = GraphTransformer(**cfg).to(self.device).train()
        self.schedule = PredefinedNoiseScheduleDiscrete(T).to(self.device)
        self.T = T
        self.lambdas = lambdas
        self.mx = torch.tensor(MX, device=self.device)
        self.me = torch.tensor(ME, device=self.device)
        self.graph_feats = ExtraFeatures(max_n_nodes=64)
        self.mol_feats = ExtraMolecularFeatures(VALENCIES, 1000.0, ATOM_WEIGHTS)
        # the denoiser's autograd graph is the same every step: the set of parameters without gradient (the global
        # output head when only node / edge logits enter the loss) is agreed on once
        self.sync = BucketedGradSync(self.model.parameters(), static_graph=True) if distributed else None
        # same update rule as the reference's AdamW(amsgrad=True); `fused` runs it as one multi-tensor kernel per
        # dtype/device group instead of ~10 element-wise passes over the 73 M parameters
        self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=weight_decay, amsgrad=True,
                                     fused=True)
        self.last = {}
        self._params = [p for p in self.model.parameters() if p.requires_grad]
        self._opt_pending = None
        # overlap_optimizer (needs the second stream): the weight update is left running on the second stream when step()
        # returns and the NEXT step's denoiser call waits for it; anyone else who reads the weights calls
        # wait_optimizer() first.  Off: step() returns with the update ordered into the caller's stream.
        self.overlap_optimizer = overlap_optimizer
        # identical weights everywhere, but every rank draws its own timesteps, noise and dropout seeds
        rank = torch.distributed.get_rank() if (distributed and torch.distributed.is_initialized()) else 0
        torch.manual_seed(seed + 1 + rank)

    def wait_optimizer(self):
        """Orders a weight update that was left running on the second stream into the current stream."""
        if self._opt_pending is not None:
            torch.cuda.current_stream().wait_stream(self._opt_pending)
            self._opt_pending = None

    def step(self, X0, E0, mask, cond):
        bs, n, cx = X0.shape
        ce = E0.shape[-1]
        dev = self.device
        # forward diffusion q(G_t | G_0): same RNG draw order as the reference (t, then X, then E)
        t_int = torch.randint(0, self.T + 1, (bs, 1), device=dev).float()
        t = t_int / self.T
        abar_t = self.schedule.get_alpha_bar(t_normalized=t)
        eX, eE = exp_draws(bs * n, cx, dev), exp_draws(bs * n * n, ce, dev)
        _, _, X_t, E_t = ops.apply_noise(X0, E0, mask, abar_t, self.mx, self.me, eX, eE)
        Xg, yg = self.graph_feats(E_t, mask)
        Xm, ym = self.mol_feats(X_t, E_t, truncate_weight=True)
        X_in = torch.cat((X_t, Xg, Xm), 2)
        y_in = torch.cat((cond, yg, ym, t), 1)
        self.wait_optimizer()                       # the previous step's weight update may still run on the second stream
        pred = self.model(X_in, E_t, y_in, mask)
        lX, sX = ops.masked_ce(pred.X, X0)
        lE, sE = ops.masked_ce(pred.E, E0)
        loss = self.lambdas[0] * lX + self.lambdas[1] * lE
        loss.backward()
        if self.sync is not None:
            self.sync.finish()
        side = ops.side_stream(dev) if self.sync is None else None
        if side is None:
            self.opt.step()
            refresh_stacked_weights(self.model)
        else:
            # The optimiser's sweep over the weights (memory bound) on the second stream: the next step's noising and
            # graph features (latency bound, no weights involved) start next to it; the denoiser waits for it below.
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            for p in self._params:
                if p.grad is not None:
                    p.grad.record_stream(side)       # (freed right after this call; the allocator must wait for the sweep)
            with torch.cuda.stream(side):
                self.opt.step()
                refresh_stacked_weights(self.model)      # (the next forward then finds its stacked projections ready)
            self._opt_pending = side
            if not self.overlap_optimizer:
                self.wait_optimizer()
        if self.sync is not None:
            self.sync.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        self.last = {"loss": loss.detach(), "ce_X": sX, "ce_E": sE}
        return loss.detach()
