"""Import the reference's hot-path modules in this container (CPU only).

Used ONLY by tools/gen_golden.py to emit the fixtures under tests/golden/.
Nothing here (and nothing from /root/reference) travels to the GPU box.

The reference cannot be imported as a plain package: its top-level __init__
pulls in data-loading / validation code and third-party packages that are not
installed here.  We therefore
  1. register empty namespace packages for `mb_vae_dti` and
     `mb_vae_dti.training` whose __path__ points at the reference tree, so the
     sub-packages (models, diffusion, metrics, modules) import normally;
  2. stub the absent third-party packages (rdkit, wandb, torch_geometric,
     torcheval, lifelines) with MagicMock modules -- none of them is touched by
     the arithmetic we capture;
  3. provide a minimal `pytorch_lightning.LightningModule` / `torchmetrics`
     shim (nn.Module + save_hyperparameters + add_state) so FullDTIModel can be
     constructed.  The shims perform no arithmetic.
"""
import inspect
import sys
import types
from unittest.mock import MagicMock

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"
sys.dont_write_bytecode = True


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _install_lightning_shim():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = _AttrDict()
            self.trainer = None

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            for name in args.args:
                if name != "self":
                    self.hparams[name] = args.locals[name]

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def on_train_start(self):
            pass

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl


def _install_torchmetrics_shim():
    tm = types.ModuleType("torchmetrics")

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default
            setattr(self, name, default.clone() if torch.is_tensor(default) else list(default))

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone() if torch.is_tensor(v) else list(v))

        def forward(self, *a, **k):
            self.update(*a, **k)
            return self.compute()

    class MetricCollection(nn.ModuleDict):
        def __init__(self, metrics=None, prefix=None, **k):
            super().__init__()
            items = metrics.items() if isinstance(metrics, dict) else [(type(m).__name__, m) for m in (metrics or [])]
            for n, m in items:
                if isinstance(m, nn.Module):      # mocked third-party metrics are dropped
                    self[n] = m

        def forward(self, *a, **k):
            return {n: m(*a, **k) for n, m in self.items()}

        def update(self, *a, **k):
            for m in self.values():
                m.update(*a, **k)

        def compute(self):
            return {n: m.compute() for n, m in self.items()}

        def reset(self):
            for m in self.values():
                m.reset()

    tm.Metric = Metric
    tm.MetricCollection = MetricCollection
    sys.modules["torchmetrics"] = tm
    for sub in ("regression", "classification"):
        m = MagicMock()
        sys.modules[f"torchmetrics.{sub}"] = m
        setattr(tm, sub, m)


def install():
    if "mb_vae_dti" in sys.modules:
        return
    for name, path in (("mb_vae_dti", f"{REF_ROOT}/mb_vae_dti"),
                       ("mb_vae_dti.training", f"{REF_ROOT}/mb_vae_dti/training")):
        mod = types.ModuleType(name)
        mod.__path__ = [path]
        sys.modules[name] = mod
    for name in ("rdkit", "rdkit.Chem", "rdkit.Chem.Draw", "rdkit.Chem.rdchem", "rdkit.Chem.AllChem",
                 "rdkit.RDLogger", "wandb", "torch_geometric", "torch_geometric.data",
                 "torch_geometric.utils", "torcheval", "torcheval.metrics", "lifelines",
                 "lifelines.utils", "tqdm"):
        if name == "tqdm":
            continue
        sys.modules[name] = MagicMock()
    _install_lightning_shim()
    _install_torchmetrics_shim()


def load():
    """Returns a namespace with the reference classes/functions on the hot path."""
    install()
    ns = types.SimpleNamespace()
    import mb_vae_dti.training.diffusion.utils as dutils
    import mb_vae_dti.training.diffusion.discrete_noise as dnoise
    import mb_vae_dti.training.diffusion.augmentation as daug
    import mb_vae_dti.training.models.graph_transformer as gt
    import mb_vae_dti.training.models.blocks as blocks
    import mb_vae_dti.training.models.heads as heads
    import mb_vae_dti.training.data_containers as dc
    ns.dutils, ns.dnoise, ns.daug, ns.gt, ns.blocks, ns.heads, ns.dc = dutils, dnoise, daug, gt, blocks, heads, dc
    return ns


def load_full():
    install()
    import mb_vae_dti.training.metrics.dti_metrics  # noqa: F401  (MagicMock bases are fine)
    import mb_vae_dti.training.modules.full as full
    # full.py:30 sets torch.set_float32_matmul_precision('medium') at import; with torch 2.10 that lets
    # oneDNN run fp32 CPU matmuls in bf16 (observed 7e-4 relative gradient error).  Goldens are true fp32.
    torch.set_float32_matmul_precision("highest")
    return full
