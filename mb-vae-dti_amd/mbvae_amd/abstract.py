"""Shared base of the DTI model variants on the HIP path.

Drop-in for AbstractDTIModel, mb_vae_dti/training/modules/utils.py:28-451: activation parsing (:62-106), optimiser and
scheduler set-up (:109-177), metric slots (:180-222), encoder freezing (:225-251), shape-matched checkpoint merge
(:253-279), the batch-dictionary accessors (:286-388), dense graph extraction (:391-426) and the epoch-end metric hooks
(:429-451).  It subclasses pytorch_lightning.LightningModule when that package is importable and a small nn.Module
stand-in with the same surface otherwise.

The DTI / molecule metric collections of the reference (torchmetrics / lifelines / torcheval / RDKit wrappers) are not
part of this package: the metric slots stay None unless the caller assigns objects with update / compute / reset.
"""
import inspect
import logging
from pathlib import Path

import torch
import torch.nn as nn

from . import ops
from .containers import GraphData, PlaceHolder

logger = logging.getLogger(__name__)

try:  # pragma: no cover - depends on the environment
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class _Base(nn.Module):
        """Minimal LightningModule surface: hparams, device, log() (values kept in `logged`), trainer, current_epoch."""

        def __init__(self):
            super().__init__()
            self.hparams = _AttrDict()
            self.trainer = None
            self.current_epoch = 0
            self.logged = {}

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            for name in info.args:
                if name != "self":
                    self.hparams[name] = info.locals[name]

        @property
        def device(self):
            p = next(self.parameters(), None)
            return p.device if p is not None else torch.device("cpu")

        def log(self, name, value, *args, **kwargs):
            self.logged[name] = value

        def on_train_start(self):
            pass


ACTIVATIONS = {"relu": nn.ReLU, "gelu": nn.GELU, "elu": nn.ELU, "leaky_relu": nn.LeakyReLU, "selu": nn.SELU, "swish": nn.SiLU,
               "silu": nn.SiLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "mish": nn.Mish, "glu": nn.GLU}


# ------------------------------------------------------------------------------------------------ dense batches
def encode_no_edge(E):
    """Prepend the 'no bond' class (1 where no bond type is set) and clear the diagonal (diffusion/utils.py:102-120)."""
    none = (E.sum(dim=3, keepdim=True) == 0).to(E.dtype)
    out = torch.cat((none, E), dim=3)
    n = E.shape[1]
    out[:, torch.arange(n), torch.arange(n)] = 0
    return out


def to_dense(x, edge_index, edge_attr, batch):
    """Sparse (PyG-style) batch -> dense X [bs,n,cx], E [bs,n,n,1+bond types], node_mask [bs,n]; self loops dropped.
    Follows to_dense_batch / to_dense_adj semantics (duplicate edges add up).  X and E are NOT masked."""
    bs = int(batch.max().item()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=bs)
    n = int(counts.max().item()) if bs else 0
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(batch.numel(), device=batch.device) - start[batch]
    X = x.new_zeros(bs, n, x.shape[-1])
    X[batch, pos] = x
    node_mask = torch.zeros(bs, n, dtype=torch.bool, device=x.device)
    node_mask[batch, pos] = True
    keep = edge_index[0] != edge_index[1]
    src, dst, attr = edge_index[0][keep], edge_index[1][keep], edge_attr[keep]
    E = attr.new_zeros(bs, n, n, attr.shape[-1])
    E.index_put_((batch[src], pos[src], pos[dst]), attr, accumulate=True)
    return PlaceHolder(X=X, E=encode_no_edge(E), y=None), node_mask


class AbstractDTIModel(_Base):
    def __init__(self):
        super().__init__()
        self.train_metrics = self.val_metrics = self.test_metrics = None

    # ------------------------------------------------------------------ bookkeeping
    def count_trainable_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def on_train_start(self):
        super().on_train_start()
        n = self.count_trainable_parameters()
        self.log("model/trainable_parameters", float(n), on_step=False, on_epoch=True)
        logger.info("Model has %s trainable parameters", f"{n:,}")

    @staticmethod
    def parse_activation(activation):
        if isinstance(activation, nn.Module):
            return activation
        if isinstance(activation, str):
            if activation.lower() in ACTIVATIONS:
                return ACTIVATIONS[activation.lower()]()
            raise ValueError(f"Unknown activation '{activation}'. Available activations: {list(ACTIVATIONS)}")
        raise TypeError(f"Activation must be either a string or nn.Module, got {type(activation)}")

    def configure_optimizers(self):
        """AdamW(amsgrad) + {const, step, one_cycle, cosine} exactly as modules/utils.py:109-177."""
        lr = float(self.hparams.learning_rate)
        opt = torch.optim.AdamW(self.parameters(), lr=lr, weight_decay=float(self.hparams.weight_decay), amsgrad=True)
        sched = self.hparams.scheduler
        if sched == "const" or sched is None:
            return opt
        if sched == "step":
            s = torch.optim.lr_scheduler.StepLR(opt, step_size=30, gamma=0.1)
        elif sched == "one_cycle":
            if self.trainer is None:
                raise ValueError("Trainer must be provided for OneCycleLR scheduler")
            s = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=self.trainer.estimated_stepping_batches,
                                                    pct_start=0.3, anneal_strategy="cos")
            return [opt], [{"scheduler": s, "name": "learning_rate", "interval": "step", "frequency": 1}]
        elif sched == "cosine":
            t_max = self.trainer.max_epochs if self.trainer else self.hparams.max_epochs
            s = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t_max, eta_min=lr * 0.01)
        else:
            raise ValueError(f"Unknown scheduler: {sched}")
        return {"optimizer": opt, "lr_scheduler": {"scheduler": s, "monitor": "val/loss", "interval": "epoch", "frequency": 1}}

    def setup_metrics(self, phase, finetune_score=None):
        """The reference fills the three slots with torchmetrics collections (modules/utils.py:180-222); those wrappers
        are outside this package, so the slots are left empty for the caller to fill."""
        if phase == "finetune" and finetune_score is None:
            raise ValueError("finetune_score must be specified for finetune phase")
        self.train_metrics = self.val_metrics = self.test_metrics = None

    def _set_requires_grad(self, flag):
        for name in ("drug_encoder", "target_encoder", "drug_encoders", "target_encoders"):
            mod = getattr(self, name, None)
            if mod is not None:
                for p in mod.parameters():
                    p.requires_grad = flag

    def freeze_encoders(self):
        self._set_requires_grad(False)

    def unfreeze_encoders(self):
        self._set_requires_grad(True)

    def load_pretrained_weights(self, checkpoint_path, prefix=None):
        """Shape-matched, optionally prefix-filtered state_dict merge (modules/utils.py:253-279).  The checkpoint is
        read with weights_only=True (tensors only)."""
        checkpoint_path = Path(checkpoint_path)
        if not checkpoint_path.exists():
            raise FileNotFoundError(f"Checkpoint not found: {checkpoint_path}")
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        own = self.state_dict()
        picked = {k: v for k, v in state.items()
                  if k in own and own[k].shape == v.shape and (prefix is None or k.startswith(prefix))}
        own.update(picked)
        self.load_state_dict(own)
        logger.info("Loaded %d pretrained weights", len(picked))

    # ------------------------------------------------------------------ batch dictionary accessors
    def _get_features_from_batch(self, batch):
        hp = self.hparams
        if hp.phase == "pretrain_drug":
            d, t = [batch["features"][k] for k in hp.drug_features], []
        elif hp.phase == "pretrain_target":
            d, t = [], [batch["features"][k] for k in hp.target_features]
        else:
            d = [batch["drug"]["features"][k] for k in hp.drug_features]
            t = [batch["target"]["features"][k] for k in hp.target_features]
        return (d[0] if len(d) == 1 else d), (t[0] if len(t) == 1 else t)

    def _get_smiles_from_batch(self, batch):
        if self.hparams.phase == "pretrain_drug":
            return batch["representations"]["smiles"]
        return batch["drug"]["representations"]["smiles"]

    def _get_fingerprints_from_batch(self, batch):
        if self.hparams.phase == "pretrain_drug":
            return batch["features"]["FP-Morgan"], None
        if self.hparams.phase == "pretrain_target":
            return None, batch["features"]["FP-ESP"]
        return batch["drug"]["features"]["FP-Morgan"], batch["target"]["features"]["FP-ESP"]

    def _get_targets_from_batch(self, batch):
        y = batch["y"]
        if self.hparams.phase == "finetune":
            return y[self.hparams.finetune_score].squeeze(-1)
        return {k: y[k].squeeze(-1) for k in ("Y", "Y_pKd", "Y_KIBA", "Y_pKi")}

    def _get_masks_from_batch(self, batch):
        y = batch["y"]
        if self.hparams.phase == "finetune":
            return y[f"{self.hparams.finetune_score}_mask"].squeeze(-1)
        return {k: y[f"{k}_mask"].squeeze(-1) for k in ("Y_pKd", "Y_KIBA", "Y_pKi")}

    def _get_targets_masks_from_batch(self, batch):
        return self._get_targets_from_batch(batch), self._get_masks_from_batch(batch)

    def _extract_graph_data(self, batch):
        """Dense clean graph from batch['drug']['G'] / batch['G']: any object with x, edge_index, edge_attr, batch
        (PyG style), a ready GraphData, or {'atom_ids': int8 [bs,n], 'bond_ids': int8 [bs,n,n]} class ids.
        X and E are not masked; node_mask is returned alongside."""
        G = batch["drug"]["G"] if "drug" in batch and "G" in batch["drug"] else batch.get("G")
        if G is None:
            raise ValueError("No graph data found in batch")
        if isinstance(G, GraphData):
            return G
        if isinstance(G, dict) and "atom_ids" in G:      # compact wire format: int8 class ids, expanded on the device
            X, E, node_mask = ops.dense_from_ids(G["atom_ids"].to(self.device), G["bond_ids"].to(self.device),
                                                 len(self.atom_types), self.limit_dist.E.numel())
            return GraphData(X=X, E=E, node_mask=node_mask)
        dense, node_mask = to_dense(G.x, G.edge_index, G.edge_attr, G.batch)
        return GraphData(pyg_batch=G, X=dense.X, E=dense.E, node_mask=node_mask)

    # ------------------------------------------------------------------ epoch-end metric hooks
    def _flush_metrics(self, metrics):
        if metrics is not None:
            for name, value in metrics.compute().items():
                self.log(name, value)
            metrics.reset()

    def on_train_epoch_end(self):
        self._flush_metrics(self.train_metrics)

    def on_validation_epoch_end(self):
        self._flush_metrics(self.val_metrics)

    def on_test_epoch_end(self):
        self._flush_metrics(self.test_metrics)
