"""Data containers of the hot path (host-side mirror of the reference interface).

Mirrors mb_vae_dti/training/data_containers.py:14-210 and PlaceHolder of
mb_vae_dti/training/diffusion/utils.py:20-57: same field names, same semantics.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Union

import torch


class PlaceHolder:
    """(X, E, y) triple for node / edge / global tensors."""

    def __init__(self, X, E, y=None):
        self.X, self.E, self.y = X, E, y

    def type_as(self, x):
        self.X = self.X.type_as(x)
        self.E = self.E.type_as(x)
        if torch.is_tensor(self.y):
            self.y = self.y.type_as(x)
        return self

    def mask(self, node_mask, collapse=False):
        """Zero padded nodes / pairs; with collapse=True return class ids with -1 on padding."""
        xm = node_mask.unsqueeze(-1)
        pair = xm.unsqueeze(2) * xm.unsqueeze(1)
        if collapse:
            self.X = torch.argmax(self.X, dim=-1)
            self.E = torch.argmax(self.E, dim=-1)
            self.X[node_mask == 0] = -1
            self.E[pair.squeeze(-1) == 0] = -1
        else:
            self.X = self.X * xm
            self.E = self.E * pair
        return self


@dataclass
class GraphData:
    pyg_batch: Any = None
    X: Optional[torch.Tensor] = None
    E: Optional[torch.Tensor] = None
    node_mask: Optional[torch.Tensor] = None
    X_t: Optional[torch.Tensor] = None
    E_t: Optional[torch.Tensor] = None
    X_extra: Optional[torch.Tensor] = None
    y_extra: Optional[torch.Tensor] = None
    noise_params: Optional[Dict[str, torch.Tensor]] = None


@dataclass
class EmbeddingData:
    drug_embedding: Optional[torch.Tensor] = None
    target_embedding: Optional[torch.Tensor] = None
    drug_mu: Optional[torch.Tensor] = None
    drug_logvar: Optional[torch.Tensor] = None
    drug_attention: Optional[torch.Tensor] = None
    target_attention: Optional[torch.Tensor] = None
    fused_embedding: Optional[torch.Tensor] = None


@dataclass
class PredictionData:
    dti_scores: Optional[Union[Dict[str, torch.Tensor], torch.Tensor]] = None
    graph_reconstruction: Optional[PlaceHolder] = None


@dataclass
class LossData:
    accuracy: Optional[torch.Tensor] = None
    complexity: Optional[torch.Tensor] = None
    contrastive: Optional[torch.Tensor] = None
    reconstruction: Optional[torch.Tensor] = None
    components: Dict[str, torch.Tensor] = field(default_factory=dict)

    def compute_loss(self, weights: List[float]) -> torch.Tensor:
        """sum_i w_i * L_i over (accuracy, complexity, contrastive, reconstruction), skipping w_i == 0."""
        if len(weights) != 4:
            raise AssertionError("Weights must be a list of 4 elements (accuracy, complexity, contrastive, reconstruction)")
        parts = [self.accuracy, self.complexity, self.contrastive, self.reconstruction]
        dev = next((p.device for p in parts if p is not None), torch.device("cpu"))
        total = torch.zeros((), device=dev)
        for w, p in zip(weights, parts):
            if w != 0 and p is not None:
                total = total + w * p
        return total


@dataclass
class BatchData:
    raw_batch: Optional[Dict[str, Any]] = None
    drug_features: Any = None
    target_features: Any = None
    graph_data: Optional[GraphData] = None
    dti_targets: Any = None
    dti_masks: Any = None
    drug_fp: Optional[torch.Tensor] = None
    target_fp: Optional[torch.Tensor] = None
    smiles: Optional[List[str]] = None
