"""The four non-diffusion DTI model variants on the HIP encoders / fusion / heads.

Drop-ins (same constructor arguments, attribute / state_dict names, batch dictionary, return values) for
  BaselineDTIModel     mb_vae_dti/training/modules/baseline.py:19-219      two encoders -> dot product -> masked MSE
  MultiModalDTIModel   mb_vae_dti/training/modules/multi_modal.py:19-341   per-view encoders -> aggregator -> dot product
  MultiOutputDTIModel  mb_vae_dti/training/modules/multi_output.py:20-323  two encoders -> fusion -> DTI head (4 scores)
  MultiHybridDTIModel  mb_vae_dti/training/modules/multi_hybrid.py:19-480  views -> aggregators -> fusion -> DTI head + InfoNCE
Every Linear / LayerNorm runs on the HIP kernels through blocks.py / heads.py; the [B]-sized tails (dot product, masked
MSE / BCE) are torch device ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .abstract import AbstractDTIModel
from .blocks import AttentiveAggregator, ConcatAggregator, CrossAttentionFusion, ResidualEncoder, TransformerEncoder
from .containers import BatchData, EmbeddingData, LossData, PredictionData
from .heads import DTIHead, InfoNCEHead

DTI_SCORES = ("Y", "Y_pKd", "Y_pKi", "Y_KIBA")


def _encoder_cls(encoder_type):
    return ResidualEncoder if encoder_type == "resnet" else TransformerEncoder


def _view_branch(features, encoder_type, aggregator_type, embedding_dim, hidden_dim, n_layers, factor, activation, dropout, bias):
    """(ModuleDict of per-view encoders, aggregator): views project to hidden_dim and are aggregated to embedding_dim;
    a single view goes straight to embedding_dim through nn.Identity."""
    many = len(features) > 1
    encoders = nn.ModuleDict({
        name: _encoder_cls(encoder_type)(input_dim=dim, hidden_dim=hidden_dim, output_dim=hidden_dim if many else embedding_dim,
                                         n_layers=n_layers, factor=factor, activation=activation, dropout=dropout, bias=bias)
        for name, dim in features.items()})
    Agg = ConcatAggregator if aggregator_type == "concat" else AttentiveAggregator
    agg = Agg(input_dim=hidden_dim, n_features=len(features), output_dim=embedding_dim) if many else nn.Identity()
    return encoders, agg


def _aggregate(encoders, aggregator, names, feats, attentive):
    """Embedding (and attention weights, if any) of one branch.  `feats` is a tensor when the branch has one view."""
    feats = [feats] if torch.is_tensor(feats) else feats
    views = [encoders[name](f.float()) for name, f in zip(names, feats)]
    if isinstance(aggregator, nn.Identity):
        return views[0], None
    if attentive:
        return aggregator(views)
    return aggregator(views), None


def masked_mse(model, pred, batch_data):
    """Keeps the valid pairs of a single-score batch in place (scores and targets) and returns their MSE, or a zero
    when the batch has none (baseline.py:161-172)."""
    m = batch_data.dti_masks
    if not bool(m.any()):
        return None
    pred.dti_scores = pred.dti_scores[m]
    batch_data.dti_targets = batch_data.dti_targets[m]
    return F.mse_loss(pred.dti_scores, batch_data.dti_targets)


def multi_score_loss(model, pred, batch_data, dti_weights):
    """BCE-with-logits on the binary score + MSE on the valid entries of the three real-valued scores, weighted
    (multi_output.py:222-243).  Masks scores and targets in place, returns (total, components)."""
    zero = torch.tensor(0.0, device=model.device)
    comps = {"Y": F.binary_cross_entropy_with_logits(pred.dti_scores["Y"], batch_data.dti_targets["Y"])}
    for name in DTI_SCORES[1:]:
        m = batch_data.dti_masks[name]
        pred.dti_scores[name], batch_data.dti_targets[name] = pred.dti_scores[name][m], batch_data.dti_targets[name][m]
        comps[name] = F.mse_loss(pred.dti_scores[name], batch_data.dti_targets[name]) if bool(m.any()) else zero
    return sum(w * c for w, c in zip(dti_weights, comps.values())), comps


class _SingleLossSteps:
    """training / validation / test steps of the variants whose loss is the accuracy term alone."""

    def _step(self, batch, split, metrics):
        bd, emb, pred, loss = self._common_step(batch)
        if metrics is not None and pred.dti_scores is not None and (isinstance(pred.dti_scores, dict) or len(pred.dti_scores) > 0):
            metrics.update(pred.dti_scores, bd.dti_targets)
        for name, value in (loss.components or {}).items():
            self.log(f"{split}/loss_{name}", value)
        self.log(f"{split}/loss", loss.accuracy)
        return emb, loss.accuracy

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train", self.train_metrics)[1]

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val", self.val_metrics)[1]

    def test_step(self, batch, batch_idx):
        emb, loss = self._step(batch, "test", self.test_metrics)
        self._track_attention(emb)
        return loss

    def _track_attention(self, emb):
        pass


class _AttentionTracking:
    """Mean attention weight per view over the test epoch (multi_modal.py:281-341, multi_hybrid.py:416-480)."""

    def _init_attention_tracking(self):
        if self.attentive:
            self.drug_attention_accumulator, self.target_attention_accumulator, self.test_batch_count = {}, {}, 0

    def _track_attention(self, emb):
        if not self.attentive:
            return
        self.test_batch_count += 1
        for att, names, acc in ((emb.drug_attention, self.hparams.drug_features, self.drug_attention_accumulator),
                                (emb.target_attention, self.hparams.target_features, self.target_attention_accumulator)):
            if att is not None:
                w = att.squeeze(1).mean(0).tolist()                  # [n_views]
                for name, v in zip(names, w):
                    acc.setdefault(name, []).append(v)

    def on_test_epoch_end(self):
        super().on_test_epoch_end()
        if self.attentive and getattr(self, "test_batch_count", 0) > 0:
            for side, acc in (("drug", self.drug_attention_accumulator), ("target", self.target_attention_accumulator)):
                for name, vals in acc.items():
                    self.log(f"test/{side}_att_{name}", sum(vals) / len(vals))
            self.drug_attention_accumulator, self.target_attention_accumulator, self.test_batch_count = {}, {}, 0


class BaselineDTIModel(_SingleLossSteps, AbstractDTIModel):
    def __init__(self, phase, finetune_score, learning_rate, weight_decay, scheduler, drug_features, target_features,
                 embedding_dim, hidden_dim, factor, n_layers, activation, dropout, bias, encoder_type):
        super().__init__()
        self.save_hyperparameters()
        self.model_dtype = torch.float32
        self.phase, self.finetune_score = phase, finetune_score
        activation = self.parse_activation(activation)
        assert phase == "finetune" and finetune_score in ["Y_pKd", "Y_KIBA"], \
            "Baseline model only supports finetune phase and Y_pKd/Y_KIBA finetune score must be specified"
        assert len(drug_features) == 1 and len(target_features) == 1, \
            "Baseline model only supports single feature for drug and target"
        Enc = _encoder_cls(encoder_type)
        kw = dict(hidden_dim=hidden_dim, output_dim=embedding_dim, n_layers=n_layers, factor=factor, dropout=dropout, bias=bias,
                  activation=activation)
        self.drug_encoder = Enc(input_dim=list(drug_features.values())[0], **kw)
        self.target_encoder = Enc(input_dim=list(target_features.values())[0], **kw)
        self.setup_metrics(phase=phase, finetune_score=finetune_score)

    def _create_batch_data(self, batch):
        bd = BatchData(raw_batch=batch)
        bd.drug_features, bd.target_features = self._get_features_from_batch(batch)
        bd.dti_targets, bd.dti_masks = self._get_targets_masks_from_batch(batch)
        return bd

    def forward(self, drug_features, target_features):
        emb = EmbeddingData(drug_embedding=self.drug_encoder(drug_features.float()),
                            target_embedding=self.target_encoder(target_features.float()))
        return emb, PredictionData(dti_scores=torch.sum(emb.drug_embedding * emb.target_embedding, dim=-1).squeeze(-1))

    def _common_step(self, batch):
        bd = self._create_batch_data(batch)
        emb, pred = self.forward(bd.drug_features, bd.target_features)
        mse = masked_mse(self, pred, bd)
        return bd, emb, pred, LossData(accuracy=torch.tensor(0.0, device=self.device) if mse is None else mse)


class MultiModalDTIModel(_AttentionTracking, _SingleLossSteps, AbstractDTIModel):
    def __init__(self, phase, finetune_score, learning_rate, weight_decay, scheduler, drug_features, target_features,
                 embedding_dim, hidden_dim, factor, n_layers, activation, dropout, bias, encoder_type, aggregator_type):
        super().__init__()
        self.save_hyperparameters()
        self.model_dtype = torch.float32
        self.phase, self.finetune_score = phase, finetune_score
        activation = self.parse_activation(activation)
        self.attentive = aggregator_type == "attentive"
        assert phase == "finetune" and finetune_score in ["Y_pKd", "Y_KIBA"], \
            "Multi-modal model only supports finetune phase with Y_pKd or Y_KIBA score"
        args = (encoder_type, aggregator_type, embedding_dim, hidden_dim, n_layers, factor, activation, dropout, bias)
        self.drug_encoders, self.drug_aggregator = _view_branch(drug_features, *args)
        self.target_encoders, self.target_aggregator = _view_branch(target_features, *args)
        self.setup_metrics(phase=phase, finetune_score=finetune_score)
        self._init_attention_tracking()

    _create_batch_data = BaselineDTIModel._create_batch_data

    def forward(self, drug_features, target_features):
        emb = EmbeddingData()
        emb.drug_embedding, emb.drug_attention = _aggregate(self.drug_encoders, self.drug_aggregator, self.hparams.drug_features,
                                                            drug_features, self.attentive)
        emb.target_embedding, emb.target_attention = _aggregate(self.target_encoders, self.target_aggregator,
                                                                self.hparams.target_features, target_features, self.attentive)
        return emb, PredictionData(dti_scores=torch.sum(emb.drug_embedding * emb.target_embedding, dim=-1).squeeze(-1))

    _common_step = BaselineDTIModel._common_step


class MultiOutputDTIModel(_SingleLossSteps, AbstractDTIModel):
    def __init__(self, phase, finetune_score, learning_rate, weight_decay, scheduler, drug_features, target_features,
                 embedding_dim, hidden_dim, factor, n_layers, activation, dropout, bias, encoder_type, dti_weights=None):
        super().__init__()
        self.save_hyperparameters()
        self.model_dtype = torch.float32
        self.phase, self.finetune_score, self.dti_weights = phase, finetune_score, dti_weights
        activation = self.parse_activation(activation)
        assert phase in ["finetune", "train"], "Multi-output model only supports finetune and train phases"
        if phase == "finetune":
            assert finetune_score is not None, "finetune_score must be specified for finetune phase"
        else:
            assert dti_weights is not None, "dti_weights must be specified for general DTI training"
        assert len(drug_features) == 1 and len(target_features) == 1, \
            "Multi-output model only supports single feature for drug and target"
        Enc = _encoder_cls(encoder_type)
        kw = dict(hidden_dim=hidden_dim, output_dim=embedding_dim, n_layers=n_layers, factor=factor, dropout=dropout, bias=bias,
                  activation=activation)
        self.drug_encoder = Enc(input_dim=list(drug_features.values())[0], **kw)
        self.target_encoder = Enc(input_dim=list(target_features.values())[0], **kw)
        self.fusion = CrossAttentionFusion(input_dim=embedding_dim, output_dim=hidden_dim, n_layers=n_layers, factor=factor,
                                           dropout=dropout, bias=bias, activation=activation)
        self.dti_head = DTIHead(input_dim=hidden_dim, proj_dim=hidden_dim // 2, dropout=dropout, bias=bias, activation=activation)
        self.setup_metrics(phase=phase, finetune_score=finetune_score)

    _create_batch_data = BaselineDTIModel._create_batch_data

    def forward(self, drug_features, target_features):
        emb = EmbeddingData(drug_embedding=self.drug_encoder(drug_features.float()),
                            target_embedding=self.target_encoder(target_features.float()))
        emb.fused_embedding, _ = self.fusion(emb.drug_embedding, emb.target_embedding)
        scores = self.dti_head(emb.fused_embedding)
        return emb, PredictionData(dti_scores=scores[self.finetune_score] if self.phase == "finetune" else scores)

    def _common_step(self, batch):
        bd = self._create_batch_data(batch)
        emb, pred = self.forward(bd.drug_features, bd.target_features)
        if self.phase == "finetune":
            mse = masked_mse(self, pred, bd)
            return bd, emb, pred, LossData(accuracy=torch.tensor(0.0, device=self.device) if mse is None else mse)
        total, comps = multi_score_loss(self, pred, bd, self.dti_weights)
        return bd, emb, pred, LossData(accuracy=total, components=comps)


class MultiHybridDTIModel(_AttentionTracking, AbstractDTIModel):
    def __init__(self, phase, finetune_score, learning_rate, weight_decay, scheduler, weights, dti_weights, contrastive_temp,
                 drug_features, target_features, embedding_dim, hidden_dim, factor, n_layers, activation, dropout, bias,
                 encoder_type, aggregator_type):
        super().__init__()
        self.save_hyperparameters()
        self.model_dtype = torch.float32
        self.phase, self.finetune_score = phase, finetune_score
        activation = self.parse_activation(activation)
        self.attentive = aggregator_type == "attentive"
        self.weights, self.dti_weights, self.contrastive_temp = weights, dti_weights, contrastive_temp
        if phase == "finetune":
            assert finetune_score is not None, "finetune_score must be specified for finetune phase"
            assert dti_weights is None, "dti_weights must be None for finetune phase"
            assert contrastive_temp is None, "contrastive_temp must be None for finetune phase"
        elif phase == "train":
            assert dti_weights is not None, "dti_weights must be specified for general DTI training"
            assert contrastive_temp is not None, "contrastive_temp must be specified for general DTI training"
        elif phase in ("pretrain_drug", "pretrain_target"):
            assert contrastive_temp is not None, "contrastive_temp must be specified for pretrain phases"
        else:
            raise ValueError(f"Invalid phase: {phase}")
        args = (encoder_type, aggregator_type, embedding_dim, hidden_dim, n_layers, factor, activation, dropout, bias)
        head = dict(input_dim=embedding_dim, proj_dim=hidden_dim // 2, dropout=dropout, bias=bias, activation=activation)
        if phase != "pretrain_target":
            self.drug_encoders, self.drug_aggregator = _view_branch(drug_features, *args)
            self.drug_contrastive_head = InfoNCEHead(**head)
        if phase != "pretrain_drug":
            self.target_encoders, self.target_aggregator = _view_branch(target_features, *args)
            self.target_contrastive_head = InfoNCEHead(**head)
        if phase in ("train", "finetune"):
            self.fusion = CrossAttentionFusion(input_dim=embedding_dim, output_dim=hidden_dim, n_layers=n_layers, factor=factor,
                                               dropout=dropout, bias=bias, activation=activation)
            self.dti_head = DTIHead(input_dim=hidden_dim, proj_dim=hidden_dim // 2, dropout=dropout, bias=bias,
                                    activation=activation)
        self.setup_metrics(phase=phase, finetune_score=finetune_score)
        self._init_attention_tracking()

    def _create_batch_data(self, batch):
        bd = BatchData(raw_batch=batch)
        bd.drug_features, bd.target_features = self._get_features_from_batch(batch)
        bd.drug_fp, bd.target_fp = self._get_fingerprints_from_batch(batch)
        if self.phase not in ("pretrain_drug", "pretrain_target"):
            bd.dti_targets, bd.dti_masks = self._get_targets_masks_from_batch(batch)
        return bd

    def forward(self, drug_features, target_features):
        emb, pred = EmbeddingData(), PredictionData()
        if self.phase != "pretrain_target":
            emb.drug_embedding, emb.drug_attention = _aggregate(self.drug_encoders, self.drug_aggregator,
                                                                self.hparams.drug_features, drug_features, self.attentive)
        if self.phase != "pretrain_drug":
            emb.target_embedding, emb.target_attention = _aggregate(self.target_encoders, self.target_aggregator,
                                                                    self.hparams.target_features, target_features, self.attentive)
        if self.phase not in ("pretrain_drug", "pretrain_target"):
            emb.fused_embedding, _ = self.fusion(emb.drug_embedding, emb.target_embedding)
            scores = self.dti_head(emb.fused_embedding)
            pred.dti_scores = scores[self.finetune_score] if self.phase == "finetune" else scores
        return emb, pred

    def _common_step(self, batch):
        bd = self._create_batch_data(batch)
        emb, pred = self.forward(bd.drug_features, bd.target_features)
        zero = torch.tensor(0.0, device=self.device)
        loss = LossData()
        dc = (self.drug_contrastive_head(x=emb.drug_embedding, fingerprints=bd.drug_fp, temperature=self.contrastive_temp)
              if self.phase not in ("pretrain_target", "finetune") else zero)
        tc = (self.target_contrastive_head(x=emb.target_embedding, fingerprints=bd.target_fp, temperature=self.contrastive_temp)
              if self.phase not in ("pretrain_drug", "finetune") else zero)
        loss.contrastive = dc + tc
        loss.components = {"drug_contrastive": dc, "target_contrastive": tc}
        if self.phase == "finetune":
            m = bd.dti_masks
            pred.dti_scores, bd.dti_targets = pred.dti_scores[m], bd.dti_targets[m]
            loss.accuracy = F.mse_loss(pred.dti_scores, bd.dti_targets) if bool(m.any()) else zero
        elif self.phase == "train":
            loss.accuracy, comps = multi_score_loss(self, pred, bd, self.dti_weights)
            loss.components.update(comps)
        return bd, emb, pred, loss

    def _step(self, batch, split, metrics):
        bd, emb, pred, loss_data = self._common_step(batch)
        if self.phase not in ("pretrain_drug", "pretrain_target"):
            if metrics is not None and pred.dti_scores is not None:
                metrics.update(pred.dti_scores, bd.dti_targets)
            for name, value in loss_data.components.items():
                self.log(f"{split}/loss_{name}", value)
        total = loss_data.compute_loss(self.weights)
        if loss_data.contrastive is not None:
            self.log(f"{split}/loss_contrastive", loss_data.contrastive)
        if loss_data.accuracy is not None:
            self.log(f"{split}/loss_accuracy", loss_data.accuracy)
        self.log(f"{split}/loss", total)
        return emb, total

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train", self.train_metrics)[1]

    def validation_step(self, batch, batch_idx):
        return self._step(batch, "val", self.val_metrics)[1]

    def test_step(self, batch, batch_idx):
        emb, total = self._step(batch, "test", self.test_metrics)
        self._track_attention(emb)
        return total
