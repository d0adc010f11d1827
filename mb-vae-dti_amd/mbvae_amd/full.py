"""FullDTIModel: multi-branch variational encoders + DTI head + discrete-diffusion graph decoder, on the HIP path.

Drop-in for mb_vae_dti/training/modules/full.py:40-1584 (and the AbstractDTIModel helpers of
mb_vae_dti/training/modules/utils.py:28-451 it relies on): same constructor arguments, attribute / state_dict names,
batch dictionary, method names and return values.  It subclasses pytorch_lightning.LightningModule when that package
is importable and a small nn.Module stand-in otherwise.  The reference's host-synchronising asserts are dropped;
RDKit molecule building is used only if RDKit is importable (sample_batch then returns Mol objects, otherwise the
(atom classes, bond classes) tensors).
"""
import logging

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .abstract import ACTIVATIONS, AbstractDTIModel, encode_no_edge, to_dense  # noqa: F401  (re-exported)
from .blocks import AttentiveAggregator, ConcatAggregator, CrossAttentionFusion, ResidualEncoder, TransformerEncoder
from .containers import BatchData, EmbeddingData, GraphData, LossData, PlaceHolder, PredictionData
from .diffusion import (DistributionNodes, MarginalUniformTransition, PredefinedNoiseScheduleDiscrete, exp_draws,
                        sample_discrete_feature_noise)
from .features import ExtraFeatures, ExtraMolecularFeatures
from .graph_transformer import GraphTransformer
from .heads import DTIHead, InfoNCEHead, KLVariationalHead, ReconstructionHead

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------ running metrics
class _RunningMean(nn.Module):
    """sum / count accumulator with the update signatures of the reference's torchmetrics classes
    (mb_vae_dti/training/metrics/graph_metrics.py:13-65); forward() updates and returns the batch value."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.total, self.count = 0.0, 0.0

    def reset(self):
        self.total, self.count = 0.0, 0.0

    def forward(self, a, b=None):
        if self.kind == "nll":
            val, cnt = a.sum(), a.numel()
        elif self.kind == "kl":                               # (p, log q)
            val, cnt = F.kl_div(b, a, reduction="sum"), a.size(0)
        else:                                                 # sum over everything but the batch dim
            val, cnt = a.sum(), a.size(0)
        self.total = self.total + val.detach()
        self.count = self.count + cnt
        return val / cnt

    def compute(self):
        return self.total / max(self.count, 1)


class TrainMolecularMetricsDiscrete(nn.Module):
    """Per-class BCE-with-logits trackers over atom / bond logits (molecular_metrics.py:68-229), fed from the
    statistics the masked-CE kernel already produced (no extra pass over the logits)."""

    def __init__(self, atom_types, prefix="train/"):
        super().__init__()
        self.atom_types, self.prefix = list(atom_types), prefix
        self.reset()

    def reset(self):
        self.atom_sum = self.bond_sum = None
        self.atom_n = self.bond_n = 0.0

    def update_from_stats(self, stats_X, stats_E):
        cx = len(self.atom_types)
        ax, ae = stats_X[2:2 + cx].detach(), stats_E[2:7].detach()
        self.atom_sum = ax if self.atom_sum is None else self.atom_sum + ax
        self.bond_sum = ae if self.bond_sum is None else self.bond_sum + ae
        self.atom_n = self.atom_n + stats_X[1].detach()
        self.bond_n = self.bond_n + stats_E[1].detach()

    def compute(self):
        out = {}
        if self.atom_sum is not None:
            for i, a in enumerate(self.atom_types):
                out[f"{self.prefix}atom_{a}CE"] = self.atom_sum[i] / self.atom_n
            for i, b in enumerate(("NoBond", "Single", "Double", "Triple", "Aromatic")):
                out[f"{self.prefix}bond_{b}CE"] = self.bond_sum[i] / self.bond_n
        return out


def _mask_distributions(true_X, true_E, pred_X, pred_E, node_mask):
    """Masked rows become the point mass on class 0, then everything is renormalised with +1e-7
    (mb_vae_dti/training/diffusion/utils.py:200-249)."""
    n = node_mask.shape[1]
    offdiag = ~torch.eye(n, dtype=torch.bool, device=node_mask.device).unsqueeze(0)
    bad_x = ~node_mask
    bad_e = ~(node_mask.unsqueeze(1) & node_mask.unsqueeze(2) & offdiag)

    def fix(t, bad):
        row = torch.zeros(t.shape[-1], device=t.device, dtype=t.dtype)
        row[0] = 1.0
        t = torch.where(bad.unsqueeze(-1), row, t) + 1e-7
        return t / t.sum(-1, keepdim=True)

    return fix(true_X, bad_x), fix(true_E, bad_e), fix(pred_X, bad_x), fix(pred_E, bad_e)


def _posterior(M, M_t, Qt, Qsb, Qtb):
    """(z_t Qt^T * x Qsb) / (x Qtb z_t^T), zero denominators -> 1e-6 (diffusion/utils.py:258-280)."""
    M, M_t = M.flatten(1, -2).float(), M_t.flatten(1, -2).float()
    num = (M_t @ Qt.transpose(-2, -1)) * (M @ Qsb)
    den = ((M @ Qtb) * M_t).sum(-1)
    den = torch.where(den == 0, torch.full_like(den, 1e-6), den)
    return num / den.unsqueeze(-1)


class FullDTIModel(AbstractDTIModel):
    def __init__(self, phase, finetune_score, learning_rate, weight_decay, scheduler, weights, dti_weights, diff_weights,
                 contrastive_temp, drug_features, target_features, embedding_dim, hidden_dim, factor, n_layers, activation,
                 dropout, bias, encoder_type, aggregator_type, diffusion_steps, sample_every_val, val_samples_per_embedding,
                 test_samples_per_embedding, graph_transformer_kwargs, dataset_infos):
        super().__init__()
        self.save_hyperparameters()
        self.model_dtype = torch.float32
        self.phase, self.finetune_score = phase, finetune_score
        activation = self.parse_activation(activation)
        self.attentive = aggregator_type == "attentive"
        self.weights, self.dti_weights, self.diff_weights = weights, dti_weights, diff_weights
        self.contrastive_temp = contrastive_temp
        # Molecule-quality trackers of sampled graphs (RDKit validity / match metrics in the reference): slots for
        # caller-supplied objects with update(generated_mols=, target_smiles=) / compute() / reset()
        self.val_mol = self.test_mol = None
        # Optional dict of pre-drawn random tensors {"t_int", "expX", "expE", "lcc_noise", "eps", "eigvec"} consumed
        # instead of the device generator: lets a step be replayed on exactly the draws (and the eigenvector gauge)
        # another implementation made (parity tests).  A value may be a list: the calls then consume it front to back
        # (a validation step draws its Exp(1) / lcc tensors twice, a sampling run once per reverse step).
        self.replay = None

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
        if phase != "pretrain_target":
            for name, val in (("diff_weights", diff_weights), ("diffusion_steps", diffusion_steps),
                              ("graph_transformer_kwargs", graph_transformer_kwargs), ("dataset_infos", dataset_infos),
                              ("sample_every_val", sample_every_val), ("val_samples_per_embedding", val_samples_per_embedding),
                              ("test_samples_per_embedding", test_samples_per_embedding)):
                assert val is not None, f"{name} must be specified for instantiating the drug branch"
            assert graph_transformer_kwargs["output_dims"]["y"] == embedding_dim, \
                "embedding_dim must match y_dim of graph_transformer_kwargs"

        Encoder = ResidualEncoder if encoder_type == "resnet" else TransformerEncoder
        Aggregator = ConcatAggregator if aggregator_type == "concat" else AttentiveAggregator

        def branch(features):
            encoders = nn.ModuleDict({
                name: Encoder(input_dim=dim, hidden_dim=hidden_dim, output_dim=hidden_dim if len(features) > 1 else embedding_dim,
                              n_layers=n_layers, factor=factor, activation=activation, dropout=dropout, bias=bias)
                for name, dim in features.items()})
            agg = (Aggregator(input_dim=hidden_dim, n_features=len(features), output_dim=embedding_dim)
                   if len(features) > 1 else nn.Identity())
            head = InfoNCEHead(input_dim=embedding_dim, proj_dim=hidden_dim // 2, dropout=dropout, bias=bias, activation=activation)
            return encoders, agg, head

        if phase != "pretrain_target":
            self.drug_encoders, self.drug_aggregator, self.drug_contrastive_head = branch(drug_features)
            self.drug_kl_head = KLVariationalHead(input_dim=embedding_dim, proj_dim=embedding_dim)
            ds, gen = dataset_infos["dataset"], dataset_infos["general"]
            self.T = diffusion_steps
            self.sample_every_val = sample_every_val
            self.val_samples_per_embedding = val_samples_per_embedding
            self.test_samples_per_embedding = test_samples_per_embedding
            self.reconstruction_head = ReconstructionHead(diff_weights)
            self.atom_types = list(gen["atom_types"])
            self.nodes_dist = DistributionNodes(ds["node_count_distribution"])
            ydim = graph_transformer_kwargs["output_dims"]["y"]
            self.limit_dist = PlaceHolder(X=torch.tensor(ds["node_marginals"]), E=torch.tensor(ds["edge_marginals"]),
                                          y=torch.ones(ydim) / ydim)
            self.noise_schedule = PredefinedNoiseScheduleDiscrete(timesteps=diffusion_steps)
            self.transition_model = MarginalUniformTransition(x_marginals=torch.tensor(ds["node_marginals"]),
                                                              e_marginals=torch.tensor(ds["edge_marginals"]), y_classes=ydim)
            self.graph_structure_features = ExtraFeatures(max_n_nodes=gen["max_n_nodes"])
            self.molecular_features = ExtraMolecularFeatures(valencies=gen["atom_valencies"], max_weight=gen["max_weight"],
                                                             atom_weights=gen["atom_weights"])
            self.Xdim_output = graph_transformer_kwargs["output_dims"]["X"]
            self.Edim_output = graph_transformer_kwargs["output_dims"]["E"]
            self.drug_decoder = GraphTransformer(
                n_layers=graph_transformer_kwargs["n_layers"], input_dims=graph_transformer_kwargs["input_dims"],
                output_dims=graph_transformer_kwargs["output_dims"], hidden_mlp_dims=graph_transformer_kwargs["hidden_mlp_dims"],
                hidden_dims=graph_transformer_kwargs["hidden_dims"], act_fn_in=nn.ReLU(), act_fn_out=nn.ReLU())
        if phase != "pretrain_drug":
            self.target_encoders, self.target_aggregator, self.target_contrastive_head = branch(target_features)
        if phase in ("train", "finetune"):
            self.fusion = CrossAttentionFusion(input_dim=embedding_dim, output_dim=hidden_dim, n_layers=n_layers, factor=factor,
                                               dropout=dropout, bias=bias, activation=activation)
            self.dti_head = DTIHead(input_dim=hidden_dim, proj_dim=hidden_dim // 2, dropout=dropout, bias=bias, activation=activation)

        if phase != "pretrain_target":
            for split in ("val", "test"):
                setattr(self, f"{split}_nll", _RunningMean("nll"))
                setattr(self, f"{split}_X_kl", _RunningMean("kl"))
                setattr(self, f"{split}_E_kl", _RunningMean("kl"))
                setattr(self, f"{split}_X_logp", _RunningMean("sum"))
                setattr(self, f"{split}_E_logp", _RunningMean("sum"))
            self.train_mol = TrainMolecularMetricsDiscrete(atom_types=gen["atom_types"], prefix="train/")
        else:
            self.train_mol = None

        if phase == "finetune":
            for p in self.parameters():
                p.requires_grad = False
            for mod in (self.fusion, self.dti_head, self.drug_encoders, self.target_encoders, self.drug_aggregator,
                        self.target_aggregator, self.drug_kl_head):
                for p in mod.parameters():
                    p.requires_grad = True

    def _create_batch_data(self, batch):
        bd = BatchData(raw_batch=batch)
        bd.drug_features, bd.target_features = self._get_features_from_batch(batch)
        bd.drug_fp, bd.target_fp = self._get_fingerprints_from_batch(batch)
        if self.phase not in ("pretrain_drug", "pretrain_target"):
            bd.dti_targets, bd.dti_masks = self._get_targets_masks_from_batch(batch)
        if self.phase not in ("pretrain_target", "finetune"):
            bd.graph_data = self._extract_graph_data(batch)
            rep = batch.get("representations") if self.hparams.phase == "pretrain_drug" else batch["drug"].get("representations")
            bd.smiles = rep.get("smiles") if rep else None
        return bd

    # ------------------------------------------------------------------ encoders / DTI head
    def forward(self, drug_features, target_features):
        emb = EmbeddingData()

        def encode(encoders, aggregator, names, feats):
            feats = [feats] if torch.is_tensor(feats) else feats
            views = [encoders[name](f.float()) for name, f in zip(names, feats)]
            if len(views) == 1:
                return views[0], None
            if self.attentive:
                return aggregator(views)
            return aggregator(views), None

        if self.phase != "pretrain_target":
            e, emb.drug_attention = encode(self.drug_encoders, self.drug_aggregator, self.hparams.drug_features, drug_features)
            emb.drug_embedding, emb.drug_mu, emb.drug_logvar = self.drug_kl_head(e, self._replayed("eps", e.device))
        if self.phase != "pretrain_drug":
            emb.target_embedding, emb.target_attention = encode(self.target_encoders, self.target_aggregator,
                                                                self.hparams.target_features, target_features)
        pred = PredictionData()
        if self.phase not in ("pretrain_drug", "pretrain_target"):
            emb.fused_embedding, _ = self.fusion(emb.drug_embedding, emb.target_embedding)
            scores = self.dti_head(emb.fused_embedding)
            pred.dti_scores = scores[self.finetune_score] if self.phase == "finetune" else scores
        return emb, pred

    # ------------------------------------------------------------------ diffusion
    def _replayed(self, key, device=None):
        """Next pre-drawn tensor for `key`, or None when the step draws it itself."""
        rp = self.replay
        if not rp or key not in rp:
            return None
        v = rp[key]
        if isinstance(v, list):
            if not v:
                return None
            v = v.pop(0)
        return v if device is None or v is None else v.to(device)

    def _marginals(self, device):
        key = str(device)
        cache = self.__dict__.setdefault("_marginal_cache", {})
        if key not in cache:                       # one host->device copy per device, not one per step
            cache[key] = (self.limit_dist.X.to(device).float(), self.limit_dist.E.to(device).float())
        return cache[key]

    def apply_noise(self, X, E, node_mask):
        """t ~ U{0|1..T}, G_t ~ q(G_t | G) through the forward-noising kernel; same RNG draw order as the reference
        (timestep, then the X draws, then all n*n E draws)."""
        bs, n, cx = X.shape
        dev = X.device
        low = 0 if self.training else 1
        t_int = self._replayed("t_int", dev)
        t_int = t_int.float() if t_int is not None else torch.randint(low, self.T + 1, size=(bs, 1), device=dev).float()
        t, s = t_int / self.T, (t_int - 1) / self.T
        beta_t = self.noise_schedule(t_normalized=t)
        a_s = self.noise_schedule.get_alpha_bar(t_normalized=s)
        a_t = self.noise_schedule.get_alpha_bar(t_normalized=t)
        mx, me = self._marginals(dev)
        eX, eE = self._exp_pair(bs, n, cx, E.shape[-1], dev)
        _, _, X_t, E_t = ops.apply_noise(X.float(), E.float(), node_mask, a_t, mx, me, eX, eE)
        return X_t, E_t, {"t_int": t_int, "t": t, "beta_t": beta_t, "alpha_s_bar": a_s, "alpha_t_bar": a_t}

    def _exp_pair(self, bs, n, cx, ce, dev):
        """Exp(1) draws for one categorical sample of all nodes, then all n*n pairs (the reference's draw order)."""
        eX, eE = self._replayed("expX", dev), self._replayed("expE", dev)
        return (eX if eX is not None else exp_draws(bs * n, cx, dev)), (eE if eE is not None else exp_draws(bs * n * n, ce, dev))

    def get_extra_features(self, X_t, E_t, noise_params, node_mask):
        Xg, yg = self.graph_structure_features(E_t, node_mask, lcc_noise=self._replayed("lcc_noise", E_t.device),
                                               eigvec=self._replayed("eigvec", E_t.device))
        # apply_noise's output is an int64 one-hot in the reference (t_int marks that path): see ExtraMolecularFeatures
        Xm, ym = self.molecular_features(X_t, E_t, truncate_weight="t_int" in noise_params)
        return torch.cat((Xg, Xm), dim=2), torch.cat((yg, ym, noise_params["t"]), dim=1)

    def denoise_drug_graph(self, embedding_data, batch_data):
        if self.phase in ("pretrain_target", "finetune"):
            return None
        gd = batch_data.graph_data
        return self.drug_decoder(X=torch.cat([gd.X_t, gd.X_extra], dim=2).float(), E=gd.E_t.float(),
                                 y=torch.cat([embedding_data.drug_embedding, gd.y_extra], dim=1).float(), node_mask=gd.node_mask)

    # ------------------------------------------------------------------ ELBO (validation / test)
    def kl_prior(self, batch_data):
        gd = batch_data.graph_data
        if self.phase == "pretrain_target" or gd is None:
            return torch.tensor(0.0, device=self.device)
        X, E, node_mask = gd.X.float(), gd.E.float(), gd.node_mask
        bs, n = X.shape[:2]
        a_T = self.noise_schedule.get_alpha_bar(t_int=self.T * torch.ones((bs, 1), device=X.device))
        Qtb = self.transition_model.get_Qt_bar(a_T, X.device)
        probX, probE = X @ Qtb.X, E @ Qtb.E.unsqueeze(1)
        mx, me = self._marginals(X.device)
        lX, lE, pX, pE = _mask_distributions(mx.expand(bs, n, -1), me.expand(bs, n, n, -1), probX, probE, node_mask)
        kx = F.kl_div(pX.log(), lX, reduction="none").reshape(bs, -1).sum(-1)
        ke = F.kl_div(pE.log(), lE, reduction="none").reshape(bs, -1).sum(-1)
        return kx + ke

    def compute_Lt(self, batch_data, prediction_data, test=False):
        gd = batch_data.graph_data
        if self.phase == "pretrain_target" or gd is None:
            return torch.tensor(0.0, device=self.device)
        X, E, node_mask, npar = gd.X.float(), gd.E.float(), gd.node_mask, gd.noise_params
        pred = prediction_data.graph_reconstruction
        bs, n = X.shape[:2]
        dev = X.device
        Qtb = self.transition_model.get_Qt_bar(npar["alpha_t_bar"], dev)
        Qsb = self.transition_model.get_Qt_bar(npar["alpha_s_bar"], dev)
        Qt = self.transition_model.get_Qt(npar["beta_t"], dev)
        pX, pE = F.softmax(pred.X, -1), F.softmax(pred.E, -1)
        tX = _posterior(X, gd.X_t, Qt.X, Qsb.X, Qtb.X)
        tE = _posterior(E, gd.E_t, Qt.E, Qsb.E, Qtb.E).reshape(bs, n, n, -1)
        qX = _posterior(pX, gd.X_t, Qt.X, Qsb.X, Qtb.X)
        qE = _posterior(pE, gd.E_t, Qt.E, Qsb.E, Qtb.E).reshape(bs, n, n, -1)
        tX, tE, qX, qE = _mask_distributions(tX, tE, qX, qE, node_mask)
        split = "test" if test else "val"
        kl_x = getattr(self, f"{split}_X_kl")(tX, torch.log(torch.clamp(qX, min=1e-8)))
        kl_e = getattr(self, f"{split}_E_kl")(tE, torch.log(torch.clamp(qE, min=1e-8)))
        return self.T * (kl_x + kl_e)

    def reconstruction_logp(self, batch_data, embedding_data, test=False):
        gd = batch_data.graph_data
        if self.phase == "pretrain_target" or gd is None:
            return torch.tensor(0.0, device=self.device)
        X, E, node_mask = gd.X.float(), gd.E.float(), gd.node_mask
        bs, n, cx = X.shape
        dev = X.device
        t0 = torch.zeros_like(gd.noise_params["t"])
        # q(G_0 | G) uses Q_0 = (1 - beta_0) I + beta_0 1 m^T, i.e. the forward kernel with "abar" = 1 - beta_0
        beta_0 = self.noise_schedule(t_normalized=t0)
        mx, me = self._marginals(dev)
        eX, eE = self._exp_pair(bs, n, cx, E.shape[-1], dev)
        Xc, Ec, _, _ = ops.apply_noise(X, E, node_mask, 1 - beta_0, mx, me, eX, eE)
        # the reference one-hot-encodes this sample WITHOUT masking it: padded nodes / pairs carry a sampled class
        X0, E0 = F.one_hot(Xc, cx).float(), F.one_hot(Ec, E.shape[-1]).float()
        tmp = GraphData(X_t=X0, E_t=E0, node_mask=node_mask, noise_params={"t": t0})
        tmp.X_extra, tmp.y_extra = self.get_extra_features(X0, E0, {"t": t0}, node_mask)
        pred0 = self.denoise_drug_graph(embedding_data, BatchData(raw_batch=batch_data.raw_batch, graph_data=tmp))
        probX0, probE0 = F.softmax(pred0.X, -1), F.softmax(pred0.E, -1)
        pair = node_mask.unsqueeze(1) & node_mask.unsqueeze(2)
        offdiag = ~torch.eye(n, dtype=torch.bool, device=dev).unsqueeze(0)
        probX0 = torch.where(node_mask.unsqueeze(-1), probX0, torch.ones_like(probX0))
        probE0 = torch.where((pair & offdiag).unsqueeze(-1), probE0, torch.ones_like(probE0))
        return probX0, probE0

    def compute_val_loss(self, batch_data, embedding_data, prediction_data, test=False):
        gd = batch_data.graph_data
        if self.phase == "pretrain_target" or gd is None:
            return torch.tensor(0.0, device=self.device)
        log_pN = self.nodes_dist.log_prob(gd.node_mask.sum(1).long())
        kl_prior = self.kl_prior(batch_data)
        loss_all_t = self.compute_Lt(batch_data, prediction_data, test)
        probX0, probE0 = self.reconstruction_logp(batch_data, embedding_data, test)
        loss_term_0 = (self.val_X_logp(gd.X * torch.clamp(probX0, min=1e-8).log())
                       + self.val_E_logp(gd.E * torch.clamp(probE0, min=1e-8).log()))
        nlls = -log_pN + kl_prior + loss_all_t - loss_term_0
        nll = (self.test_nll if test else self.val_nll)(nlls)
        return nll, {"kl prior": kl_prior.mean(), "Estimator loss terms": loss_all_t.mean(), "log_pn": log_pN.mean(),
                     "loss_term_0": loss_term_0, "batch_test_nll" if test else "val_nll": nll}

    # ------------------------------------------------------------------ steps
    def _common_step(self, batch, step):
        bd = self._create_batch_data(batch)
        decode = self.phase != "pretrain_target" and bd.graph_data is not None
        if decode:
            gd = bd.graph_data
            gd.X_t, gd.E_t, gd.noise_params = self.apply_noise(gd.X, gd.E, gd.node_mask)
            gd.X_extra, gd.y_extra = self.get_extra_features(gd.X_t, gd.E_t, gd.noise_params, gd.node_mask)
        emb, pred = self.forward(bd.drug_features, bd.target_features)
        if decode:
            pred.graph_reconstruction = self.denoise_drug_graph(emb, bd)
        zero = torch.tensor(0.0, device=self.device)
        loss = LossData()
        use_drug = self.phase not in ("pretrain_target", "finetune")
        loss.complexity = self.drug_kl_head.kl_divergence(emb.drug_mu, emb.drug_logvar) if use_drug else zero
        dc = (self.drug_contrastive_head(x=emb.drug_embedding, fingerprints=bd.drug_fp, temperature=self.contrastive_temp)
              if use_drug else zero)
        tc = (self.target_contrastive_head(x=emb.target_embedding, fingerprints=bd.target_fp, temperature=self.contrastive_temp)
              if self.phase not in ("pretrain_drug", "finetune") else zero)
        loss.contrastive = dc + tc
        loss.components = {"drug_contrastive": dc, "target_contrastive": tc}
        if decode:
            if step == "train":
                g = pred.graph_reconstruction
                loss.reconstruction = self.reconstruction_head(masked_pred_X=g.X, masked_pred_E=g.E, true_X=bd.graph_data.X,
                                                               true_E=bd.graph_data.E)
            else:
                loss.reconstruction, parts = self.compute_val_loss(bd, emb, pred, test=step == "test")
                loss.components.update(parts)
        else:
            loss.reconstruction = zero
        if self.phase == "finetune":
            m = bd.dti_masks
            pred.dti_scores, bd.dti_targets = pred.dti_scores[m], bd.dti_targets[m]
            loss.accuracy = F.mse_loss(pred.dti_scores, bd.dti_targets) if m.any() else zero
        elif self.phase == "train":
            comps = {"Y": F.binary_cross_entropy_with_logits(pred.dti_scores["Y"], bd.dti_targets["Y"])}
            for name in ("Y_pKd", "Y_pKi", "Y_KIBA"):
                m = bd.dti_masks[name]
                pred.dti_scores[name], bd.dti_targets[name] = pred.dti_scores[name][m], bd.dti_targets[name][m]
                comps[name] = F.mse_loss(pred.dti_scores[name], bd.dti_targets[name]) if m.any() else zero
            loss.accuracy = sum(w * c for w, c in zip(self.dti_weights, comps.values()))
            loss.components.update(comps)
        return bd, emb, pred, loss

    def _log_losses(self, split, loss_data, total):
        for name, value in loss_data.components.items():
            self.log(f"{split}/loss_{name}", value)
        for name in ("accuracy", "complexity", "contrastive", "reconstruction"):
            v = getattr(loss_data, name)
            if v is not None:
                self.log(f"{split}/loss_{name}", v)
        self.log(f"{split}/loss", total)

    def training_step(self, batch, batch_idx):
        bd, emb, pred, loss_data = self._common_step(batch, step="train")
        if self.train_mol is not None and bd.graph_data is not None and self.reconstruction_head.last_stats is not None:
            self.train_mol.update_from_stats(*self.reconstruction_head.last_stats)
        total = loss_data.compute_loss(self.weights)
        self._log_losses("train", loss_data, total)
        return total

    def validation_step(self, batch, batch_idx):
        bd, emb, pred, loss_data = self._common_step(batch, step="val")
        if self.phase not in ("pretrain_target", "finetune") and bd.graph_data is not None:
            if not hasattr(self, "_val_embeddings_buffer"):
                self._val_embeddings_buffer, self._val_smiles_buffer = [], []
            if len(self._val_embeddings_buffer) < 5:
                self._val_embeddings_buffer.append(emb.drug_embedding.detach())
                self._val_smiles_buffer.extend(bd.smiles or [])
        total = loss_data.compute_loss(self.weights)
        self._log_losses("val", loss_data, total)
        return total

    def test_step(self, batch, batch_idx):
        bd, emb, pred, loss_data = self._common_step(batch, step="test")
        if self.phase not in ("pretrain_target", "finetune") and bd.graph_data is not None:
            if not hasattr(self, "_test_embeddings_buffer"):
                self._test_embeddings_buffer, self._test_smiles_buffer = [], []
            if len(self._test_embeddings_buffer) < 10:
                self._test_embeddings_buffer.append(emb.drug_embedding.detach())
                self._test_smiles_buffer.extend(bd.smiles or [])
        total = loss_data.compute_loss(self.weights)
        self._log_losses("test", loss_data, total)
        return total

    # ------------------------------------------------------------------ epoch ends (full.py:1298-1485 of the reference)
    def on_train_epoch_end(self):
        super().on_train_epoch_end()
        if self.phase not in ("pretrain_target", "finetune"):
            for key, value in self.train_mol.compute().items():
                self.log(key, value)
            self.train_mol.reset()

    def _sample_buffered(self, split, samples_per_embedding):
        """Conditional sampling from the drug embeddings buffered during the epoch (at most 5 validation / 10 test
        batches); feeds the optional molecule tracker with the generated groups and the matching SMILES slice.
        Returns the generated groups per buffered batch and clears the buffers."""
        emb_buf = getattr(self, f"_{split}_embeddings_buffer", None)
        if not emb_buf:
            return []
        smiles = getattr(self, f"_{split}_smiles_buffer")
        tracker = getattr(self, f"{split}_mol")
        out = []
        for i, emb in enumerate(emb_buf):
            mols = self.sample_batch(drug_embeddings=emb, num_samples_per_embedding=samples_per_embedding)
            out.append(mols)
            if tracker is not None:
                bs = emb.size(0)
                tracker.update(generated_mols=mols, target_smiles=smiles[i * bs:(i + 1) * bs])
        delattr(self, f"_{split}_embeddings_buffer")
        delattr(self, f"_{split}_smiles_buffer")
        return out

    def _flush_elbo(self, split):
        tracker = getattr(self, f"{split}_mol")
        if tracker is not None:
            for key, value in tracker.compute().items():
                self.log(key, value)
            tracker.reset()
        self.log(f"{split}/epoch_NLL", getattr(self, f"{split}_nll").compute())
        self.log(f"{split}/X_kl", getattr(self, f"{split}_X_kl").compute() * self.T)
        self.log(f"{split}/E_kl", getattr(self, f"{split}_E_kl").compute() * self.T)
        self.log(f"{split}/X_logp", getattr(self, f"{split}_X_logp").compute())
        self.log(f"{split}/E_logp", getattr(self, f"{split}_E_logp").compute())
        for name in ("nll", "X_kl", "E_kl", "X_logp", "E_logp"):
            getattr(self, f"{split}_{name}").reset()

    def on_validation_epoch_end(self):
        super().on_validation_epoch_end()
        if self.phase in ("pretrain_target", "finetune"):
            return
        self.last_val_samples = None
        if getattr(self, "current_epoch", 0) % self.sample_every_val == 0 and getattr(self, "_val_embeddings_buffer", None):
            logger.info("Performing validation sampling at epoch %s", self.current_epoch)
            self.last_val_samples = self._sample_buffered("val", self.val_samples_per_embedding)
        self._flush_elbo("val")

    def on_test_epoch_end(self):
        super().on_test_epoch_end()
        if self.phase in ("pretrain_target", "finetune"):
            return
        self.last_test_samples = self._sample_buffered("test", self.test_samples_per_embedding) or None
        self._flush_elbo("test")

    # ------------------------------------------------------------------ sampling
    def sample_p_zs_given_zt(self, s, t, X_t, E_t, drug_embeddings, node_mask):
        """One reverse step G_t -> G_s conditioned on the drug embeddings; returns a masked one-hot PlaceHolder."""
        if self.phase in ("pretrain_target", "finetune"):
            return None
        bs, n, cx = X_t.shape
        dev = X_t.device
        beta_t = self.noise_schedule(t_normalized=t)
        a_s = self.noise_schedule.get_alpha_bar(t_normalized=s)
        a_t = self.noise_schedule.get_alpha_bar(t_normalized=t)
        gd = GraphData(X_t=X_t, E_t=E_t, node_mask=node_mask, noise_params={"t": t})
        gd.X_extra, gd.y_extra = self.get_extra_features(X_t, E_t, {"t": t}, node_mask)
        pred = self.denoise_drug_graph(EmbeddingData(drug_embedding=drug_embeddings), BatchData(graph_data=gd))
        mx, me = self._marginals(dev)
        eX, eE = self._exp_pair(bs, n, cx, E_t.shape[-1], dev)
        _, _, X_s, E_s = ops.reverse_sample(pred.X, pred.E, X_t.float(), E_t.float(), node_mask, beta_t, a_s, a_t, mx, me, eX, eE)
        return PlaceHolder(X=X_s, E=E_s, y=torch.zeros(bs, 0, device=dev)).type_as(drug_embeddings)

    @torch.no_grad()
    def _reverse_diffusion(self, drug_embeddings, n_nodes, use_graph=False):
        """T reverse steps from the limit distribution for one dense batch; returns collapsed class ids on the host.

        use_graph: capture ONE reverse step (extra features, denoiser forward, posterior + draw; ~350 launches) in a
        HIP graph and replay it for the remaining T - 1 steps: the step is launch bound for typical molecule sizes.
        Step times are fed through static device buffers; random draws come from torch's graph-aware generator."""
        dev = drug_embeddings.device
        bs = drug_embeddings.size(0)
        n_max = int(n_nodes.max().item())
        node_mask = torch.arange(n_max, device=dev).unsqueeze(0) < n_nodes.unsqueeze(1)
        lim = (self._replayed("expX", dev), self._replayed("expE", dev))
        zT = sample_discrete_feature_noise(self.limit_dist, node_mask, exp=lim if lim[0] is not None else None)
        X, E = zT.X, zT.E
        trace = (self.replay or {}).get("trajectory")      # optional list: collapsed class ids after every reverse step
        ones = torch.ones((bs, 1), dtype=self.model_dtype, device=dev)
        steps = list(reversed(range(self.T)))
        self.drug_decoder.prepare_padding_shortcuts(node_mask)     # the same mask for all T steps (and before any capture)
        if use_graph and len(steps) > 2:
            emb = drug_embeddings.contiguous()
            # the first step runs eagerly: it also creates every lazily built buffer (workspace, cached constants)
            z = self.sample_p_zs_given_zt(steps[0] * ones / self.T, (steps[0] + 1) * ones / self.T, X, E, emb, node_mask)
            Xb, Eb = z.X.contiguous().clone(), z.E.contiguous().clone()
            sb, tb = torch.empty_like(ones), torch.empty_like(ones)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                z = self.sample_p_zs_given_zt(sb, tb, Xb, Eb, emb, node_mask)
                Xb.copy_(z.X)
                Eb.copy_(z.E)
            for s_int in steps[1:]:
                sb.copy_(s_int * ones / self.T)
                tb.copy_((s_int + 1) * ones / self.T)
                graph.replay()
            X, E = Xb, Eb
        else:
            for s_int in steps:
                z = self.sample_p_zs_given_zt(s_int * ones / self.T, (s_int + 1) * ones / self.T, X, E, drug_embeddings, node_mask)
                X, E = z.X, z.E
                if trace is not None:
                    c = PlaceHolder(X=X.clone(), E=E.clone()).mask(node_mask, collapse=True)
                    trace.append((c.X.cpu(), c.E.cpu()))
        final = PlaceHolder(X=X, E=E).mask(node_mask, collapse=True)
        Xc, Ec = final.X.cpu(), final.E.cpu()
        return [(Xc[i, :k], Ec[i, :k, :k]) for i, k in enumerate(n_nodes.cpu().tolist())]

    @torch.no_grad()
    def sample_batch(self, drug_embeddings, num_nodes=None, num_samples_per_embedding=1, return_graphs=False, size_buckets=1,
                     use_graph=False, balance=False):
        """Reverse diffusion from the limit distribution for T steps.  Returns molecules grouped per embedding
        (RDKit Mol objects when RDKit is available, else (atom ids, bond ids) tensor pairs).

        size_buckets > 1 sorts the molecules by size and runs that many equal-count sub-batches, each padded only to
        its own largest molecule (the dense n x n edge tensors make the cost grow with n_max^2); every molecule is
        still sampled independently from the same distribution, only the order of the random draws differs.  Inside
        a sub-batch the molecules are laid out largest first, back and forth over blocks of one molecule per compute
        unit: the attention kernel hands graph g to workgroup g mod #CUs, and this order gives every workgroup a
        similar amount of work.  The default (1) is the reference's single dense batch in the caller's order;
        balance=True applies the size-ordered layout to that single batch too.
        use_graph replays one captured reverse step as a HIP graph (see _reverse_diffusion)."""
        if self.phase in ("pretrain_target", "finetune"):
            return []
        dev = drug_embeddings.device
        n0 = drug_embeddings.size(0)
        n_nodes = self.nodes_dist.sample_n(n0, dev) if num_nodes is None else num_nodes.to(dev)
        if num_samples_per_embedding > 1:
            drug_embeddings = drug_embeddings.repeat_interleave(num_samples_per_embedding, dim=0)
            n_nodes = n_nodes.repeat_interleave(num_samples_per_embedding)
        bs = n0 * num_samples_per_embedding
        if bs < 2 * size_buckets:
            size_buckets = 1
        if size_buckets <= 1 and not balance:
            graphs = self._reverse_diffusion(drug_embeddings, n_nodes, use_graph)
        else:
            order = torch.argsort(n_nodes, descending=True, stable=True)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            graphs = [None] * bs
            for part in torch.tensor_split(order, max(size_buckets, 1)):
                if part.numel() == 0:
                    continue
                blocks = list(torch.split(part, cus))
                part = torch.cat([blk.flip(0) if k % 2 else blk for k, blk in enumerate(blocks)])
                for i, gph in zip(part.tolist(), self._reverse_diffusion(drug_embeddings[part], n_nodes[part], use_graph)):
                    graphs[i] = gph
        mols = graphs if return_graphs else [self._graph_to_mol(a, b) for a, b in graphs]
        k = num_samples_per_embedding
        return [mols[i * k:(i + 1) * k] for i in range(n0)]

    def _graph_to_mol(self, atoms, bonds):
        try:
            from rdkit import Chem
        except Exception:  # noqa: BLE001  (RDKit not installed: hand back the class-id tensors)
            return atoms, bonds
        try:
            mol = Chem.RWMol()
            idx = {}
            for i, a in enumerate(atoms.tolist()):
                if a >= 0:
                    idx[i] = mol.AddAtom(Chem.Atom(self.atom_types[a]))
            kinds = {1: Chem.rdchem.BondType.SINGLE, 2: Chem.rdchem.BondType.DOUBLE, 3: Chem.rdchem.BondType.TRIPLE,
                     4: Chem.rdchem.BondType.AROMATIC}
            b = bonds.tolist()
            for i in range(len(b)):
                for j in range(i + 1, len(b)):
                    if b[i][j] in kinds:
                        mol.AddBond(idx[i], idx[j], kinds[b[i][j]])
            return mol.GetMol()
        except Exception:  # noqa: BLE001
            return None
