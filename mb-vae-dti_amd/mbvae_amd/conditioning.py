"""Target-conditioned generation on the device: the consumer of the reverse-diffusion sampler.

Mirrors the numerical steps of mb_vae_dti/validating/conditioning.py `ConditionalDrugGenerator`
(process_targets :104-133, sample_initial_latents :135-227 / _sample_from_learned_distribution :229-290,
optimize_latents_for_interaction :301-418, generate_molecules_batch :420-525) with the same argument meanings and
return values.  What differs is the schedule, not the arithmetic:

* The reference optimises every (target, sample) latent on its own: a Python triple loop (targets x samples x 100
  Adam steps) over batch-1 forwards of `fusion` + `dti_head`, with a host synchronisation per step.  Here all latents
  are rows of ONE batch.  The objective is the SUM of the per-row probabilities, so each row receives exactly the
  gradient of its own batch-1 objective, and Adam's state is element-wise, so every row follows the trajectory its
  own optimiser would have produced; the best-so-far latent is tracked per row on the device.
* Sequence / SMILES featurisers (ESP fingerprints, Morgan fingerprints) are external models in the reference; callers
  pass the fingerprint tensors directly.
* `generate_molecules_batch` keeps the reference's chunking by `max_batch_size` and can sort each chunk into
  size buckets (see FullDTIModel.sample_batch).
"""
from typing import Dict, List, Optional, Tuple

import torch


class ConditionalDrugGenerator:
    def __init__(self, model, device=None, max_batch_size=512, size_buckets=1):
        self.model = model.eval()
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.max_batch_size = max_batch_size
        self.size_buckets = size_buckets

    @staticmethod
    def _aggregate(aggregator, views):
        out = aggregator(views)              # the attentive aggregator returns (embedding, attention weights)
        return out[0] if isinstance(out, tuple) else out

    # ------------------------------------------------------------------ step 2
    @torch.no_grad()
    def process_targets(self, target_fingerprints) -> torch.Tensor:
        """Target features -> [n_targets, embedding_dim].  A tensor is taken as the model's FIRST target view (the
        reference feeds the ESP view of a single-view model and skips the aggregator); a dict {view name: tensor}
        runs every view through its encoder and the aggregator."""
        names = list(self.model.hparams.target_features)
        if isinstance(target_fingerprints, dict):
            feats = [target_fingerprints[name] for name in names]
        else:
            feats = [target_fingerprints]
        embs = [self.model.target_encoders[name](f.to(self.device).float()) for name, f in zip(names, feats)]
        return embs[0] if len(embs) == 1 else self._aggregate(self.model.target_aggregator, embs)

    # ------------------------------------------------------------------ step 3
    @torch.no_grad()
    def sample_initial_latents(self, n_targets: int, n_samples_per_target: int = 16,
                               reference_fingerprints: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Half of each target's latents from N(0, I), half from the drug VAE posterior of reference molecules
        (given as fingerprints of the first drug view; N(0, I) again when absent), shuffled per target.
        Returns (latents [n_targets * n_samples_per_target, D], strategy flags: 0 random / 1 learned)."""
        dim = self.model.hparams.embedding_dim
        n_rand = n_samples_per_target // 2
        n_learn = n_samples_per_target - n_rand
        random_latents = torch.randn(n_targets * n_rand, dim, device=self.device)
        if reference_fingerprints is not None and len(reference_fingerprints) > 0:
            learned = self._sample_from_learned_distribution(n_targets * n_learn, reference_fingerprints)
        else:
            learned = torch.randn(n_targets * n_learn, dim, device=self.device)
        lat = torch.cat((random_latents.view(n_targets, n_rand, dim), learned.view(n_targets, n_learn, dim)), 1)
        flag = torch.cat((torch.zeros(n_targets, n_rand, device=self.device), torch.ones(n_targets, n_learn, device=self.device)), 1)
        perm = torch.stack([torch.randperm(n_samples_per_target) for _ in range(n_targets)]).to(self.device)
        lat = torch.gather(lat, 1, perm.unsqueeze(-1).expand(-1, -1, dim))
        flag = torch.gather(flag, 1, perm)
        return lat.reshape(-1, dim), flag.reshape(-1)

    @torch.no_grad()
    def _sample_from_learned_distribution(self, n_samples: int, reference_fingerprints: torch.Tensor) -> torch.Tensor:
        views = (list(reference_fingerprints.values()) if isinstance(reference_fingerprints, dict)
                 else [reference_fingerprints])          # dict {view name: tensor} in the model's view order, or view 0
        count = views[0].size(0)
        if count > n_samples:
            pick = torch.randperm(count, device=self.device)[:n_samples]
        else:
            pick = torch.arange(n_samples, device=self.device) % count
        fps = [v.to(self.device).float()[pick] for v in views]
        names = list(self.model.hparams.drug_features)
        embs = [self.model.drug_encoders[name](f) for name, f in zip(names, fps)]
        emb = embs[0] if len(embs) == 1 else self._aggregate(self.model.drug_aggregator, embs)
        z, _, _ = self.model.drug_kl_head(emb)
        return z + torch.randn_like(z) * 0.05

    # ------------------------------------------------------------------ step 4
    def optimize_latents_for_interaction(self, initial_latents: torch.Tensor, sampling_metadata: torch.Tensor,
                                         processed_targets: torch.Tensor, n_samples_per_target: int = 8,
                                         n_optimization_steps: int = 100, learning_rate: float = 0.01
                                         ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Gradient ascent on sigmoid(Y logit) of `dti_head(fusion(latent, target))` for every (target, sample) pair.
        Returns (best latents [n_targets * n_samples_per_target, D], metadata [., 4] = target index, sample index,
        sampling strategy, best probability) - the reference's layout."""
        n_targets = processed_targets.shape[0]
        rows = n_targets * n_samples_per_target
        latent = initial_latents[:rows].detach().to(self.device).clone().requires_grad_(True)
        target = processed_targets.detach().to(self.device).repeat_interleave(n_samples_per_target, dim=0)
        opt = torch.optim.Adam([latent], lr=learning_rate)
        best_latent = latent.detach().clone()
        best_score = torch.full((rows,), float("-inf"), device=self.device)
        params = [p for p in self.model.parameters() if p.requires_grad]
        for p in params:                                  # only the latents are optimised
            p.requires_grad_(False)
        try:
            for _ in range(n_optimization_steps):
                opt.zero_grad(set_to_none=True)
                with torch.enable_grad():
                    fused, _ = self.model.fusion(latent, target)
                    prob = torch.sigmoid(self.model.dti_head(fused)["Y"])
                    (-prob.sum()).backward()              # row-separable: each row sees its own batch-1 gradient
                opt.step()
                with torch.no_grad():
                    # as in the reference (:375-382) the stored latent is the one AFTER this step's update, paired
                    # with the probability evaluated before it
                    better = prob.detach() > best_score
                    best_score = torch.where(better, prob.detach(), best_score)
                    best_latent = torch.where(better.unsqueeze(1), latent.detach(), best_latent)
        finally:
            for p in params:
                p.requires_grad_(True)
        idx = torch.arange(rows, device=self.device)
        metadata = torch.stack(((idx // n_samples_per_target).float(), (idx % n_samples_per_target).float(),
                                sampling_metadata[:rows].to(self.device).float(), best_score), 1)
        return best_latent, metadata

    # ------------------------------------------------------------------ step 5
    @torch.no_grad()
    def generate_molecules_batch(self, optimized_latents: torch.Tensor, metadata: torch.Tensor,
                                 n_molecules_per_latent: int = 16, return_graphs: bool = False) -> List[Dict]:
        """Reverse diffusion for every latent, `max_batch_size // n_molecules_per_latent` latents per chunk.
        One result dict per molecule with the reference's keys; 'molecule' is an RDKit Mol when RDKit is importable,
        else the (atom ids, bond ids) tensors."""
        n_latents = optimized_latents.shape[0]
        per_chunk = max(1, min(self.max_batch_size // n_molecules_per_latent, n_latents))
        meta = metadata.cpu().tolist()
        results = []
        for start in range(0, n_latents, per_chunk):
            end = min(start + per_chunk, n_latents)
            groups = self.model.sample_batch(drug_embeddings=optimized_latents[start:end].to(self.device),
                                             num_samples_per_embedding=n_molecules_per_latent, return_graphs=return_graphs,
                                             size_buckets=self.size_buckets)
            for (t_idx, s_idx, strategy, score), group in zip(meta[start:end], groups):
                for d_idx, mol in enumerate(group):
                    res = {"target_idx": int(t_idx), "sample_idx": int(s_idx), "sampling_strategy": int(strategy),
                           "diff_idx": d_idx, "optimization_score": float(score), "molecule": mol, "smiles": None,
                           "valid": False}
                    res["smiles"], res["valid"] = self._smiles(mol)
                    results.append(res)
        return results

    @staticmethod
    def _smiles(mol):
        try:
            from rdkit import Chem
        except Exception:  # noqa: BLE001  (RDKit absent: validity cannot be judged here)
            return None, False
        if mol is None or isinstance(mol, tuple):
            return None, False
        try:
            smiles = Chem.MolToSmiles(mol, isomericSmiles=False)
            return smiles, smiles is not None and Chem.MolFromSmiles(smiles) is not None
        except Exception:  # noqa: BLE001
            return None, False
