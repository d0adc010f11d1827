"""Batch-sharded reverse-diffusion sampling over the ranks of a process group (one process per GPU).

Every molecule is sampled independently, so the job partitions with no exchange inside the T-step loop: rank r takes
a contiguous block of the conditioning embeddings, runs `FullDTIModel.sample_batch` on it in chunks (the reference's
consumer does the same chunking on one device, mb_vae_dti/validating/conditioning.py:420-525 with
max_batch_size = 512 at :800), and the collapsed class ids - int8, [atoms] and [atoms, atoms] per molecule - are
gathered on rank 0 once at the end, in the original embedding order.  Each rank seeds torch's generator with
`seed + rank`, so ranks draw different noise and a rerun with the same world size reproduces the same molecules.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of rank's contiguous block; the first `total % world` ranks hold one item more."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_graphs(graphs, n_max: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """[(atom ids [k], bond ids [k, k]), ...] -> int8 atoms [m, n_max] (-1 beyond a molecule's size), int8 bonds
    [m, n_max, n_max] (0 beyond it), int32 sizes [m]."""
    m = len(graphs)
    atoms = torch.full((m, n_max), -1, dtype=torch.int8)
    bonds = torch.zeros((m, n_max, n_max), dtype=torch.int8)
    sizes = torch.zeros(m, dtype=torch.int32)
    for i, (a, b) in enumerate(graphs):
        k = int(a.numel())
        if k > n_max:
            raise ValueError(f"molecule {i} has {k} atoms, more than n_max = {n_max}")
        atoms[i, :k] = a.to(torch.int8)
        bonds[i, :k, :k] = b.to(torch.int8)
        sizes[i] = k
    return atoms, bonds, sizes


def unpack_graphs(atoms: torch.Tensor, bonds: torch.Tensor, sizes: torch.Tensor):
    return [(atoms[i, :k].long(), bonds[i, :k, :k].long()) for i, k in enumerate(sizes.tolist())]


@torch.no_grad()
def sample_sharded(model, drug_embeddings: torch.Tensor, num_nodes: Optional[torch.Tensor] = None,
                   num_samples_per_embedding: int = 1, n_max: Optional[int] = None, chunk: int = 512, seed: int = 0,
                   group=None, device=None, **sample_kwargs) -> Optional[List[List[Tuple[torch.Tensor, torch.Tensor]]]]:
    """Samples `num_samples_per_embedding` molecules for every row of `drug_embeddings` (the SAME tensor on every
    rank, any device), this rank's block on `device`.  Rank 0 returns the molecules grouped per embedding as
    (atom ids, bond ids) pairs, other ranks return None.  Without an initialised process group it is the plain
    single-device chunked loop.  `sample_kwargs` go to `model.sample_batch` (size_buckets, use_graph)."""
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    total = drug_embeddings.size(0)
    if n_max is None:
        n_max = int(model.hparams.dataset_infos["general"]["max_n_nodes"])
    if device is None:
        device = next(model.parameters()).device
    torch.manual_seed(seed + rank)
    lo, hi = shard_bounds(total, rank, world)
    per_chunk = max(1, chunk // num_samples_per_embedding)
    graphs = []
    for s in range(lo, hi, per_chunk):
        e = min(s + per_chunk, hi)
        groups = model.sample_batch(drug_embeddings[s:e].to(device),
                                    num_nodes=None if num_nodes is None else num_nodes[s:e],
                                    num_samples_per_embedding=num_samples_per_embedding, return_graphs=True,
                                    **sample_kwargs)
        for grp in groups:
            graphs.extend(grp)
    atoms, bonds, sizes = pack_graphs(graphs, n_max)
    if not distributed or world == 1:
        flat = unpack_graphs(atoms, bonds, sizes)
    else:
        # equal-size messages: every rank pads its block to the largest block (they differ by one embedding at most)
        cap = (shard_bounds(total, 0, world)[1] - shard_bounds(total, 0, world)[0]) * num_samples_per_embedding
        backend = dist.get_backend(group)
        wire = torch.device(device) if backend == "nccl" else torch.device("cpu")

        def padded(t, fill):
            out = torch.full((cap,) + tuple(t.shape[1:]), fill, dtype=t.dtype)
            out[:t.size(0)] = t
            return out.to(wire)

        parts = []
        for t, fill in ((atoms, -1), (bonds, 0), (sizes, 0)):
            mine = padded(t, fill)
            bucket = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, bucket, dst=0, group=group)
            parts.append(bucket)
        if rank != 0:
            return None
        flat = []
        for r in range(world):
            cnt = (shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]) * num_samples_per_embedding
            flat.extend(unpack_graphs(parts[0][r][:cnt].cpu(), parts[1][r][:cnt].cpu(), parts[2][r][:cnt].cpu()))
    k = num_samples_per_embedding
    return [flat[i * k:(i + 1) * k] for i in range(total)]
