"""Multi-GPU: data-parallel over images, one process per GPU, RCCL over xGMI.

The reference is single-process / single-device (no collective exists on any OMG path).
Images (prompt, seed) are independent units, weights are replicated (~13 GB of 288 GB), every rank runs
the whole two-stage loop locally with ZERO per-step communication, and the only collective is one
``all_gather`` of the final latents per batch (4x128x128 fp32 = 256 KiB per image: latency-bound, tens of
microseconds; ring vs tree and xGMI link bandwidth are irrelevant at this size).
``backend="nccl"`` is RCCL on ROCm; the same code runs under ``gloo`` on CPU tensors for the tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from the torchrun environment; returns (rank, world, local_rank). No-op for 1 process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("OMG_FORCE_DIST") == "1"      # torchrun with one process: still
    if (world > 1 or launched) and not dist.is_initialized():                                    # bring RCCL up (world-size-1 smoke)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Contiguous block partition of ``range(n_items)``; the first ``n_items % world`` ranks get one extra."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def gather_latents(local: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """all_gather of per-rank latents ``(n_local, ...)`` into ``(n_total, ...)`` in global image order.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    base, extra = divmod(n_total, world)
    n_max = base + (1 if extra else 0)
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    parts = [out[r][: len(shard_indices(n_total, r, world))] for r in range(world)]
    return torch.cat(parts, dim=0)


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------------------------
# Finer-grain option of SURVEY §8(e) / north_star "independent per-concept UNet passes ... shard across the GPUs":
# within ONE lock-step batch, a denoising step is a set of independent forward UNITS — per request the main block
# [unc0, unc1, cond0, cond1] (4 rows; cond1 borrows cond0's Q,K at every layer, so the block stays whole) and, in fused steps,
# one [unc, cond] pair per masked concept (2 rows).  Ranks replicate the latents and the scheduler state; each step every rank
# runs only its units through the UNet, ONE all_gather exchanges the fp32 noise predictions (<= 8 x 256 KiB per request: latency
# bound), and every rank then executes the same fusion + CFG + scheduler kernel on the same inputs — the latents stay bitwise
# replicated without a broadcast.  This is a LATENCY mode (critical path of a fused step: 4 of 4 + 2K rows); for throughput the
# data-parallel split over images above has no per-step communication at all and is what bench.py measures.
def assign_units(n_requests: int, n_concepts: int, fused: bool, world: int):
    """Deterministic greedy balance of the step's units over ``world`` ranks.  Returns ``per_rank``: for every rank a pair
    (main request indices, [(request, concept position)]).  Largest units first, to the least loaded rank, ties to the lowest rank —
    every rank computes the same table."""
    units = [(4, j, -1) for j in range(n_requests)]
    if fused:
        units += [(2, j, c) for j in range(n_requests) for c in range(n_concepts)]
    load = [0] * world
    per_rank = [([], []) for _ in range(world)]
    for rows, j, c in sorted(units, key=lambda u: (-u[0], u[1], u[2])):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += rows
        (per_rank[r][0] if c < 0 else per_rank[r][1]).append(j if c < 0 else (j, c))
    return per_rank


def unit_rows(per_rank, n_requests: int, n_concepts: int):
    """Row bookkeeping of :func:`assign_units` in the pipeline's batch layout (main rows ``4 j + r``; concept rows
    ``4 n + 2 K j + 2 c + r``).  For every rank: ``src`` = rows of the MAIN block its local batch is filled from (a concept pair is
    the request's edited conditional input ``4 j + 3`` twice, lora_pipeline.py:583-585) and ``dst`` = rows of the full prediction
    buffer its outputs belong to; local order = main units first, then concept pairs."""
    nm = 4 * n_requests
    out = []
    for mains, concs in per_rank:
        src = [4 * j + r for j in mains for r in range(4)] + [4 * j + 3 for j, _ in concs for _ in range(2)]
        dst = [4 * j + r for j in mains for r in range(4)] + [nm + 2 * n_concepts * j + 2 * c + r for j, c in concs for r in range(2)]
        out.append((src, dst))
    return out


class ConceptShard:
    """Handle passed to ``LoraMultiConceptPipeline.generate_many(concept_shard=...)``: this process' rank in the group that splits
    one lock-step batch, and the exchange of the per-step predictions."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        on = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = (dist.get_rank(group) if on else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if on else 1) if world is None else world

    def exchange(self, local: torch.Tensor, counts: Sequence[int], dsts: Sequence[torch.Tensor], full: torch.Tensor) -> None:
        """``local``: (max(counts), ...) buffer whose first ``counts[rank]`` rows are this rank's predictions; after the call row
        ``dsts[r][i]`` of ``full`` holds row i of rank r, for every rank — one all_gather of equally sized buffers."""
        if self.world == 1:
            full.index_copy_(0, dsts[0], local[: counts[0]])
            return
        if local.is_cuda and dist.get_backend(self.group) == "gloo":       # gloo has no all_gather on device tensors (tests: two ranks on one GPU)
            host = local.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host, group=self.group)
            parts = [t.to(local.device) for t in parts]
        else:                                                                # RCCL over xGMI
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(parts, local.contiguous(), group=self.group)
        for r in range(self.world):
            if counts[r]:
                full.index_copy_(0, dsts[r], parts[r][: counts[r]])
