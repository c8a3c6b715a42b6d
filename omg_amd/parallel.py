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
