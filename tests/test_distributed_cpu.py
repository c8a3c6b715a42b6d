"""CPU test (-m "not gpu"): the N>1 path — image sharding + final-latent all_gather — under gloo, world_size 2
and 3 (ragged), spawned as real processes (rendezvous on 127.0.0.1)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from omg_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    mine = parallel.shard_indices(n_images, r, w)
    # stand-in for the local denoising result of image i: a tensor that depends only on i
    local = torch.stack([torch.full((4, 8, 8), float(i)) + torch.arange(8.0) for i in mine]) if mine else torch.zeros(0, 4, 8, 8)
    allt = parallel.gather_latents(local, n_images, r, w)
    parallel.barrier()
    t = parallel.max_over_ranks(float(r + 1), "cpu")
    q.put((r, allt[:, 0, 0, 0].tolist(), tuple(allt.shape), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(2, 8), (3, 7)])
def test_dp_shard_and_gather_gloo(world, n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, firsts, shape, t in res:
        assert shape == (n_images, 4, 8, 8)
        assert firsts == [float(i) for i in range(n_images)], "gathered images must be in global order on every rank"
        assert t == float(world)


# ---- concept sharding (SURVEY 8(e) finer-grain option): unit assignment, row maps and the per-step exchange under gloo
def test_unit_assignment_is_a_balanced_partition():
    from omg_amd.parallel import assign_units, unit_rows
    for n, K, world in [(1, 2, 2), (1, 2, 3), (1, 3, 8), (2, 2, 3), (8, 2, 8), (3, 0, 2)]:
        for fused in (False, True):
            per_rank = assign_units(n, K, fused, world)
            rows = unit_rows(per_rank, n, K)
            mains = sorted(j for m, _ in per_rank for j in m)
            concs = sorted(u for _, c in per_rank for u in c)
            assert mains == list(range(n))                                           # every main block exactly once
            assert concs == ([(j, c) for j in range(n) for c in range(K)] if fused else [])
            dst = sorted(d for _, ds in rows for d in ds)
            assert dst == list(range(4 * n + (2 * K * n if fused else 0)))           # the prediction buffer is covered exactly once
            for (m, c), (src, ds) in zip(per_rank, rows):
                assert len(src) == len(ds) == 4 * len(m) + 2 * len(c)
                assert src[: 4 * len(m)] == ds[: 4 * len(m)]                         # main rows read and write the same rows
                assert all(s % 4 == 3 for s in src[4 * len(m):])                     # a concept pair reads the edited conditional input
            loads = [len(ds) for _, ds in rows]
            assert max(loads) - min(loads) <= 4                                      # greedy: never worse than one main block apart
    assert assign_units(1, 2, True, 2) == [([0], []), ([], [(0, 0), (0, 1)])]       # K = 2 on two GPUs: main pass | both concept passes


def _shard_worker(rank, world, port, n, K, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from omg_amd import parallel
    parallel.init_distributed(backend="gloo")
    sh = parallel.ConceptShard()
    per_rank = parallel.assign_units(n, K, True, world)
    rows = parallel.unit_rows(per_rank, n, K)
    counts = [len(d) for _, d in rows]
    dsts = [torch.tensor(d, dtype=torch.long) for _, d in rows]
    local = torch.zeros(max(counts), 3)
    for i, d in enumerate(rows[sh.rank][1]):      # stand-in prediction of global row d: a value that depends only on d
        local[i] = float(d)
    full = torch.full((4 * n + 2 * K * n, 3), -1.0)
    sh.exchange(local, counts, dsts, full)
    parallel.barrier()
    q.put((rank, full[:, 0].tolist()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n,K", [(2, 1, 2), (3, 2, 2), (4, 1, 3)])
def test_concept_shard_exchange_gloo(world, n, K):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, col in res:
        assert col == [float(i) for i in range(4 * n + 2 * K * n)], "every rank must hold every unit's prediction at its global row"
