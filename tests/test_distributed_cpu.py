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
