"""Multi-GPU readiness on the ONE GPU a test box has (-m gpu):
  * the data-parallel path with the REAL pipeline per rank: two processes share cuda:0, each runs its shard of four requests through
    the tiny SDXL-topology UNet, final latents are all_gathered (gloo) — equal, bit for bit, to the single-process run;
  * RCCL at world size 1: bench.py under torch.distributed.run with backend nccl (process group, barrier, all_gather, all_reduce)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_requests(indices, concept_shard=None, use_graph=False, controlnet=False):
    """Final latents of the requests `indices` (seeded by their GLOBAL index), batched in lock-step as bench.py does."""
    import contextlib, io
    from omg_amd import controller as pc
    from omg_amd.pipeline import LoraMultiConceptPipeline, revise_regionally_controlnet_forward
    from omg_amd.schedulers import make_scheduler
    from omg_amd.synthetic import c2_inputs, c2_masks, make_concept_models
    from omg_amd.unet import UNet2DConditionModel, UNetConfig
    dev = torch.device("cuda:0")
    cfg = UNetConfig.tiny()
    unet = UNet2DConditionModel(cfg, dtype=torch.float16, device=dev).init_synthetic_(seed=0)
    P = "a man and a woman walking on the street"
    HW = cfg.sample_size * 8
    ctl = pc.AttentionReplace([P, P], 50, {"default_": 1.0}, 0.4, HW // 32, HW // 32, device=dev, dtype=torch.float16)
    with contextlib.redirect_stdout(io.StringIO()):
        revise_regionally_controlnet_forward(unet, ctl)
    concept = make_concept_models(unet, n_concepts=2, rank=8)
    pipe = LoraMultiConceptPipeline(unet, make_scheduler("ddim"))
    masks = c2_masks(HW, HW, device=dev)
    reqs = []
    for i in indices:
        r = c2_inputs(unet, seed=100 + i, height=HW, width=HW)
        r["region_masks"] = masks
        reqs.append(r)
    if not reqs:
        return torch.zeros((0, 2, 4, cfg.sample_size, cfg.sample_size))
    extra = {}
    if controlnet:
        from omg_amd.controlnet import ControlNetModel
        from oracle import controlnet as ocn, unet as ou          # seeded ControlNet weights in the shared key layout
        cn = ControlNetModel(cfg, dtype=torch.float16, device=dev)
        cn.load_state_dict({k: v.to(torch.float16) for k, v in ocn.init_state_dict(ou.UNetConfig.tiny(), seed=3).items()})
        extra = dict(controlnet=cn, controlnet_conditioning_scale=0.8,
                     controlnet_image=torch.rand(1, 3, HW, HW, generator=torch.Generator().manual_seed(9)))
    lat = pipe.generate_many(reqs, height=HW, width=HW, num_inference_steps=20, guidance_scale=7.5, cross_attention_kwargs={"scale": 0.8},
                             controller=ctl, concept_models=concept, stage=2, lora_list=["concept0", "concept1"], styleL=False,
                             concept_shard=concept_shard, use_graph=use_graph, **extra)
    return lat.cpu()


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    from omg_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    mine = parallel.shard_indices(n_images, r, w)
    local = _run_requests(mine)
    allt = parallel.gather_latents(local, n_images, r, w)
    parallel.barrier()
    q.put((r, allt))
    torch.distributed.destroy_process_group()


def test_pipeline_per_rank_then_gather_equals_single_process(dev):
    n_images, world = 3, 2                     # ragged: rank 0 runs requests 0, 1 in lock-step, rank 1 request 2 alone
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = torch.cat([_run_requests([i]) for i in range(n_images)])      # one request at a time, one process
    for r in range(world):
        assert res[r].shape == single.shape
        assert torch.equal(res[r], single), f"rank {r}: gathered latents differ from the single-process run"
    assert not torch.equal(single[0], single[1])


def _shard_worker(rank, world, port, n_images, use_graph, controlnet, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    from omg_amd import parallel
    parallel.init_distributed(backend="gloo")
    shard = parallel.ConceptShard()
    assert (shard.rank, shard.world) == (rank, world)
    lat = _run_requests(list(range(n_images)), concept_shard=shard, use_graph=use_graph, controlnet=controlnet)
    parallel.barrier()
    q.put((rank, lat))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n_images,use_graph,controlnet", [(2, 1, False, False), (3, 1, True, False), (2, 2, True, True)])
def test_concept_sharded_step_equals_the_unsharded_call(dev, world, n_images, use_graph, controlnet):
    """north_star's "independent per-concept UNet passes ... shard across the GPUs" (finer-grain option): the ranks of a
    ConceptShard split the forward units of every step (main block / concept pairs), exchange the noise predictions with one
    all_gather per step and apply the same fusion + CFG + scheduler kernel — every rank ends with the latents of the unsharded call,
    bit for bit (eager and with the local forward replayed from a hipGraph; with ControlNet on the main block)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n_images, use_graph, controlnet, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _run_requests(list(range(n_images)), controlnet=controlnet)
    for r in range(world):
        assert torch.equal(res[r], single), f"rank {r}: sharded latents differ from the unsharded call"


def test_rccl_world_size_one_bench_smoke(dev):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` exactly as the driver launches N > 1, backend nccl
    (= RCCL): init_process_group, barrier, all_gather of the latents and the max-over-ranks all_reduce all run; one JSON line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--tiny",
           "--images-per-step", "2", "--no-cpu-baseline", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["tiny_debug"] is True
