"""CPU tests (-m "not gpu"): host-side logic of the product — controller drop-in vs the reference-generated
golden vectors, scheduler coefficient tables vs the oracle schedulers, weight packing, sharding arithmetic."""
import os

import numpy as np
import pytest
import torch

from omg_amd import controller as pc
from omg_amd import ops, parallel
from omg_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
from oracle import controller as oc
from oracle import schedulers as osched

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = "a man and a woman walking on the street"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "controller_golden.npz"))


def cli_controller():
    return pc.AttentionReplace([P, P], 50, cross_replace_steps={"default_": 1.0}, self_replace_steps=0.4, width=32, height=32,
                               tokenizer=None, device="cpu", dtype=torch.float32)


def test_controller_construction_matches_reference(gold):
    c = cli_controller()
    assert np.array_equal(c.mapper.numpy(), gold["t1_mapper"])
    assert np.array_equal(c.cross_replace_alpha.numpy(), gold["t1_alpha"])
    assert c.num_self_replace == tuple(gold["t1_num_self_replace"]) and c.batch_size == int(gold["t1_batch_size"])
    assert c.is_pure_replacement and c.num_att_layers == -1 and (c.cur_step, c.cur_att_layer) == (0, 0)


@pytest.mark.parametrize("name,is_cross,step", [("cross_q64", True, 0), ("self_q1024_s0", False, 0), ("self_q1056_s0", False, 0),
                                                  ("self_q1024_s19", False, 19), ("self_q1024_s20", False, 20)])
def test_controller_protocol_call_bitwise(gold, name, is_cross, step):
    c = cli_controller()
    c.num_att_layers = 4
    c.cur_step = step
    x = torch.from_numpy(gold[name + "_in"].copy())
    y = c(x, is_cross, "down")
    assert y is x and np.array_equal(y.numpy(), gold[name + "_out"])
    # the fused path must make the same decision as the probability edit
    c2 = cli_controller(); c2.num_att_layers = 4; c2.cur_step = step
    src = c2.fused_qk_src(is_cross, x.shape[1], 4, "down", device="cpu")
    changed = not np.array_equal(gold[name + "_in"], gold[name + "_out"])
    assert (src is not None) == changed
    if src is not None:
        assert src.tolist() == [0, 1, 2, 2]
    assert (c2.cur_step, c2.cur_att_layer) == (c.cur_step, c.cur_att_layer)


def test_controller_counters_and_reset(gold):
    c = cli_controller()
    c.num_att_layers = 140
    for _ in range(140):
        c.fused_qk_src(True, 4096, 4, "mid", device="cpu")
    assert [c.cur_step, c.cur_att_layer] == list(gold["t5_counters"])
    c.reset()
    assert (c.cur_step, c.cur_att_layer) == (0, 0)
    with pytest.raises(ValueError):
        c.fused_qk_src(True, 64, 6, "down", device="cpu")     # batch must be 2 * len(prompts)


def test_controller_general_mapper_matches_reference(gold):
    prompts = ["a man on the road", "a woman on the road"]
    c = pc.AttentionReplace(prompts, 10, {"default_": 0.6, "road": (0.2, 0.9)}, (0.1, 0.5), 4, 4, tokenizer=oc.PieceTokenizer(),
                            device="cpu", dtype=torch.float32)
    assert np.array_equal(c.mapper.numpy(), gold["gen_mapper"]) and np.array_equal(c.cross_replace_alpha.numpy(), gold["gen_alpha"])
    assert not c.is_pure_replacement
    with pytest.raises(RuntimeError):
        c.fused_qk_src(True, 16, 4)
    c.num_att_layers = 2
    for step in (0, 3, 7):
        for kind in ("cross", "self"):
            c.reset(); c.cur_step = step
            key = f"gen_s{step}_{kind}"
            y = c(torch.from_numpy(gold[key + "_in"].copy()), kind == "cross", "mid")
            np.testing.assert_allclose(y.numpy(), gold[key + "_out"], rtol=0, atol=1e-7)
    assert np.array_equal(pc.get_replacement_mapper(["a man on the street", "a dog on the street"], oc.WhitespaceTokenizer()).numpy(),
                          gold["t6_mapper_swap"])
    with pytest.raises(ValueError):
        pc.get_replacement_mapper(["a man", "a man walking"], oc.WhitespaceTokenizer())


ALIGN_CASES = [
    (["a man on the road", "a woman on the road", "a superman on the road"], "piece"),
    (["photograph of a man walking the dog", "photograph of a woman walking the cat"], "piece"),
    (["extraordinarily dog in the garden", "x superman in the street"], "piece"),
    (["a man and a woman walking on the street", "a dog and a cat walking on the street"], "white"),
    (["woman woman woman", "man x woman"], "piece"),
]


@pytest.mark.parametrize("n", range(len(ALIGN_CASES)))
def test_alignment_tables_match_reference_generated_vectors(n):
    """The package builds the mapper from run offsets and the alpha table from broadcast windows (omg_amd/controller.py); the vectors
    are the reference's own seq_aligner / p2p_utils outputs (tests/golden/make_golden.py: controller_alignment_vectors): several
    edited prompts, several replaced words, unequal piece counts, word windows that hit several tokens.  Bit-exact."""
    gold = np.load(os.path.join(GOLD, "controller_alignment_golden.npz"))
    prompts, tk = ALIGN_CASES[n]
    tok = oc.PieceTokenizer() if tk == "piece" else oc.WhitespaceTokenizer()
    assert np.array_equal(pc.get_replacement_mapper(prompts, tok).numpy(), gold[f"c{n}_mapper"])
    specs = [(50, {"default_": 1.0}), (10, {"default_": 0.6, prompts[1].split(" ")[-1]: (0.2, 0.9)}), (7, {"default_": (0.1, 0.8), prompts[1].split(" ")[0]: 0.3})]
    for m, (S, spec) in enumerate(specs):
        assert np.array_equal(pc.get_time_words_attention_alpha(prompts, S, dict(spec), tok).numpy(), gold[f"c{n}_alpha{m}"]), (n, m)


@pytest.mark.parametrize("n,k", list(enumerate((76, 80, 75, 74))))
def test_alignment_of_prompts_longer_than_the_token_window(n, k):
    """a replaced word reached exactly where the 77-token window ends (or beyond it) made the run-offset construction index
    row 77; the reference's two-pointer walk stops there and returns the matrix.  Vectors from the reference's own seq_aligner
    (tests/golden/make_golden.py: LONG_REPLACED_WORDS, 86-word prompts on the whitespace tokenizer).  Bit-exact."""
    gold = np.load(os.path.join(GOLD, "controller_alignment_golden.npz"))
    words = [f"w{i}" for i in range(86)]
    other = list(words); other[k] = "changed"
    got = pc.get_replacement_mapper([" ".join(words), " ".join(other)], oc.WhitespaceTokenizer()).numpy()
    assert np.array_equal(got, gold[f"long{n}_mapper"])


@pytest.mark.parametrize("n", [50, 30, 10])
def test_scheduler_tables_reproduce_the_oracle_schedulers(n):
    rng = np.random.default_rng(0)
    for mine, ref in ((DDIMScheduler(), osched.DDIM(n)), (EulerDiscreteScheduler(), osched.EulerDiscrete(n))):
        mine.set_timesteps(n, device="cpu")
        assert np.array_equal(np.asarray(mine.timesteps.numpy(), dtype=np.float64), np.asarray(ref.timesteps, dtype=np.float64))
        assert abs(mine.init_noise_sigma - ref.init_noise_sigma) < 1e-12
        tab = mine.coef_table("cpu").double().numpy()
        x = rng.standard_normal(16)
        for i in range(n):
            eps = rng.standard_normal(16)
            want = ref.step(eps, i, x)
            got = tab[i, 0] * x + tab[i, 1] * eps
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(mine.cin[i] * x, ref.scale_model_input(x, i), rtol=1e-12)
            if i + 1 < n:
                np.testing.assert_allclose(tab[i, 2], mine.cin[i + 1], rtol=1e-6)
            t = mine.timesteps[i]
            np.testing.assert_allclose(mine.step(torch.from_numpy(eps), t, torch.from_numpy(x))[0].numpy(), want, rtol=2e-6, atol=2e-6)  # fp32 inside
            x = want


def test_weight_packing_layouts():
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = ops.pack_conv_weight(w)
    assert p.shape == (2, 27) and p[1, (2 * 3 + 1) * 3 + 2] == w[1, 2, 2, 1]      # [co][ky][kx][ci]
    perm = ops.geglu_row_perm(256)
    assert sorted(perm.tolist()) == list(range(256))
    assert perm[:32].tolist() == list(range(32)) and perm[32:64].tolist() == list(range(128, 160)) and perm[64] == 32


def test_shard_indices_cover_everything_once():
    for n in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            seen = sum((parallel.shard_indices(n, r, world) for r in range(world)), [])
            assert seen == list(range(n))
            sizes = [len(parallel.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_gemm_tile_choice_on_the_workload_shapes():
    """The cost model of csrc/gemm.hip (host code, no GPU): which tile the benchmark's layers get.  1 = 128x128 (two blocks per CU),
    14 = 256x128 on eight waves, 24 = 128x320 on four waves (conv only), 25 = 256x256 on four waves (ring K loop, persistent walk),
    28 = 256x320 on four waves.  Fourth argument: bit 0 = implicit-GEMM convolution, bit 1 = GEGLU epilogue.  The expectations are the
    round-5 interleaved A/B results (profiles/r05_exp_v13_ab_*.log, r05_exp_v12_ab_*.log)."""
    from omg_amd import _lib as L
    pick = L.lib().omg_debug_choose_variant
    # Linear layers of the fused step at 8 requests: 64 groups of 1024 (32x32) / 4096 (64x64) rows
    assert pick(1024, 64, 10240, 2) == 25 and pick(4096, 64, 5120, 2) == 25          # GEGLU never leaves the 256x256 tile
    assert pick(1024, 64, 1280, 0) == 25 and pick(1024, 64, 3840, 0) == 25           # N = 1280 k fills whole rounds of 256-wide tiles
    assert pick(4096, 64, 640, 0) == 28 and pick(4096, 64, 1920, 0) == 28            # 256-wide pads 640 -> 768, 1920 -> 2048
    assert pick(32768, 1, 1280, 0) == 28                                             # the concept rows alone: 512 tiles = 2 rounds instead of 640 = 3
    assert pick(32768, 1, 10240, 2) == 25
    # convs: the 256x320 tile replaces the 128x320 one wherever it fills the chip, and the 256x256 one where 320-wide tiles save a round
    assert pick(1048576, 1, 320, 1) == 28 and pick(262144, 1, 640, 1) == 28
    assert pick(32768, 1, 1280, 1) == 28 and pick(65536, 1, 1280, 1) == 28 and pick(262144, 1, 1280, 1) == 28
    assert pick(32768, 1, 320, 1) == 24              # 128 tiles of 256x320 = half a round; 256 tiles of 128x320 = one whole round of a cheaper tile
    # narrow outputs (LoRA down-projection, ControlNet conditioning) never take a 256-wide tile; tiny launches stay on v1
    assert pick(65536, 1, 64, 0) == 14 and pick(2048, 1, 1280, 0) == 1 and pick(77 * 8, 1, 2560, 0) == 1


def test_stage_cache_bookkeeping():
    """StageCache (SURVEY 7.4 between the two calls of an image): keys are content digests — equal tensors give equal keys whatever their
    identity, any change of content, shape or parameter gives another — entries are copies, the base trajectory is complete only when
    every step behind the first fused one is there, and eviction drops both."""
    from omg_amd.pipeline import StageCache
    a = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    k1 = StageCache.digest((50, 7.5, "DDIM"), a, None)
    assert k1 == StageCache.digest((50, 7.5, "DDIM"), a.clone(), None)
    assert k1 != StageCache.digest((50, 7.0, "DDIM"), a, None) != StageCache.digest((50, 7.5, "DDIM"), a + 1, None)
    assert k1 != StageCache.digest((50, 7.5, "DDIM"), a.reshape(4, 3), None) and k1 != StageCache.digest((50, 7.5, "DDIM"), a.double(), None)
    c = StageCache(max_entries=2)
    lat = torch.ones(2, 4, 2, 2)
    c.put("x", lat)
    lat.zero_()
    assert c.get("x").sum() == 2 * 4 * 2 * 2 and c.get("nope") is None, "entries are copies"
    for k in range(17, 21):
        c.put_base("x", k, torch.full((1, 4, 2, 2), float(k)), torch.zeros(1, 4, 2, 2, dtype=torch.float16))
    assert c.get_base("x", 16, 20) is not None and c.get_base("x", 16, 21) is None and c.get_base("x", 15, 20) is None
    assert float(c.get_base("x", 16, 20)[19][0].mean()) == 19.0
    c.put("y", lat); c.put("z", lat)                      # third key evicts the oldest, with its trajectory
    assert c.get("x") is None and c.get_base("x", 16, 20) is None and set(c.entries) == {"y", "z"}


def test_controller_source_vector_without_the_base_unconditional_row():
    """`drop_unc0`: a main block [unc1, cond0, cond1] — every conditional row still borrows Q, K from the FIRST conditional one, which is now
    row 1 of the block; concept rows batched behind keep their own; a batch of any other size is refused as before."""
    P = "a man and a woman walking on the street"
    ctl = pc.AttentionReplace([P, P], 50, {"default_": 1.0}, 0.4, 4, 4, device="cpu")
    ctl.num_att_layers = 4
    full = ctl.qk_src_vector(4, "cpu", total_batch=4 * 2 + 4, images=2).tolist()
    assert full == [0, 1, 2, 2, 4, 5, 6, 6, 8, 9, 10, 11]
    three = ctl.qk_src_vector(3, "cpu", total_batch=3 * 2 + 4, images=2).tolist()
    assert three == [0, 1, 1, 3, 4, 4, 6, 7, 8, 9]
    assert ctl.fused_qk_src(True, 16, 3, device="cpu", total_batch=3, images=1).tolist() == [0, 1, 1]
    with pytest.raises(ValueError):
        ctl.fused_qk_src(True, 16, 5, device="cpu")


def test_reference_parameters_that_are_not_implemented_are_refused_not_ignored():
    """B3: a non-default value of a parameter the reference's __call__ acts on and this engine does not implement (callbacks, prompt_2, clip_skip,
    ...; round 6 implemented guess_mode, control_guidance_start / _end and a list of ControlNets — tests/test_pipeline_gpu.py) must raise — silently dropping it returns an image the reference would not have produced; names nobody knows raise
    too; what the shipped scripts pass and the reference itself never reads (inference_lora.py:241-245 `spatial_condition`) is accepted."""
    from omg_amd import _lib as L
    from omg_amd.pipeline import InstantidMultiConceptPipeline, LoraMultiConceptPipeline, refuse_unimplemented

    lp, ip = object.__new__(LoraMultiConceptPipeline), object.__new__(InstantidMultiConceptPipeline)
    with pytest.raises(L.OmgHipError):      # the reference's own InstantID loop fails on guess_mode (instantid_pipeline.py:638-657): refused with that explanation
        ip(prompt_embeds=torch.zeros(2, 77, 8), guess_mode=True)
    for bad in (dict(num_images_per_prompt=2), dict(ip_adapter_image=object()), dict(guidance_rescale=0.7), dict(denoising_end=0.8), dict(no_such_argument=1)):
        with pytest.raises(L.OmgHipError, match="not implemented|unknown keyword"):
            lp(**bad)
        if "num_images_per_prompt" not in bad:
            with pytest.raises(L.OmgHipError, match="not implemented|unknown keyword"):
                ip(prompt_embeds=torch.zeros(2, 77, 8), **bad)
    # defaults in any of the reference's spellings, and the names its **kwargs swallows, pass the gate
    refuse_unimplemented(dict(num_images_per_prompt=[1], ip_adapter_image=None),
                         dict(spatial_condition=None, indices_to_alter=None), "test")


def test_controlnet_keep_is_the_reference_s_schedule():
    """lora_pipeline.py:275-286, :421-428, :511-517: the per-step factor on every ControlNet's conditioning scale.  Pinned by what the reference's own
    loop handed its ControlNets in the build container (tests/golden/make_golden_loop.py records every `conditioning_scale` it was called with)."""
    import os
    import sys
    import numpy as np
    from omg_amd.pipeline import controlnet_keep
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, gold_dir)
    import make_golden_loop as mk
    gold = np.load(os.path.join(gold_dir, "loop_golden.npz"))
    for case in mk.CASES:
        name, steps, flow = case[0], case[2], case[7]
        if f"{name}/controlnet_scales_seen" not in gold.files or not flow.startswith("lora_cn"):
            continue
        seen = gold[f"{name}/controlnet_scales_seen"]                     # (nets, 2 stages x steps)
        kw = mk.CN_VARIANTS[flow]
        keep = controlnet_keep(steps, kw.get("control_guidance_start", 0.0), kw.get("control_guidance_end", 1.0), seen.shape[0])
        sc = [mk.CN_SCALE, mk.T2I_SCALE][: seen.shape[0]]
        want = np.array([[sc[k] * keep[i][k] for i in range(steps)] * 2 for k in range(seen.shape[0])])
        assert np.array_equal(seen, want), name
    # the InstantID twin: ONE window for the IdentityNet and the t2i ControlNet (instantid_pipeline.py:477-483, :566-578)
    kw = mk.CN_VARIANTS["iid_t2i_window"]
    keep = [k_[0] for k_ in controlnet_keep(20, kw["control_guidance_start"], kw["control_guidance_end"], 1)]
    assert np.array_equal(gold["euler_instantid_t2i_window/controlnet2_scales_seen"], np.array([mk.T2I_SCALE * k_ for k_ in keep] * 2))
    # scalars broadcast to every net; a scalar start beside a list of ends (and the other way round) as :275-279
    assert controlnet_keep(4, 0.0, 1.0, 3) == [[1.0] * 3] * 4
    assert controlnet_keep(4, 0.5, [1.0, 0.75]) == [[0.0, 0.0], [0.0, 0.0], [1.0, 1.0], [1.0, 0.0]]
    assert controlnet_keep(4, [0.0, 0.25], 0.5) == [[1.0, 0.0], [1.0, 1.0], [0.0, 0.0], [0.0, 0.0]]
    with pytest.raises(ValueError):
        controlnet_keep(4, [0.0, 0.1], [1.0])
