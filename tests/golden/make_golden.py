"""Generate golden vectors by executing the REFERENCE's own code in this container.

    python tests/golden/make_golden.py        (needs /root/reference; writes *.npz next to itself)

What can be imported from the reference: ``src/prompt_attention/*`` (with a stub
``cv2`` module — p2p_utils.py:18 imports it and never uses it), ``src/ip_adapter/*`` and — with bare package objects in place of
the ``__init__`` files that import torchvision — ``src/efficientvit/models/nn/ops.py`` (LiteMLA).
``src/pipelines/*`` cannot (diffusers/peft are not installed), so no pipeline-level vectors exist.
The fixtures pin: the oracle's controller restatement (facts T1-T6 + random tensors, including a
non-identity mapper), and the IP-Adapter cross-attention processor arithmetic.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

from oracle.controller import PieceTokenizer, WhitespaceTokenizer  # tokenizers only (no CLIP vocab offline)
from src.prompt_attention.p2p_attention import AttentionReplace  # noqa: E402  (reference code)
from src.prompt_attention import seq_aligner  # noqa: E402
from src.ip_adapter.attention_processor import IPAttnProcessor2_0, AttnProcessor2_0  # noqa: E402
from src.ip_adapter.resampler import Resampler  # noqa: E402


def controller_vectors():
    out = {}
    P = "a man and a woman walking on the street"
    g = torch.Generator().manual_seed(1234)
    # --- T1: construction with the CLI's arguments (inference_lora.py:156)
    c = AttentionReplace([P, P], 50, cross_replace_steps={"default_": 1.0}, self_replace_steps=0.4, width=32,
                         height=32, tokenizer=WhitespaceTokenizer(), device="cpu", dtype=torch.float32)
    out["t1_mapper"] = c.mapper.numpy()
    out["t1_alpha"] = c.cross_replace_alpha.numpy()
    out["t1_num_self_replace"] = np.array(c.num_self_replace)
    out["t1_batch_size"] = np.array(c.batch_size)
    # --- T2-T4: calls on random probability tensors (heads = 3, main batch layout [unc0,unc1,cond0,cond1])
    h = 3
    c.num_att_layers = 4
    cases = [("cross_q64", True, 64, 77, 0, 3), ("self_q1024_s0", False, 1024, 8, 0, 1), ("self_q1056_s0", False, 1056, 8, 0, 1),
             ("self_q1024_s19", False, 1024, 8, 19, 1), ("self_q1024_s20", False, 1024, 8, 20, 1)]
    for name, is_cross, q, k, step, h in cases:
        probs = torch.softmax(torch.randn(4 * h, q, k, generator=g), dim=-1)
        out[f"{name}_in"] = probs.numpy().copy()
        c.reset()
        c.cur_step = step
        res = c(probs, is_cross, "down")
        assert res is probs  # in place (T2)
        out[f"{name}_out"] = res.numpy().copy()
    # --- T5: counters after exactly num_att_layers calls
    c.reset()
    c.num_att_layers = 140
    tiny = torch.softmax(torch.randn(4, 2, 77, generator=g), dim=-1)
    for _ in range(140):
        c(tiny, True, "down")
    out["t5_counters"] = np.array([c.cur_step, c.cur_att_layer])
    # --- T6 and beyond: mappers for unequal prompts
    out["t6_mapper_swap"] = seq_aligner.get_replacement_mapper(["a man on the street", "a dog on the street"], WhitespaceTokenizer()).numpy()
    try:
        seq_aligner.get_replacement_mapper(["a man", "a man walking"], WhitespaceTokenizer())
        out["t6_raises"] = np.array(0)
    except ValueError:
        out["t6_raises"] = np.array(1)
    prompts = ["a man on the road", "a woman on the road"]     # "woman" -> 2 pieces with PieceTokenizer
    out["mapper_pieces"] = seq_aligner.get_replacement_mapper(prompts, PieceTokenizer()).numpy()
    # --- general path: non-identity mapper + partial cross-replace window + word-specific window
    c2 = AttentionReplace(prompts, 10, cross_replace_steps={"default_": 0.6, "road": (0.2, 0.9)}, self_replace_steps=(0.1, 0.5),
                          width=4, height=4, tokenizer=PieceTokenizer(), device="cpu", dtype=torch.float32)
    out["gen_alpha"] = c2.cross_replace_alpha.numpy()
    out["gen_mapper"] = c2.mapper.numpy()
    out["gen_num_self_replace"] = np.array(c2.num_self_replace)
    c2.num_att_layers = 2
    for step in (0, 3, 7):
        for is_cross, q, k in ((True, 16, 77), (False, 16, 16)):
            probs = torch.softmax(torch.randn(4 * 2, q, k, generator=g), dim=-1)
            key = f"gen_s{step}_{'cross' if is_cross else 'self'}"
            out[key + "_in"] = probs.numpy().copy()
            c2.reset()
            c2.cur_step = step
            out[key + "_out"] = c2(probs, is_cross, "mid").numpy().copy()
    np.savez_compressed(os.path.join(HERE, "controller_golden.npz"), **out)
    print("controller_golden.npz:", len(out), "arrays")


def controller_alignment_vectors():
    """Round 3: harder alignment cases for the product's own (run-offset) construction of the mapper / alpha tables — several edited
    prompts, several replaced words per prompt, replaced words with different piece counts on either side, word-specific
    windows that hit more than one token — all produced by the reference's seq_aligner / p2p_utils."""
    from src.prompt_attention import p2p_utils
    out = {}
    cases = [
        (["a man on the road", "a woman on the road", "a superman on the road"], "piece"),
        (["photograph of a man walking the dog", "photograph of a woman walking the cat"], "piece"),
        (["extraordinarily dog in the garden", "x superman in the street"], "piece"),
        (["a man and a woman walking on the street", "a dog and a cat walking on the street"], "white"),
        (["woman woman woman", "man x woman"], "piece"),
    ]
    for n, (prompts, tk) in enumerate(cases):
        tok = PieceTokenizer() if tk == "piece" else WhitespaceTokenizer()
        out[f"c{n}_mapper"] = seq_aligner.get_replacement_mapper(prompts, tok).numpy()
        for m, (S, spec) in enumerate([(50, {"default_": 1.0}), (10, {"default_": 0.6, prompts[1].split(" ")[-1]: (0.2, 0.9)}),
                                       (7, {"default_": (0.1, 0.8), prompts[1].split(" ")[0]: 0.3})]):
            out[f"c{n}_alpha{m}"] = p2p_utils.get_time_words_attention_alpha(prompts, S, dict(spec), tok).numpy()
    # round 4: prompts LONGER than the 77-token window — `tokenizer.encode` does not truncate, the reference's two-pointer walk
    # simply stops at 77.  Replaced word exactly at token 77 (the first position outside), beyond it, and just inside.
    for n, w in enumerate(LONG_REPLACED_WORDS):
        a, b = long_prompt_pair(w)
        out[f"long{n}_mapper"] = seq_aligner.get_replacement_mapper([a, b], WhitespaceTokenizer()).numpy()
    np.savez_compressed(os.path.join(HERE, "controller_alignment_golden.npz"), **out)
    print("controller_alignment_golden.npz:", len(out), "arrays")


LONG_REPLACED_WORDS = (76, 80, 75, 74)        # word index k is token k + 1 for the whitespace tokenizer (BOS at 0)


def long_prompt_pair(k: int, n_words: int = 86):
    """two prompts of n_words one-token words that differ in word k only"""
    words = [f"w{i}" for i in range(n_words)]
    other = list(words)
    other[k] = "changed"
    return " ".join(words), " ".join(other)


class _FakeAttn(torch.nn.Module):
    """The slice of diffusers' Attention interface the IP-Adapter processors touch."""

    def __init__(self, C, ctx, heads, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.heads = heads
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(ctx, C, bias=False)
        self.to_v = torch.nn.Linear(ctx, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0


def ip_adapter_vectors():
    out = {}
    C, ctx, heads, ntok = 128, 96, 2, 16
    attn = _FakeAttn(C, ctx, heads, seed=7)
    proc = IPAttnProcessor2_0(hidden_size=C, cross_attention_dim=ctx, scale=0.8, num_tokens=ntok)
    torch.manual_seed(8)
    torch.nn.init.normal_(proc.to_k_ip.weight, std=ctx ** -0.5)
    torch.nn.init.normal_(proc.to_v_ip.weight, std=ctx ** -0.5)
    g = torch.Generator().manual_seed(9)
    hs = torch.randn(2, 40, C, generator=g)
    ehs = torch.randn(2, 77 + ntok, ctx, generator=g)
    with torch.no_grad():
        y = proc(attn, hs, ehs)
        y_self = AttnProcessor2_0()(_FakeAttn(C, C, heads, seed=7), hs)
    for k, v in attn.state_dict().items():
        out["attn." + k] = v.numpy()
    out["to_k_ip"] = proc.to_k_ip.weight.detach().numpy()
    out["to_v_ip"] = proc.to_v_ip.weight.detach().numpy()
    out["hidden_states"], out["encoder_hidden_states"] = hs.numpy(), ehs.numpy()
    out["out_cross"] = y.numpy()
    self_attn = _FakeAttn(C, C, heads, seed=7)
    for k, v in self_attn.state_dict().items():
        out["self_attn." + k] = v.numpy()
    out["out_self"] = y_self.numpy()
    out["meta"] = np.array([C, ctx, heads, ntok])
    np.savez_compressed(os.path.join(HERE, "ip_adapter_golden.npz"), **out)
    print("ip_adapter_golden.npz:", len(out), "arrays")


def resampler_vectors():
    """The reference's own Resampler (InstantID image_proj_model topology at reduced width) on seeded weights."""
    torch.manual_seed(21)
    m = Resampler(dim=64, depth=2, dim_head=32, heads=2, num_queries=8, embedding_dim=32, output_dim=48, ff_mult=4).eval()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n.endswith("bias"):
                p_.copy_(0.1 * torch.randn_like(p_))
            elif "norm" in n or n.endswith(".0.weight"):
                p_.copy_(1.0 + 0.1 * torch.randn_like(p_))
        x = torch.randn(3, 2, 32)
        y = m(x)
    out = {"cfg": np.array([64, 2, 32, 2, 8, 32, 48, 4]), "x": x.numpy(), "y": y.numpy()}
    for k, v in m.state_dict().items():
        out["sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "resampler_golden.npz"), **out)
    print("resampler_golden.npz", {k: v.shape for k, v in out.items() if not k.startswith("sd.")})


def litemla_vectors():
    """EfficientViT LiteMLA (src/efficientvit/models/nn/ops.py:335-455), run by the reference's own class in eval mode.  The
    package __init__ files of src/efficientvit import torchvision (not installed here); bare package objects are registered so
    that only models/utils/*, models/nn/act.py, norm.py and ops.py are executed."""
    import importlib
    for name, sub in [("src.efficientvit", "src/efficientvit"), ("src.efficientvit.models", "src/efficientvit/models"),
                      ("src.efficientvit.models.nn", "src/efficientvit/models/nn")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, sub)]
            sys.modules[name] = m
    ref_ops = importlib.import_module("src.efficientvit.models.nn.ops")
    from oracle import litemla as ol
    out = {}
    cases = [("c64_d16", 64, 64, 16, (5,), 2, 8, 8, 11), ("c128_d32", 128, 96, 32, (5,), 1, 12, 10, 12), ("c64_d16_s35", 64, 64, 16, (3, 5), 2, 6, 6, 13)]
    for tag, cin, cout, dim, scales, B, H, W, seed in cases:
        m = ref_ops.LiteMLA(cin, cout, dim=dim, scales=scales).eval()
        sd = ol.init_state_dict(cin, cout, dim, scales, seed=seed)
        res = m.load_state_dict(sd, strict=False)
        assert all(k.endswith("num_batches_tracked") for k in res.missing_keys) and not res.unexpected_keys, res
        x = torch.randn(B, cin, H, W, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            y = m(x)
        out[f"{tag}_x"], out[f"{tag}_y"] = x.numpy(), y.numpy()
        out[f"{tag}_cfg"] = np.array([cin, cout, dim, B, H, W] + list(scales))
        for k, v in sd.items():
            out[f"{tag}_sd_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "litemla_golden.npz"), **out)
    print("litemla_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    litemla_vectors()
    controller_vectors()
    controller_alignment_vectors()
    ip_adapter_vectors()
    resampler_vectors()
