"""Row N4: CLIP text encoders on the HIP kernels vs oracle/text_encoder.py (itself pinned against transformers,
tests/test_oracle_text.py).  Tolerance: 16-bit residual stream through a few pre-LN layers, max |d| / output rms."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omg_amd.text_encoder import ClipTextConfig, ClipTextEncoder, encode_prompt
from oracle import text_encoder as ot


def _pair(act, proj, dev, dtype, seed):
    ocfg = ot.ClipTextConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                             hidden_act=act, projection_dim=64, eos_token_id=299)
    cfg = ClipTextConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         hidden_act=act, projection_dim=64, eos_token_id=299, with_projection=proj)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in ot.param_shapes(ocfg, with_projection=proj).items():
        if k.endswith("layer_norm1.weight") or k.endswith("layer_norm2.weight") or k.endswith("final_layer_norm.weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.1 * torch.randn(shp, generator=g)
        elif "embedding" in k:
            w = 0.5 * torch.randn(shp, generator=g)
        else:
            w = torch.randn(shp, generator=g) * shp[-1] ** -0.5
        sd[k] = w.to(dtype)
    enc = ClipTextEncoder(cfg, dtype=dtype, device=dev)
    enc.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    return ocfg, {k: v.float() for k, v in sd.items()}, enc


def _ids(B, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(2, 298, (B, 77), generator=g)
    ids[:, 0] = 298
    for b, n in enumerate([4, 40, 76][:B]):
        ids[b, n:] = 299
    return ids


@pytest.mark.parametrize("act,proj", [("quick_gelu", False), ("gelu", True)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_text_encoder_matches_oracle(dev, dtype, act, proj):
    ocfg, sd, enc = _pair(act, proj, dev, dtype, seed=0)
    ids = _ids(3, 1)
    hidden, last, pooled = ot.text_model(sd, ocfg, ids)
    h, p = enc(ids.to(dev))
    tol = 2e-2 if dtype == torch.float16 else 1e-1
    ref_h = hidden[-2]
    assert h.shape == ref_h.shape and p.shape == pooled.shape
    assert (h.float().cpu() - ref_h).abs().max() / ref_h.pow(2).mean().sqrt() < tol
    assert (p.float().cpu() - pooled).abs().max() / pooled.pow(2).mean().sqrt() < tol


def test_encode_prompt_concatenates_both_encoders(dev):
    _, _, enc_l = _pair("quick_gelu", False, dev, torch.float16, seed=2)
    _, _, enc_g = _pair("gelu", True, dev, torch.float16, seed=3)
    ids = _ids(2, 4).to(dev)
    emb, pooled = encode_prompt(enc_l, enc_g, ids, ids)
    assert emb.shape == (2, 77, 256) and pooled.shape == (2, 64) and torch.isfinite(emb).all()
    assert torch.equal(emb[..., :128], enc_l(ids)[0])


def test_make_encode_prompt_contract(dev):
    from omg_amd.text_encoder import make_encode_prompt
    _, _, enc_l = _pair("quick_gelu", False, dev, torch.float16, seed=2)
    _, _, enc_g = _pair("gelu", True, dev, torch.float16, seed=3)

    def tok(texts):                                      # stand-in for the CLIP tokenizer: bos, one id per word, eos padding
        out = torch.full((len(texts), 77), 299, dtype=torch.long)
        out[:, 0] = 298
        for b, t in enumerate(texts):
            for i, w in enumerate(t.split()[:75]):
                out[b, 1 + i] = 2 + sum(map(ord, w)) % 296
        return out

    fn = make_encode_prompt(enc_l, enc_g, tok)
    pe, ne, pp, npp = fn(["a man and a woman", "a man and a woman"], "blurry", None)
    assert pe.shape == ne.shape == (2, 77, 256) and pp.shape == npp.shape == (2, 64)
    assert torch.equal(pe[0], pe[1]) and torch.equal(ne[0], ne[1]) and not torch.equal(pe, ne)
    ids = tok(["a man and a woman"]).to(dev)
    assert torch.equal(pe[:1], encode_prompt(enc_l, enc_g, ids, ids)[0])
    pe2, ne2, pp2, npp2 = fn("a man and a woman")
    assert torch.equal(pe2, pe[:1]) and float(ne2.abs().max()) == 0.0 and float(npp2.abs().max()) == 0.0
