"""Row N4 groundwork: the text-encoder oracle is PINNED against transformers' CLIPTextModelWithProjection (importable here),
seeded random weights, both activation variants used by SDXL's two encoders."""
import pytest
import torch

from oracle import text_encoder as ot

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_encoder_oracle_matches_transformers(act):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    cfg = ot.ClipTextConfig(vocab_size=500, hidden_size=64, intermediate_size=160, num_hidden_layers=3, num_attention_heads=4,
                            hidden_act=act, projection_dim=48, eos_token_id=499)
    tcfg = CLIPTextConfig(vocab_size=500, hidden_size=64, intermediate_size=160, num_hidden_layers=3, num_attention_heads=4,
                          max_position_embeddings=77, hidden_act=act, projection_dim=48, eos_token_id=499, bos_token_id=498,
                          pad_token_id=1, layer_norm_eps=1e-5)
    torch.manual_seed(0)
    ref = CLIPTextModelWithProjection(tcfg).eval()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    assert set(ot.param_shapes(cfg)) == {k for k in sd if "position_ids" not in k}
    assert all(tuple(sd[k].shape) == s for k, s in ot.param_shapes(cfg).items())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(2, 498, (3, 77), generator=g)
    ids[:, 0] = 498
    for b, n in enumerate((5, 30, 76)):
        ids[b, n:] = 499                                  # EOS then EOS padding, like the CLIP tokenizer of SDXL's second encoder
    with torch.no_grad():
        out = ref(ids, output_hidden_states=True)
    hidden, last, pooled = ot.text_model(sd, cfg, ids)
    assert len(hidden) == len(out.hidden_states) == 4
    for a, b in zip(hidden, out.hidden_states):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(last, out.last_hidden_state, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pooled, out.text_embeds, rtol=1e-5, atol=1e-5)


def test_sdxl_encoder_shapes():
    l, g = ot.ClipTextConfig.clip_l(), ot.ClipTextConfig.open_clip_bigg()
    import math
    n_l = sum(math.prod(s) for s in ot.param_shapes(l, with_projection=False).values())
    n_g = sum(math.prod(s) for s in ot.param_shapes(g, with_projection=True).values())
    assert n_l == 123_060_480 and n_g == 694_659_840          # the published sizes of SDXL's text_encoder / text_encoder_2
    assert l.hidden_size + g.hidden_size == 2048             # = the UNet's cross_attention_dim
