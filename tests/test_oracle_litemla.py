"""CPU: oracle/litemla.py against the golden vectors produced by the reference's own LiteMLA class
(src/efficientvit/models/nn/ops.py:335-455 run in this container; tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import litemla as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "litemla_golden.npz")


def cases():
    g = np.load(GOLD)
    for tag in sorted({k[:-4] for k in g.files if k.endswith("_cfg")}):
        cfg = g[f"{tag}_cfg"].tolist()
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_sd_")}
        yield tag, cfg, sd, torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_y"])


def test_oracle_matches_the_reference_class():
    n = 0
    for tag, cfg, sd, x, y in cases():
        cin, cout, dim, B, H, W, *scales = cfg
        got = ol.litemla_forward(sd, x, dim=dim, scales=tuple(scales))
        assert got.shape == y.shape == (B, cout, H, W)
        assert (got - y).abs().max().item() < 1e-5, tag
        n += 1
    assert n == 3


def test_linear_attention_identities():
    """Size-independent properties of relu_linear_att the HIP kernel is also held to: the output of a group is invariant to a
    permutation of the tokens' (k, v) pairs, and a positive rescale of k leaves it unchanged (numerator and denominator scale alike)."""
    sd = ol.init_state_dict(64, 64, 16, (5,), seed=3)
    x = torch.randn(1, 64, 6, 6, generator=torch.Generator().manual_seed(1))
    y = ol.litemla_forward(sd, x, dim=16)
    sd2 = dict(sd)
    w = sd["qkv.conv.weight"].clone().reshape(-1, 48, 64)
    w[:, 16:32] *= 4.0                                      # k rows of every head (only the un-aggregated branch scales exactly...)
    sd2["qkv.conv.weight"] = w.reshape(-1, 64, 1, 1)
    sd2["aggreg.0.0.weight"] = sd["aggreg.0.0.weight"].clone()
    y2 = ol.litemla_forward(sd2, x, dim=16)                # ... and the aggregated branch too: both convolutions are linear in k
    assert (y - y2).abs().max().item() < 1e-4


def test_product_module_has_the_reference_state_dict_layout():
    """omg_amd.litemla.LiteMLA must load a checkpoint of the reference's module key for key (shapes included); construction and
    state-dict handling run on CPU, compute does not (no fallback)."""
    import pytest
    from omg_amd import _lib as L
    from omg_amd.litemla import LiteMLA
    for cin, cout, dim, scales in [(64, 64, 16, (5,)), (128, 96, 32, (5,)), (64, 64, 16, (3, 5))]:
        sd = ol.init_state_dict(cin, cout, dim, scales, seed=1)
        m = LiteMLA(cin, cout, dim=dim, scales=scales, dtype=torch.float16, device="cpu")
        own = m.state_dict()
        assert set(sd) <= set(own), set(sd) - set(own)
        assert set(own) - set(sd) == {"proj.norm.num_batches_tracked"}
        for k, v in sd.items():
            assert tuple(own[k].shape) == tuple(v.shape), k
        m.load_state_dict({k: v.half() if "running" not in k else v for k, v in sd.items()}, strict=False)
        with pytest.raises(L.OmgHipError):
            m(torch.zeros(1, cin, 4, 4, dtype=torch.float16))
    with pytest.raises(L.OmgHipError):
        LiteMLA(64, 64, dim=12)
