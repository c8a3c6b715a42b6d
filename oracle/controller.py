"""ORACLE (test infrastructure, never imported by omg_amd): CPU restatement of OMG's
prompt-to-prompt attention controller and of the attention-processor sequence that calls it.

Follows, by file:line of /root/reference:
  * ``AttentionControl.__call__``           src/prompt_attention/p2p_attention.py:28-40
  * ``AttentionControlEdit.forward``        src/prompt_attention/p2p_attention.py:124-138
  * ``replace_self_attention``              src/prompt_attention/p2p_attention.py:114-118
  * ``AttentionReplace.replace_cross_attention`` src/prompt_attention/p2p_attention.py:146-147
  * ``get_time_words_attention_alpha``      src/prompt_attention/p2p_utils.py:55-73 (+ :23-33, :35-53)
  * ``get_replacement_mapper``              src/prompt_attention/seq_aligner.py:25-66
  * ``RegionControlNet_AttnProcessor.__call__`` src/pipelines/lora_pipeline.py:98-121 (q/k/v are
    projected by the caller, oracle/unet.py)

PINNED: tests/golden/controller_golden.npz was generated in this container by importing the
reference's own ``src/prompt_attention`` (tests/golden/make_golden.py); tests/test_oracle.py
checks this restatement against it bit-for-bit (facts T1-T6 of SURVEY.md §4.3 and random
probability tensors).  Round 5: ``reference_attn_fn`` + ``AttentionReplaceOracle`` inside a whole two-stage loop are compared with the
reference's own ``RegionControlNet_AttnProcessor`` + ``AttentionReplace`` executing at all 34 attention layers of the same loop
(tests/golden/make_golden_loop.py -> tests/test_oracle_loop.py: stage 1 bit-equal, stage 2 within one fp32 ulp).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .precision import r as _r

MAX_WORDS = 77


class WhitespaceTokenizer:
    """Stand-in for CLIPTokenizer (no vocabulary is available offline): one id per
    distinct word, BOS=0, EOS=1.  Only ``encode``/``decode`` are used by the aligner."""

    def __init__(self):
        self.vocab: Dict[str, int] = {}
        self.inv: Dict[int, str] = {0: "<s>", 1: "</s>"}

    def encode(self, text: str) -> List[int]:
        ids = [0]
        for w in text.split(" "):
            if w not in self.vocab:
                self.vocab[w] = len(self.vocab) + 2
                self.inv[self.vocab[w]] = w
            ids.append(self.vocab[w])
        return ids + [1]

    def decode(self, ids: Sequence[int]) -> str:
        return " ".join(self.inv[int(i)] for i in ids)


class PieceTokenizer(WhitespaceTokenizer):
    """Like WhitespaceTokenizer but words longer than 4 characters become TWO tokens
    ("street" -> "str" + "eet"), so that swapping a short word for a long one yields a
    non-identity, non-square replacement mapper (the general A5 path)."""

    def encode(self, text: str) -> List[int]:
        ids = [0]
        for w in text.split(" "):
            pieces = [w] if len(w) <= 4 else [w[: len(w) // 2], w[len(w) // 2:]]
            for pc in pieces:
                if pc not in self.vocab:
                    self.vocab[pc] = len(self.vocab) + 2
                    self.inv[self.vocab[pc]] = pc
                ids.append(self.vocab[pc])
        return ids + [1]


def word_token_indices(text: str, word_place, tokenizer) -> np.ndarray:
    """Token positions (1-based, BOS at 0) of the word(s) `word_place` — seq_aligner.py:5-23."""
    words = text.split(" ")
    if isinstance(word_place, str):
        places = [i for i, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, int):
        places = [word_place]
    else:
        places = list(word_place)
    out: List[int] = []
    if places:
        pieces = [tokenizer.decode([t]).strip("#") for t in tokenizer.encode(text)][1:-1]
        consumed, ptr = 0, 0
        for i, piece in enumerate(pieces):
            consumed += len(piece)
            if ptr in places:
                out.append(i + 1)
            if consumed >= len(words[ptr]):
                ptr += 1
                consumed = 0
    return np.array(out)


def replacement_mapper(prompts: Sequence[str], tokenizer, max_len: int = MAX_WORDS) -> torch.Tensor:
    """(n_prompts-1, 77, 77) token remap; identity when the prompts are equal — seq_aligner.py:25-66."""
    base = prompts[0]
    mats = []
    for other in prompts[1:]:
        wx, wy = base.split(" "), other.split(" ")
        if len(wx) != len(wy):
            raise ValueError("attention replacement edit can only be applied on prompts with the same length"
                             f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
        changed = [i for i in range(len(wy)) if wy[i] != wx[i]]
        src = [word_token_indices(base, i, tokenizer) for i in changed]
        tgt = [word_token_indices(other, i, tokenizer) for i in changed]
        m = np.zeros((max_len, max_len))
        i = j = cur = 0
        while i < max_len and j < max_len:
            if cur < len(src) and src[cur][0] == i:
                s, t = src[cur], tgt[cur]
                if len(s) == len(t):
                    m[s, t] = 1
                else:
                    for it in t:
                        m[s, it] = 1 / len(t)
                cur += 1
                i += len(s)
                j += len(t)
            elif cur < len(src):
                m[i, j] = 1
                i += 1
                j += 1
            else:
                m[j, j] = 1
                i += 1
                j += 1
        mats.append(torch.from_numpy(m).float())
    return torch.stack(mats)


def time_words_alpha(prompts: Sequence[str], num_steps: int, cross_replace_steps, tokenizer,
                     max_words: int = MAX_WORDS) -> torch.Tensor:
    """(num_steps+1, n_prompts-1, 1, 1, 77) schedule — p2p_utils.py:55-73."""
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0.0, 1.0)
    n = len(prompts) - 1
    alpha = torch.zeros(num_steps + 1, n, max_words)

    def apply(bounds, prompt_ind, word_inds=None):
        if isinstance(bounds, float):
            bounds = (0, bounds)
        lo, hi = int(bounds[0] * alpha.shape[0]), int(bounds[1] * alpha.shape[0])
        if word_inds is None:
            word_inds = torch.arange(alpha.shape[2])
        alpha[:lo, prompt_ind, word_inds] = 0
        alpha[lo:hi, prompt_ind, word_inds] = 1
        alpha[hi:, prompt_ind, word_inds] = 0

    for i in range(n):
        apply(cross_replace_steps["default_"], i)
    for key, bounds in cross_replace_steps.items():
        if key == "default_":
            continue
        for i in range(n):
            inds = word_token_indices(prompts[i + 1], key, tokenizer)
            if len(inds) > 0:
                apply(bounds, i, torch.as_tensor(inds))
    return alpha.reshape(num_steps + 1, n, 1, 1, max_words)


class AttentionReplaceOracle:
    """State machine + in-place edit of the conditional half of the probabilities."""

    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, width: int, height: int,
                 tokenizer=None):
        tokenizer = tokenizer or WhitespaceTokenizer()
        self.batch_size = len(prompts)
        self.width, self.height = width, height
        self.cross_replace_alpha = time_words_alpha(prompts, num_steps, cross_replace_steps, tokenizer)
        if isinstance(self_replace_steps, float):
            self_replace_steps = (0, self_replace_steps)
        self.num_self_replace = (int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1]))
        self.mapper = replacement_mapper(prompts, tokenizer)
        self.num_att_layers = -1
        self.reset()

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def _edit(self, attn: torch.Tensor, is_cross: bool) -> torch.Tensor:
        lo, hi = self.num_self_replace
        if not (is_cross or lo <= self.cur_step < hi):
            return attn
        h = attn.shape[0] // self.batch_size
        a = attn.reshape(self.batch_size, h, *attn.shape[1:])
        base, edit = a[0], a[1:]
        if is_cross:
            al = self.cross_replace_alpha[self.cur_step].to(attn.dtype)
            mapped = torch.einsum("hpw,bwn->bhpn", base, self.mapper.to(attn.dtype))
            a[1:] = mapped * al + (1 - al) * edit
        elif edit.shape[2] <= self.width * self.height:
            a[1:] = base.unsqueeze(0).expand(edit.shape[0], *base.shape)
        return a.reshape(self.batch_size * h, *a.shape[2:])

    def __call__(self, attn: torch.Tensor, is_cross: bool, place_in_unet: str) -> torch.Tensor:
        h = attn.shape[0]
        attn[h // 2:] = self._edit(attn[h // 2:], is_cross)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
        return attn


def reference_attn_fn(controller):
    """attn_fn for oracle.unet.unet_forward reproducing RegionControlNet_AttnProcessor:
    head_to_batch_dim -> get_attention_scores (softmax(QK^T*scale)) -> controller -> bmm."""

    def fn(name: str, heads: int, q, k, v, is_cross: bool):
        B, N, C = q.shape
        d = C // heads

        def h2b(t):
            return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)

        qb, kb, vb = h2b(q), h2b(k), h2b(v)
        # oracle.precision: identity by default; under rounding(fp16) the scores, the probabilities and the output are STORED in fp16
        # as get_attention_scores' baddbmm / softmax and the processor's bmm store them (lora_pipeline.py:114-116)
        probs = _r(torch.softmax(_r(torch.baddbmm(torch.empty(()), qb, kb.transpose(-1, -2), beta=0, alpha=d ** -0.5)), dim=-1))
        probs = controller(probs, is_cross, "unet")
        o = _r(torch.bmm(probs, vb))
        return o.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)

    return fn
