"""Prompt-to-prompt attention controller — drop-in for the reference's ``AttentionReplace``
(/root/reference src/prompt_attention/p2p_attention.py:141-147 and its bases :11-138), boundary B2
of SURVEY.md §8b: same constructor, ``reset()``, ``num_att_layers`` (set by the installer),
``cur_step`` / ``cur_att_layer`` counters, ``batch_size``, ``mapper``, ``cross_replace_alpha``,
``num_self_replace``, ``width``/``height``, and ``__call__(attn, is_cross, place_in_unet)`` editing
the conditional half of a probability tensor in place.

On top of that protocol the controller exposes what the fused kernel needs:

* ``is_pure_replacement`` — True when mapper == I, alpha == 1 for every step and no local blend:
  the edit degenerates to ``probs[cond_i] := probs[cond_0]``.
* ``fused_qk_src(is_cross, n_tokens, batch, place)`` — advances the layer/step counters exactly like
  ``__call__`` and returns the per-sample "borrow Q,K from" index vector (device int32) for
  ``omg_attn_fwd``, or None when this call leaves the probabilities untouched.

The counters live on the host (they select which pre-built index tensor / captured graph is used);
nothing here synchronises with the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

MAX_NUM_WORDS = 77


def get_word_inds(text: str, word_place, tokenizer) -> np.ndarray:
    """Token indices (BOS = 0) covered by a word position / word string (seq_aligner.py:5-23)."""
    split_text = text.split(" ")
    if isinstance(word_place, str):
        word_place = [i for i, word in enumerate(split_text) if word_place == word]
    elif isinstance(word_place, int):
        word_place = [word_place]
    out = []
    if len(word_place) > 0:
        decoded = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
        run, ptr = 0, 0
        for idx, piece in enumerate(decoded):
            run += len(piece)
            if ptr in word_place:
                out.append(idx + 1)
            if run >= len(split_text[ptr]):
                ptr, run = ptr + 1, 0
    return np.array(out)


def get_replacement_mapper(prompts: Sequence[str], tokenizer, max_len: int = MAX_NUM_WORDS) -> torch.Tensor:
    """(len(prompts)-1, 77, 77) token-remap matrices (seq_aligner.py:25-66).  Equal prompts give the
    identity without consulting the tokenizer; a word-count mismatch raises ValueError like the reference."""
    eye = torch.eye(max_len, dtype=torch.float32)
    mappers = []
    for other in prompts[1:]:
        wa, wb = prompts[0].split(" "), other.split(" ")
        if len(wa) != len(wb):
            raise ValueError("attention replacement edit can only be applied on prompts with the same length"
                             f" but prompt A has {len(wa)} words and prompt B has {len(wb)} words.")
        diff = [i for i in range(len(wb)) if wb[i] != wa[i]]
        if not diff:
            mappers.append(eye.clone())
            continue
        if tokenizer is None:
            raise ValueError("a tokenizer is required to align prompts that differ")
        src = [get_word_inds(prompts[0], i, tokenizer) for i in diff]
        tgt = [get_word_inds(other, i, tokenizer) for i in diff]
        m = np.zeros((max_len, max_len))
        i = j = n = 0
        while i < max_len and j < max_len:
            if n < len(src) and src[n][0] == i:
                s, t = src[n], tgt[n]
                if len(s) == len(t):
                    m[s, t] = 1
                else:
                    for col in t:
                        m[s, col] = 1 / len(t)
                n += 1
                i += len(s)
                j += len(t)
            elif n < len(src):
                m[i, j] = 1
                i, j = i + 1, j + 1
            else:
                m[j, j] = 1
                i, j = i + 1, j + 1
        mappers.append(torch.from_numpy(m).float())
    return torch.stack(mappers)


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=MAX_NUM_WORDS):
    """(num_steps+1, len(prompts)-1, 1, 1, 77) blend schedule (p2p_utils.py:23-33, :55-73)."""
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0.0, 1.0)
    n_edit = len(prompts) - 1
    alpha = torch.zeros(num_steps + 1, n_edit, max_num_words)

    def window(bounds, k, words=None):
        if isinstance(bounds, float):
            bounds = (0, bounds)
        a, b = int(bounds[0] * alpha.shape[0]), int(bounds[1] * alpha.shape[0])
        words = torch.arange(alpha.shape[2]) if words is None else words
        alpha[:a, k, words] = 0
        alpha[a:b, k, words] = 1
        alpha[b:, k, words] = 0

    for k in range(n_edit):
        window(cross_replace_steps["default_"], k)
    for word, bounds in cross_replace_steps.items():
        if word == "default_":
            continue
        for k in range(n_edit):
            inds = get_word_inds(prompts[k + 1], word, tokenizer)
            if len(inds) > 0:
                window(bounds, k, torch.as_tensor(inds))
    return alpha.reshape(num_steps + 1, n_edit, 1, 1, max_num_words)


class AttentionReplace:
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, width, height,
                 local_blend=None, tokenizer=None, device=None, dtype=None):
        self.low_resource = False
        self.width, self.height = width, height
        self.batch_size = len(prompts)
        self.local_blend = local_blend
        self.device, self.dtype = device, dtype
        self.cross_replace_alpha = get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer).to(device)
        if isinstance(self_replace_steps, float):
            self_replace_steps = (0, self_replace_steps)
        self.num_self_replace = (int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1]))
        self.mapper = get_replacement_mapper(prompts, tokenizer).to(dtype=dtype, device=device)
        self.num_att_layers = -1
        self.cur_step = 0
        self.cur_att_layer = 0
        eye = torch.eye(MAX_NUM_WORDS)
        self.is_pure_replacement = bool(
            local_blend is None
            and all(torch.equal(m.float().cpu(), eye) for m in self.mapper)
            and bool((self.cross_replace_alpha == 1).all())
        )
        self._src_cache: Dict[Tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------ protocol (B2)
    @property
    def num_uncond_att_layers(self):
        return 0

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def between_steps(self):
        return

    def step_callback(self, x_t):
        return x_t

    def _tick(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    def _self_window(self) -> bool:
        return self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]

    def replace_self_attention(self, attn_base, att_replace):
        if att_replace.shape[2] <= self.width * self.height:
            return attn_base.unsqueeze(0).expand(att_replace.shape[0], *attn_base.shape)
        return att_replace

    def replace_cross_attention(self, attn_base, att_replace):
        return torch.einsum("hpw,bwn->bhpn", attn_base, self.mapper.to(attn_base.dtype).to(attn_base.device))

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        if is_cross or self._self_window():
            h = attn.shape[0] // self.batch_size
            attn = attn.reshape(self.batch_size, h, *attn.shape[1:])
            base, edit = attn[0], attn[1:]
            if is_cross:
                a = self.cross_replace_alpha[self.cur_step].to(attn.dtype).to(attn.device)
                attn[1:] = self.replace_cross_attention(base, edit) * a + (1 - a) * edit
            else:
                attn[1:] = self.replace_self_attention(base, edit)
            attn = attn.reshape(self.batch_size * h, *attn.shape[2:])
        return attn

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        h = attn.shape[0]
        attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place_in_unet)
        self._tick()
        return attn

    # ------------------------------------------------------------------ fused path
    def replaces(self, is_cross: bool, n_tokens: int) -> bool:
        """Does the edit at the current counters change the conditional half? (host-side, no sync)"""
        if is_cross:
            return True
        return self._self_window() and n_tokens <= self.width * self.height

    def qk_src_vector(self, batch: int, device, total_batch: Optional[int] = None, images: int = 1) -> torch.Tensor:
        """Per request [0..n-1 | n, n, ...]: every conditional sample borrows Q,K from the first conditional one.
        ``batch`` = 2 * len(prompts) in the reference's layout [unc_0..unc_{n-1}, cond_0..cond_{n-1}]; ``images``
        such blocks follow each other; samples beyond them (concept passes batched behind) keep their own Q,K."""
        total = total_batch or batch * images
        key = (batch, total, images, str(device))
        t = self._src_cache.get(key)
        if t is None:
            n = batch // 2
            v = []
            for j in range(images):
                v += [j * batch + i for i in range(n)] + [j * batch + n] * (batch - n)
            v += list(range(batch * images, total))
            t = torch.tensor(v, dtype=torch.int32, device=device)
            self._src_cache[key] = t
        return t

    def fused_qk_src(self, is_cross: bool, n_tokens: int, batch: int, place_in_unet: str = "",
                     device=None, total_batch: Optional[int] = None, images: int = 1) -> Optional[torch.Tensor]:
        if not self.is_pure_replacement:
            raise RuntimeError("fused_qk_src needs a pure-replacement controller (identity mapper, alpha == 1)")
        if batch != 2 * self.batch_size:
            raise ValueError(f"controller built for {self.batch_size} prompts expects a batch of {2 * self.batch_size} "
                             f"([unc..., cond...]), got {batch}")
        src = self.qk_src_vector(batch, device or self.device or "cuda", total_batch, images) if self.replaces(is_cross, n_tokens) else None
        self._tick()
        return src
