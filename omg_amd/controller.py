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

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

MAX_NUM_WORDS = 77


def _word_of_token(text: str, tokenizer) -> Tuple[List[str], np.ndarray]:
    """Which word each inner token (BOS / EOS dropped) of ``text`` belongs to.

    The rule being reproduced (seq_aligner.py:5-23, p2p_utils.py:35-53): tokens are handed to the words of ``text.split(" ")`` left to
    right, and a word is complete at the first token at which the characters collected for it reach the word's length (an
    over-long last piece is not carried over).  Stated on cumulative character offsets: with ``cum[t]`` = characters up to and
    including token t, word k ends at the first t with ``cum[t] >= cum[end of word k-1] + len(word k)`` — one ``searchsorted`` per word
    instead of a walk over the tokens.  Tokens left after the last word get the index ``len(words)`` (the reference would raise)."""
    words = text.split(" ")
    pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
    cum = np.cumsum([len(pc_) for pc_ in pieces], dtype=np.int64) if pieces else np.zeros(0, np.int64)
    owner = np.full(len(pieces), len(words), dtype=np.int64)
    first, base = 0, 0
    for k, word in enumerate(words):
        if first >= len(pieces):
            break
        last = int(np.searchsorted(cum, base + len(word), side="left"))      # first token reaching the word's length
        last = max(last, first)                                                # an empty word still consumes one token
        last = min(last, len(pieces) - 1)
        owner[first:last + 1] = k
        first, base = last + 1, int(cum[last])
    return words, owner


def get_word_inds(text: str, word_place, tokenizer) -> np.ndarray:
    """Token positions (BOS = 0) of a word given by position (int), by string (every occurrence) or by a list of positions."""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = [i for i, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, int):
        wanted = [word_place]
    else:
        wanted = list(word_place)
    if not wanted:
        return np.array([])
    _, owner = _word_of_token(text, tokenizer)
    return np.flatnonzero(np.isin(owner, wanted)) + 1


def _alignment_matrix(src_spans: Sequence[np.ndarray], tgt_spans: Sequence[np.ndarray], max_len: int) -> np.ndarray:
    """The (max_len, max_len) token-remap matrix of one prompt pair from the token spans of its replaced words
    (semantics of seq_aligner.py:25-58, built from run offsets instead of a two-pointer walk).

    Rows are source tokens, columns target tokens.  The sequences are cut at the replaced words into runs:
      * before replaced word k both prompts share their tokens, shifted by d_k = sum over earlier replacements of
        (target pieces - source pieces): M[i, i + d_k] = 1 for the run's rows;
      * a replaced word with equally many pieces pairs them one to one; otherwise every source piece spreads uniformly
        over the target pieces (weight 1 / #target pieces);
      * after the LAST replaced word the matrix continues with ones on the diagonal of the TARGET index (M[j, j], the
        reference's tail rule), for as long as both positions stay below max_len."""
    m = np.zeros((max_len, max_len))
    i = j = 0                                             # next unassigned source / target position
    for s, t in zip(src_spans, tgt_spans):
        s, t = np.asarray(s, dtype=np.int64), np.asarray(t, dtype=np.int64)
        run = min(int(s[0]) - i, max_len - i, max_len - j)      # shared tokens in front of this replacement
        if run < int(s[0]) - i:                                 # the window closes before the replacement is reached
            r = np.arange(max(run, 0))
            m[i + r, j + r] = 1
            return m
        r = np.arange(run)
        m[i + r, j + r] = 1
        i, j = i + run, j + run
        if i >= max_len or j >= max_len:                        # the replacement starts exactly where the window ends (prompts longer than
            return m                                            # 77 tokens: the reference's `while i < max_len and j < max_len` stops here)
        if len(s) == len(t):
            m[s, t] = 1
        else:
            m[np.ix_(s, t)] = 1.0 / len(t)
        i, j = i + len(s), j + len(t)
    tail = np.arange(max(0, min(max_len - i, max_len - j)))
    if src_spans:
        m[j + tail, j + tail] = 1
    else:
        m[i + tail, j + tail] = 1
    return m


def get_replacement_mapper(prompts: Sequence[str], tokenizer, max_len: int = MAX_NUM_WORDS) -> torch.Tensor:
    """(len(prompts)-1, 77, 77) token-remap matrices between prompts[0] and each other prompt (seq_aligner.py:25-66).  Equal
    prompts give the identity without consulting the tokenizer; a word-count mismatch raises ValueError like the reference."""
    out = []
    words0 = prompts[0].split(" ")
    for other in prompts[1:]:
        words = other.split(" ")
        if len(words0) != len(words):
            raise ValueError("attention replacement edit can only be applied on prompts with the same length"
                             f" but prompt A has {len(words0)} words and prompt B has {len(words)} words.")
        changed = [k for k, (a, b) in enumerate(zip(words0, words)) if a != b]
        if not changed:
            out.append(torch.eye(max_len, dtype=torch.float32))
            continue
        if tokenizer is None:
            raise ValueError("a tokenizer is required to align prompts that differ")
        _, own0 = _word_of_token(prompts[0], tokenizer)
        _, own1 = _word_of_token(other, tokenizer)
        src = [np.flatnonzero(own0 == k) + 1 for k in changed]
        tgt = [np.flatnonzero(own1 == k) + 1 for k in changed]
        out.append(torch.from_numpy(_alignment_matrix(src, tgt, max_len)).float())
    return torch.stack(out)


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=MAX_NUM_WORDS):
    """(num_steps+1, len(prompts)-1, 1, 1, 77) blend schedule (p2p_utils.py:23-33, :55-73): alpha[s, k, w] = 1 while step s lies in the
    window [int(lo (S+1)), int(hi (S+1))) that applies to token w of edited prompt k — the "default_" window for every token,
    overridden (in the dict's order) by the windows given for individual words."""
    spec = dict(cross_replace_steps) if isinstance(cross_replace_steps, dict) else {"default_": cross_replace_steps}
    spec.setdefault("default_", (0.0, 1.0))
    if isinstance(cross_replace_steps, dict):
        cross_replace_steps.setdefault("default_", (0.0, 1.0))      # the reference adds the key to the caller's dict too
    n_rows, n_edit = num_steps + 1, len(prompts) - 1
    steps = torch.arange(n_rows).view(n_rows, 1)

    def inside(bounds) -> torch.Tensor:                               # (n_rows, 1) 0/1 column of one window
        lo, hi = (0, bounds) if isinstance(bounds, float) else bounds
        return ((steps >= int(lo * n_rows)) & (steps < int(hi * n_rows))).to(torch.float32)

    alpha = inside(spec["default_"]).view(n_rows, 1, 1).repeat(1, n_edit, max_num_words)
    for word, bounds in spec.items():
        if word == "default_":
            continue
        col = inside(bounds)
        for k in range(n_edit):
            pos = get_word_inds(prompts[k + 1], word, tokenizer)
            if len(pos) > 0:
                alpha[:, k, torch.as_tensor(pos, dtype=torch.long)] = col
    return alpha.reshape(n_rows, n_edit, 1, 1, max_num_words)


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
        self._identity_mapper = all(torch.equal(m.float().cpu(), eye) for m in self.mapper)
        self.is_pure_replacement = bool(local_blend is None and self._identity_mapper and bool((self.cross_replace_alpha == 1).all()))
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

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        """The edit on the CONDITIONAL half ``attn`` (prompts * heads, q, k) — p2p_attention.py:124-138 with :114-118 / :146-147.
        Sample 0 is the base prompt; every other sample e becomes
            cross:  (base @ mapper[e]) * alpha + (1 - alpha) * own      (all steps; alpha per token, per step)
            self :  base                                                  (inside the self-replace window, maps of <= width*height queries)
        written into ``attn`` where it is a view (the protocol is in place) and returned."""
        if not (is_cross or self._self_window()):
            return attn
        v = attn.reshape(self.batch_size, attn.shape[0] // self.batch_size, *attn.shape[1:])      # (prompts, heads, q, k)
        base, edits = v[0], v[1:]
        if is_cross:
            a = self.cross_replace_alpha[self.cur_step].to(device=v.device, dtype=v.dtype)            # (edits, 1, 1, 77)
            mapped = base.unsqueeze(0).expand(edits.shape[0], *base.shape) if self._identity_mapper \
                else torch.matmul(base.unsqueeze(0), self.mapper.to(device=v.device, dtype=v.dtype).unsqueeze(1))
            edits.copy_(mapped * a + (1 - a) * edits)
        elif edits.shape[2] <= self.width * self.height:
            edits.copy_(base.unsqueeze(0).expand_as(edits))
        return v.reshape(attn.shape)

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        cond = attn[attn.shape[0] // 2:]                          # [unc..., cond...]: only the conditional half is edited
        out = self.forward(cond, is_cross, place_in_unet)
        if out.data_ptr() != cond.data_ptr():                     # a non-contiguous input was edited on a copy
            cond.copy_(out)
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

    def skip_layer(self) -> None:
        """One attention call whose conditional samples are all identical to the base sample (the caller has proven it): a pure
        replacement leaves them unchanged, so only the layer / step counters advance — exactly as ``__call__`` would."""
        if not self.is_pure_replacement:
            raise RuntimeError("skip_layer needs a pure-replacement controller")
        self._tick()

    def fused_qk_src(self, is_cross: bool, n_tokens: int, batch: int, place_in_unet: str = "",
                     device=None, total_batch: Optional[int] = None, images: int = 1) -> Optional[torch.Tensor]:
        if not self.is_pure_replacement:
            raise RuntimeError("fused_qk_src needs a pure-replacement controller (identity mapper, alpha == 1)")
        if batch not in (2 * self.batch_size, 2 * self.batch_size - 1):
            # 2 n - 1: the block without the base sample's unconditional row ([unc_1.., cond_0, cond_1..]: pipeline `drop_unc0`) — the same
            # rule "every conditional row borrows from the first conditional one" with one unconditional row fewer in front
            raise ValueError(f"controller built for {self.batch_size} prompts expects a batch of {2 * self.batch_size} "
                             f"([unc..., cond...]), got {batch}")
        src = self.qk_src_vector(batch, device or self.device or "cuda", total_batch, images) if self.replaces(is_cross, n_tokens) else None
        self._tick()
        return src
