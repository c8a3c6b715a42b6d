"""ORACLE (test infrastructure, never imported by omg_amd): an optional STORAGE-PRECISION EMULATION for the fp32 restatements.

The reference runs its UNets in ``torch.float16`` (inference_lora.py:153-159): every torch op reads fp16 tensors, accumulates in
fp32 inside the op and ROUNDS ITS OUTPUT to fp16.  ``with rounding(torch.float16):`` makes the oracle do the same at op granularity —
``r(x)`` (= ``x.to(dtype).float()``) is applied to the output of every Linear / conv / norm / activation / residual add / attention
score, probability and output — so that three distances can be quoted side by side (BASELINE's |d| < 1e-3 is asked "vs reference",
and the reference's own arithmetic is fp16):

    d(HIP, fp32 oracle)     what the tests bound          d(fp16 oracle, fp32 oracle)    what the reference's arithmetic itself costs
    d(HIP, fp16 oracle)     how far the HIP path is from the reference AS EXECUTED (both carry fp16 rounding; the roundings differ in
                            place — the flash kernel never rounds scores, the merged LoRA rounds once — so this is not zero either)

Outside the context manager ``r`` is the identity and the oracles are exactly the fp32 functions they were (module default).
The weights are the caller's business: ``init_state_dict(dtype=torch.float16)`` already returns fp16-representable values."""
from __future__ import annotations

import contextlib

import torch

_DT = None


def r(x: torch.Tensor) -> torch.Tensor:
    """round to the emulated storage dtype (identity outside ``rounding``)"""
    return x if _DT is None else x.to(_DT).float()


def active():
    return _DT


@contextlib.contextmanager
def rounding(dtype):
    global _DT
    prev, _DT = _DT, dtype
    try:
        yield
    finally:
        _DT = prev
