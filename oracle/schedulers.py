"""ORACLE (test infrastructure): DDIM and Euler-discrete schedulers as the reference's loop uses
them (``self.scheduler.set_timesteps`` / ``scale_model_input`` / ``step`` at
src/pipelines/lora_pipeline.py:390-392,492,615).

The schedulers are third-party code (``diffusers==0.25.0`` ``schedulers.scheduling_ddim`` /
``scheduling_euler_discrete``, not vendored, not installable here); this restates their
published algorithm for the SDXL-base scheduler config (beta_start 0.00085,
beta_end 0.012, scaled_linear, 1000 train steps, timestep_spacing "leading", steps_offset 1,
epsilon prediction, eta = 0, no clipping).  PARITY UNPINNED by the reference (no vectors exist);
self-consistency (DDIM telescoping identity, Euler sigma monotonicity) is checked in tests.
float64 numpy throughout.
"""
from __future__ import annotations

import numpy as np


def alphas_cumprod(n_train: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> np.ndarray:
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def leading_timesteps(n_steps: int, n_train: int = 1000, offset: int = 1) -> np.ndarray:
    ratio = n_train // n_steps
    return (np.arange(0, n_steps) * ratio).round()[::-1].astype(np.int64) + offset


class DDIM:
    init_noise_sigma = 1.0

    def __init__(self, n_steps: int, n_train: int = 1000):
        self.ac = alphas_cumprod(n_train)
        self.final_alpha = self.ac[0]          # set_alpha_to_one = False
        self.timesteps = leading_timesteps(n_steps, n_train)
        self.ratio = n_train // n_steps

    def scale_model_input(self, x, i):
        return x

    def step(self, eps, i, x):
        t = int(self.timesteps[i])
        prev = t - self.ratio
        a_t = self.ac[t]
        a_p = self.ac[prev] if prev >= 0 else self.final_alpha
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


class EulerDiscrete:
    def __init__(self, n_steps: int, n_train: int = 1000):
        ac = alphas_cumprod(n_train)
        sig = ((1 - ac) / ac) ** 0.5
        self.timesteps = leading_timesteps(n_steps, n_train).astype(np.float64)
        s = np.interp(self.timesteps, np.arange(n_train), sig)
        self.sigmas = np.concatenate([s, [0.0]])
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)   # "leading" spacing

    def scale_model_input(self, x, i):
        return x / (self.sigmas[i] ** 2 + 1) ** 0.5

    def step(self, eps, i, x):
        return x + eps * (self.sigmas[i + 1] - self.sigmas[i])


def make(name: str, n_steps: int):
    return {"ddim": DDIM, "euler": EulerDiscrete}[name](n_steps)
