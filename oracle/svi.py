"""SVI loop restatement: ``Trace_ELBO`` step + ``ClippedAdam`` (oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Follows
``bean/model/run.py:347-396`` (``run_inference``): ``ClippedAdam({"lr": 0.01,
"lrd": gamma ** (1 / num_steps)})`` on the unconstrained parameters, loss
returned as a Python float each step.

``pyro.optim.ClippedAdam`` (pyro-ppl 1.8/1.9, ``pyro/optim/clipped_adam.py``) per
step and parameter: ``lr *= lrd``; ``grad.clamp_(-clip, clip)`` with
``clip_norm = 10``; Adam moments with betas (0.9, 0.999); bias-corrected step
``lr * sqrt(1 - b2^t) / (1 - b1^t)``; ``p -= step * m / (sqrt(v) + 1e-8)``.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch


class ClippedAdam:
    def __init__(self, params: Dict[str, torch.Tensor], lr=0.01, lrd=1.0,
                 betas=(0.9, 0.999), eps=1e-8, clip_norm=10.0):
        self.params = params
        self.lr, self.lrd, self.betas, self.eps, self.clip = lr, lrd, betas, eps, clip_norm
        self.state = {
            k: {"step": 0, "m": torch.zeros_like(v), "v": torch.zeros_like(v), "lr": lr}
            for k, v in params.items()
        }

    @torch.no_grad()
    def step(self):
        b1, b2 = self.betas
        for k, p in self.params.items():
            if p.grad is None:
                continue
            st = self.state[k]
            st["lr"] *= self.lrd  # Pyro keeps one optimiser (one lr) per parameter
            g = p.grad.clamp_(-self.clip, self.clip)
            st["step"] += 1
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = st["v"].sqrt().add_(self.eps)
            bc1 = 1 - b1 ** st["step"]
            bc2 = 1 - b2 ** st["step"]
            p.addcdiv_(st["m"], denom, value=-st["lr"] * math.sqrt(bc2) / bc1)
            p.grad = None


def svi_step(loss_fn: Callable, data, params, optim: ClippedAdam, noise=None, **kw) -> float:
    loss = loss_fn(data, params, noise=noise, **kw)
    loss.backward()
    optim.step()
    return float(loss.detach())


def loss_and_grads(loss_fn: Callable, data, params, noise=None, **kw):
    """Loss and d loss / d (unconstrained params) for one noise draw."""
    for p in params.values():
        p.grad = None
    rec = {}
    loss = loss_fn(data, params, noise=noise, record=rec, **kw)
    loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    return float(loss.detach()), grads, rec


def run_svi(loss_fn: Callable, data, params, num_steps=2000, initial_lr=0.01, gamma=0.1,
            noise_fn: Optional[Callable[[int], dict]] = None, detect_anomaly=False, **kw):
    """``run_inference`` (``bean/model/run.py:347-396``) on the oracle.

    ``noise_fn(step)`` may supply the draws of each step (exact-noise trajectory
    tests); otherwise torch's global RNG is used.  ``detect_anomaly`` mirrors
    ``torch.autograd.set_detect_anomaly(True)`` at ``bean/model/model.py:399``.
    """
    lrd = gamma ** (1 / num_steps)
    optim = ClippedAdam(params, lr=initial_lr, lrd=lrd)
    losses: List[float] = []
    prev = torch.is_anomaly_enabled()
    torch.autograd.set_detect_anomaly(detect_anomaly)
    try:
        for t in range(num_steps):
            noise = noise_fn(t) if noise_fn is not None else None
            losses.append(svi_step(loss_fn, data, params, optim, noise=noise, **kw))
    finally:
        torch.autograd.set_detect_anomaly(prev)
    return params, losses
