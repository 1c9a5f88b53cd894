"""Freeze small ELBO cases (inputs, injected draws, oracle loss and gradients).

    python tests/golden/make_elbo_golden.py

The expected values come from the repository's own CPU oracle (``oracle/``), not
from the reference (which cannot run here - "parity unpinned", see
oracle/__init__.py).  The fixtures guard the oracle and the HIP path against
drifting *together*: the GPU tests compare the kernels both with the live oracle
and with these frozen numbers.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bean_amd  # noqa: E402,F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen  # noqa: E402
from oracle import elbo, svi  # noqa: E402

CASES = [
    ("mix", "MixtureNormal", dict(n_guides=150, n_reps=3, seed=21, mask_fraction=0.05), {}),
    ("mixacc", "MixtureNormal", dict(n_guides=130, n_reps=2, seed=22, with_accessibility=True),
     dict(scale_by_accessibility=True)),
    ("normal", "Normal", dict(n_guides=100, n_reps=2, seed=23), {}),
    ("control", "ControlNormal", dict(n_guides=70, n_reps=3, seed=24), {}),
]


def main():
    out = {}
    for tag, fam, gen_kw, loss_kw in CASES:
        torch.manual_seed(sum(map(ord, tag)))
        data = make_sorting_variant_screen(**gen_kw)
        params = elbo.init_params(fam, data, scale_by_acc=loss_kw.get("scale_by_accessibility", False))
        params = {k: (v.detach() + 0.2 * torch.randn_like(v)).requires_grad_(True) for k, v in params.items()}
        R, G, T = data.n_reps, data.n_guides, data.n_targets
        shape = () if fam == "ControlNormal" else (T, 1)
        noise = {"eps_mu": torch.randn(shape, dtype=torch.float64), "eps_sd": torch.randn(shape, dtype=torch.float64)}
        if fam == "MixtureNormal":
            conc = torch.tensor(np.random.default_rng(1).uniform(0.5, 20, (G, 2)))
            noise["pi"] = torch.distributions.Dirichlet(conc).sample((R,)).unsqueeze(1)
            if loss_kw:
                noise["eps_noise"] = torch.randn(G, dtype=torch.float64)
        loss, grads, rec = svi.loss_and_grads(elbo.LOSSES[fam], data, params, noise=noise, **loss_kw)
        out[f"{tag}__gen"] = np.asarray(repr(gen_kw))
        for k, v in params.items():
            out[f"{tag}__param__{k}"] = v.detach().numpy()
        for k, v in noise.items():
            out[f"{tag}__noise__{k}"] = v.numpy()
        for k, v in grads.items():
            out[f"{tag}__grad__{k}"] = v.numpy()
        out[f"{tag}__loss"] = np.asarray(loss)
        print(tag, loss, rec["model"])
    np.savez_compressed(os.path.join(HERE, "elbo_cases.npz"), **out)


if __name__ == "__main__":
    main()
