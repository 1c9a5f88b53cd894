"""Generate golden fixtures from the parts of the reference that import here.

Run in the build container only (``/root/reference`` does not travel):

    python tests/golden/make_golden.py

The reference's ``bean`` package cannot be imported (pyro, anndata, ... are
absent - SURVEY.md F2), but these files have no such dependency and load by file
path: ``bean/preprocessing/get_alpha0.py``, ``bean/preprocessing/get_pi_alpha0.py``.
Only numeric inputs/outputs are written (``alpha0_cases.npz``); no reference
source is copied.
"""
import importlib.util
import io
import contextlib
import os

import numpy as np
import torch

REF = "/root/reference/bean"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ga = load(os.path.join(REF, "preprocessing/get_alpha0.py"), "ref_get_alpha0")
    gp = load(os.path.join(REF, "preprocessing/get_pi_alpha0.py"), "ref_get_pi_alpha0")
    rng = np.random.default_rng(20240501)
    out = {}
    cases = [(3, 5, 400, False), (2, 4, 60, True), (4, 3, 3, False), (3, 6, 800, True)]
    for i, (R, B, G, masked) in enumerate(cases):
        depth = np.exp(rng.normal(5.0, 1.0, G))
        p = rng.dirichlet(np.ones(B) * 3, size=(R, G))
        a0_true = np.exp(-1.5 + 0.8 * np.log(depth))
        alpha = p * a0_true[None, :, None]
        probs = rng.standard_gamma(alpha)
        probs /= probs.sum(-1, keepdims=True)
        n = rng.poisson(depth)[None, :].repeat(R, 0)
        X = np.stack([[rng.multinomial(n[r, g], probs[r, g]) for g in range(G)] for r in range(R)])
        X = np.moveaxis(X, -1, 1).astype(np.float32)  # (R, B, G)
        sf = X.mean(-1).astype(np.float64)
        sf = sf / sf.mean()
        mask = np.ones((R, B), dtype=np.int64)
        if masked:
            mask[R - 1, 0] = 0
        with contextlib.redirect_stdout(io.StringIO()):
            a0, popt = ga.get_fitted_alpha0(torch.tensor(X), torch.tensor(sf), torch.tensor(mask))
            a0s, popts = ga.get_fitted_alpha0(torch.tensor(X), torch.tensor(sf), torch.tensor(mask), shrink=True)
            pred = ga.get_pred_alpha0(torch.tensor(X * 0.8), torch.tensor(sf), popt, torch.tensor(mask))
        out[f"c{i}_X"], out[f"c{i}_sf"], out[f"c{i}_mask"] = X, sf, mask
        out[f"c{i}_a0"] = a0.numpy()
        out[f"c{i}_popt"] = np.asarray(popt, dtype=np.float64)
        out[f"c{i}_a0_shrunk"] = a0s.numpy()
        out[f"c{i}_pred"] = np.asarray(pred)
        # control allele counts (R, C, G, A)
        C, A = 1, 2 if i % 2 == 0 else 4
        pi = rng.dirichlet(np.ones(A) * 2, size=G)
        tot = rng.poisson(depth * 0.3) + 1
        ac = np.stack([[[rng.multinomial(tot[g], rng.dirichlet(pi[g] * 30)) for g in range(G)] for _ in range(C)]
                       for _ in range(R)]).astype(np.float32)
        sfc = np.abs(rng.normal(1.0, 0.1, (R, C)))
        with contextlib.redirect_stdout(io.StringIO()):
            pa0, ppopt = gp.get_fitted_alpha0(torch.tensor(ac), torch.tensor(sfc))
            pa0s, _ = gp.get_fitted_alpha0(torch.tensor(ac), torch.tensor(sfc), shrink=True)
            praw, _ = gp.get_fitted_alpha0(torch.tensor(ac), torch.tensor(sfc), fit=False)
            ppred = gp.get_pred_alpha0(torch.tensor(ac), torch.tensor(sfc), ppopt)
        out[f"c{i}_ac"], out[f"c{i}_sfc"] = ac, sfc
        out[f"c{i}_pi_a0"] = pa0.numpy()
        out[f"c{i}_pi_popt"] = np.asarray(ppopt, dtype=np.float64)
        out[f"c{i}_pi_a0_shrunk"] = pa0s.numpy()
        out[f"c{i}_pi_a0_raw"] = praw.numpy()
        out[f"c{i}_pi_pred"] = np.asarray(ppred)
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "alpha0_cases.npz"), **out)
    print("wrote alpha0_cases.npz", {k: v.shape for k, v in out.items() if k.startswith("c0")})


if __name__ == "__main__":
    main()
