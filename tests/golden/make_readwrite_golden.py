"""Golden result tables from the reference's own ``bean/model/readwrite.py``
(imported by file path; it only needs numpy/pandas/scipy).

    python tests/golden/make_readwrite_golden.py

Writes ``readwrite_cases.npz`` (inputs) and ``readwrite_<case>_{element,sgRNA}.csv``
(the reference's outputs).  Data only - no reference source is copied.
"""
import contextlib
import importlib.util
import io
import os

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_readwrite", "/root/reference/bean/model/readwrite.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CASES = {
    # name: (T, G, ndim, sd_fitted, negctrl, adjust, n_neg, acc, noise, survival)
    "plain": (40, 120, 2, True, False, True, 15, False, False, False),
    "negctrl": (40, 120, 2, True, True, True, 15, True, True, False),
    "fewneg": (25, 60, 2, True, False, True, 4, True, False, False),
    "noadjust": (25, 60, 2, True, True, False, 0, False, False, False),
    "survival": (30, 90, 2, False, True, True, 12, False, False, True),
    "tiling": (50, 70, 1, True, False, True, 11, False, False, False),
}


def build(name, seed):
    T, G, ndim, sd_fit, negctrl, adjust, n_neg, acc, noise, surv = CASES[name]
    g = torch.Generator().manual_seed(seed)
    shp = (T, 1) if ndim == 2 else (T,)
    P = {"mu_loc": torch.randn(shp, generator=g), "mu_scale": torch.rand(shp, generator=g) * 0.5 + 0.05}
    if sd_fit:
        P["sd_loc"] = torch.randn(shp, generator=g) * 0.2
    A = 2 if ndim == 2 else 5
    P["alpha_pi"] = torch.rand((G, A), generator=g) + 0.1
    if noise:
        P["noise_scale"] = torch.rand(G, generator=g) + 0.2
    neg = None
    if negctrl:
        neg = {"mu_loc": torch.tensor(0.13), "mu_scale": torch.tensor(0.4)}
        if sd_fit:
            neg["sd_loc"] = torch.tensor(-0.2)
    target = pd.DataFrame({"edit_rate_mean": np.linspace(0.1, 0.9, T), "n_guides": np.arange(T) % 5 + 1},
                          index=pd.Index([f"var{i}" for i in range(T)], name="target"))
    guide = pd.DataFrame({"edit_rate": np.linspace(0.0, 1.0, G), "target": [f"var{i % T}" for i in range(G)]},
                         index=pd.Index([f"g{i}" for i in range(G)], name="name"))
    negs = np.arange(0, 2 * n_neg, 2)[:n_neg] if adjust else None
    gacc = np.linspace(0.5, 30.0, G) if acc else None
    kw = dict(negctrl_params=neg, adjust_confidence_by_negative_control=adjust,
              adjust_confidence_negatives=negs, guide_acc=gacc, sd_is_fitted=sd_fit, is_survival_screen=surv)
    return target, guide, P, neg, kw


def main():
    store = {}
    for i, name in enumerate(CASES):
        target, guide, P, neg, kw = build(name, 100 + i)
        for k, v in P.items():
            store[f"{name}__P__{k}"] = v.numpy()
        if neg is not None:
            for k, v in neg.items():
                store[f"{name}__N__{k}"] = v.numpy()
        prefix = os.path.join(HERE, f"readwrite_{name}_")
        with contextlib.redirect_stdout(io.StringIO()):
            ref.write_result_table(target.copy(), guide.copy(), P, "M", prefix=prefix, **kw)
        os.replace(prefix + "bean_element_result.M.csv", prefix + "element.csv")
        os.replace(prefix + "bean_sgRNA_result.M.csv", prefix + "sgRNA.csv")
    np.savez_compressed(os.path.join(HERE, "readwrite_cases.npz"), **store)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.startswith("readwrite_")))


if __name__ == "__main__":
    main()
