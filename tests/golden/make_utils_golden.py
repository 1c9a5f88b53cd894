"""Golden vectors from the reference's own pure-torch helpers of the ELBO.

Run in the build container only (``/root/reference`` does not travel):

    python tests/golden/make_utils_golden.py

``bean/model/utils.py`` cannot be imported as a module (its line 3 is ``import pyro``, absent here -
SURVEY.md F2), but three of its functions call nothing from Pyro: ``get_alpha`` (10-31),
``get_std_normal_prob`` (34-76) and ``_scale_edited_pi`` (79-103).  This script parses the file,
takes exactly those three ``FunctionDef`` nodes UNCHANGED, compiles them into a namespace that
holds only ``torch`` and ``torch.distributions as tdist`` (the two names they use) and evaluates
them on seeded inputs shaped and typed as the models call them (``bean/model/model.py:480-547,
675-700``).  No stand-in for anything is written: a function that reached for Pyro would raise
``NameError``.  Only numeric inputs / outputs are stored (``utils_cases.npz``); no reference source
is copied.  ``tests/test_utils_golden.py`` pins ``oracle.elbo.std_normal_bin_prob``,
``dirmult_concentration`` and the scaling half of ``scale_pi_by_accessibility`` to them, and the
GPU suite feeds the same inputs to the device functions.
"""
import ast
import os

import numpy as np
import torch
import torch.distributions as tdist

REF = "/root/reference/bean/model/utils.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = ("get_alpha", "get_std_normal_prob", "_scale_edited_pi")


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in nodes) == sorted(WANTED)
    ns = {"torch": torch, "tdist": tdist}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF, "exec"), ns)  # noqa: S102
    return [ns[n] for n in WANTED]


def main():
    get_alpha, get_std_normal_prob, scale_edited_pi = reference_functions()
    rng = np.random.default_rng(20261004)
    out = {}

    # ---- get_std_normal_prob as the variant models call it: quantiles f64 (B, G, A), mu / sd f32
    cases = [
        # (upper quantiles, lower quantiles) incl. the open edges uq == 1.0 / lq == 0.0 and a bulk bin (0, 1)
        ([0.2, 0.4, 0.8, 1.0, 1.0], [0.0, 0.2, 0.6, 0.8, 0.0]),
        ([0.3, 1.0], [0.0, 0.7]),
        ([0.25, 0.5, 0.75, 0.999], [0.001, 0.25, 0.5, 0.75]),
    ]
    for i, (uq, lq) in enumerate(cases):
        B, G, A = len(uq), 37 + 11 * i, 2 + 3 * (i == 2)
        uq_t = torch.tensor(uq, dtype=torch.float64)
        lq_t = torch.tensor(lq, dtype=torch.float64)
        mu = torch.tensor(rng.normal(0, 1.5, (G, A)), dtype=torch.float32)
        sd = torch.tensor(np.exp(rng.normal(0, 0.5, (G, A))), dtype=torch.float32)
        mu[:, 0], sd[:, 0] = 0.0, 1.0
        mask = None
        if i == 2:  # tiling: allele mask, column 0 always valid (data_class.py:851-872)
            mask = torch.tensor(rng.random((G, A)) < 0.6)
            mask[:, 0] = True
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            res = get_std_normal_prob(
                uq_t.unsqueeze(-1).unsqueeze(-1).expand((-1, G, A)),
                lq_t.unsqueeze(-1).unsqueeze(-1).expand((-1, G, A)),
                mu.to(dt).unsqueeze(0).expand((B, -1, -1)),
                sd.to(dt).unsqueeze(0).expand((B, -1, -1)),
                **({} if mask is None else {"mask": mask.unsqueeze(0).expand((B, -1, -1))}),
            )
            out[f"snp{i}_{tag}_out"] = res.numpy()
            assert res.dtype == torch.float64
        out[f"snp{i}_uq"], out[f"snp{i}_lq"] = uq_t.numpy(), lq_t.numpy()
        out[f"snp{i}_mu"], out[f"snp{i}_sd"] = mu.numpy(), sd.numpy()
        if mask is not None:
            out[f"snp{i}_mask"] = mask.numpy()

    # ---- get_alpha: expected_guide_p (R, B, G) f32 or f64, size factor f64 (R, B), sample mask (R, B), a0 f64 (G,)
    for i, (R, B, G, masked, dt) in enumerate([(3, 5, 41, False, torch.float64), (2, 4, 17, True, torch.float64),
                                               (4, 6, 29, True, torch.float32), (1, 2, 5, False, torch.float32)]):
        p = torch.tensor(rng.dirichlet(np.ones(B), size=(R, G)), dtype=dt).permute(0, 2, 1).contiguous()
        if i == 1:
            p[0, :, 3] = 0.0  # a guide with no expected mass at all: only the epsilon terms remain
        sf = torch.tensor(np.exp(rng.normal(0, 0.3, (R, B))), dtype=torch.float64)
        sm = torch.ones((R, B), dtype=torch.int64)
        if masked:
            sm[R - 1, 0] = 0
            sm[0, B - 1] = 0
        a0 = torch.tensor(np.exp(rng.normal(3, 1, G)), dtype=torch.float64)
        res = get_alpha(p, sf, sm, a0)
        out[f"ga{i}_p"], out[f"ga{i}_sf"], out[f"ga{i}_mask"], out[f"ga{i}_a0"] = p.numpy(), sf.numpy(), sm.numpy(), a0.numpy()
        out[f"ga{i}_out"] = res.numpy()
        # float mask, as the ScreenData of the build holds it
        res2 = get_alpha(p, sf, sm.to(torch.float64), a0)
        assert torch.equal(res, res2)

    # ---- _scale_edited_pi: pi[..., 1:] (R, 1, G, A-1), accessibility (G,) f64
    for i, (R, G, A1, dt) in enumerate([(3, 23, 1, torch.float32), (2, 11, 4, torch.float32), (2, 9, 1, torch.float64)]):
        pi = torch.tensor(rng.dirichlet(np.ones(A1 + 1), size=(R, 1, G)), dtype=dt)[..., 1:].contiguous()
        acc = torch.tensor(np.exp(rng.normal(0, 1, G)) + 1.0, dtype=torch.float64)
        res = scale_edited_pi(pi, acc)
        out[f"sep{i}_pi"], out[f"sep{i}_acc"], out[f"sep{i}_out"] = pi.numpy(), acc.numpy(), res.numpy()

    path = os.path.join(HERE, "utils_cases.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays")


if __name__ == "__main__":
    main()
