"""Fixture from the reference's OWN shipped run (`docs/example_run_output/variant/`): the parameters its 2000-step
`bean run sorting variant ... --fit-negctrl --scale-by-acc` fit ended with (`MixtureNormal+Acc.result.pkl`) and the
element / sgRNA tables it wrote from them (`bean_element_result.MixtureNormal+Acc.csv`, `bean_sgRNA_result...csv`).
`tests/test_example_golden.py` feeds the parameters to this build's `write_result_table` and expects the tables.

Run in the build container only (it reads /root/reference; nothing of the reference is executed):
    python tests/golden/make_example_golden.py
The pickle is read with a restricted unpickler that admits torch tensors / storages, torch's constraint
singletons and OrderedDict - no class of the reference or of Pyro.  The pickle predates the current
`save_dict` layout: `params` is the parameter store's `get_state()`, i.e. UNCONSTRAINED values (log of the
positive ones); they are stored here as they are, the test applies `exp` as `pyro.param` would.

Writes tests/golden/example_variant.npz: parameters, negative-control parameters, and every column of the two
CSVs (numeric columns as float64, the others as strings).
"""
import os
import pickle

import numpy as np
import pandas as pd

REF = "/root/reference/docs/example_run_output/variant"
HERE = os.path.dirname(os.path.abspath(__file__))


class TensorsOnly(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("torch", "collections"):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refused {module}.{name}")


def main():
    with open(os.path.join(REF, "MixtureNormal+Acc.result.pkl"), "rb") as fh:
        d = TensorsOnly(fh).load()
    out = {}
    for k, v in d["params"]["params"].items():
        out[f"P__{k}"] = v.detach().cpu().numpy()
    for k, v in d["negctrl"]["params"]["params"].items():
        out[f"N__{k}"] = v.detach().cpu().numpy()
    out["loss_first_last"] = np.array([d["loss"][0], d["loss"][-1]])
    for kind in ("element", "sgRNA"):
        df = pd.read_csv(os.path.join(REF, f"bean_{kind}_result.MixtureNormal+Acc.csv"), index_col=0)
        out[f"{kind}__columns"] = np.array(df.columns.tolist())
        out[f"{kind}__index"] = np.array(df.index.astype(str).tolist())
        for c in df.columns:
            col = df[c]
            key = f"{kind}__{c}"
            out[key] = col.values.astype(np.float64) if col.dtype.kind in "fiu" else np.array(col.astype(str).tolist())
    np.savez_compressed(os.path.join(HERE, "example_variant.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items() if not k.endswith("columns")})


if __name__ == "__main__":
    main()
