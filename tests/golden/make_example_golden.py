"""Fixture from the reference's OWN shipped run (`docs/example_run_output/variant/`): the parameters its 2000-step
`bean run sorting variant ... --fit-negctrl --scale-by-acc` fit ended with (`MixtureNormal+Acc.result.pkl`) and the
element / sgRNA tables it wrote from them (`bean_element_result.MixtureNormal+Acc.csv`, `bean_sgRNA_result...csv`).
`tests/test_example_golden.py` feeds the parameters to this build's `write_result_table` and expects the tables.

Run in the build container only (it reads /root/reference; nothing of the reference is executed):
    python tests/golden/make_example_golden.py
The pickle is read with a restricted unpickler that admits an exact list of five globals (tensor rebuild, a
weights-only storage loader, OrderedDict, two torch constraint classes) - no class of the reference or of Pyro.  The pickle predates the current
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


def _storage_from_bytes(b):
    """What the pickle calls as torch.storage._load_from_bytes - which is torch.load(..., weights_only=False), an
    unrestricted nested unpickle; the nested stream holds a bare storage, so the weights-only loader reads it."""
    import io

    import torch

    return torch.load(io.BytesIO(b), weights_only=True)


class TensorsOnly(pickle.Unpickler):
    """Admits exactly the globals this pickle names (listed with pickletools, nothing executed): the tensor
    rebuild function, the storage loader (replaced by its weights-only form), OrderedDict and the two constraint
    classes of the parameter store's state.  Everything else - torch.load, torch.hub, cpp_extension ... - is refused."""

    ALLOWED = {
        ("torch._utils", "_rebuild_tensor_v2"),
        ("collections", "OrderedDict"),
        ("torch.distributions.constraints", "_Real"),
        ("torch.distributions.constraints", "_GreaterThan"),
    }

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _storage_from_bytes
        if (module, name) in self.ALLOWED:
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
