"""Read an AnnData ``.h5ad`` (the ReporterScreen container) into plain numpy /
pandas objects.

The reference loads screens through ``anndata`` (``bean/framework/ReporterScreen.py:
1009-1011``); neither ``anndata`` nor ``h5py`` is installed for the interpreter
that runs PyTorch here.  This module decodes the AnnData on-disk encodings it
needs (array, string-array, categorical, dataframe, dict, scalars, csr/csc
matrices) with ``h5py`` when that imports, and otherwise runs itself as a script
under a helper interpreter that has ``h5py`` (``$BEAN_H5PY_PYTHON``, default
``/opt/conda/bin/python3.9``) and ships the decoded tree back as a pickle of
numpy arrays and builtins.
"""
from __future__ import annotations

import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

HELPER_PYTHON = os.environ.get("BEAN_H5PY_PYTHON", "/opt/conda/bin/python3.9")


# ----------------------------------------------------------- h5py-side decoding
def _attr(obj, key, default=None):
    v = obj.attrs.get(key, default)
    return v.decode() if isinstance(v, bytes) else v


def _strings(ds):
    arr = ds[()]
    if isinstance(arr, bytes):
        return arr.decode()
    if isinstance(arr, str):
        return arr
    return np.array([x.decode() if isinstance(x, bytes) else x for x in arr.reshape(-1)], dtype=object).reshape(arr.shape)


def _decode(obj):
    """h5py object -> builtin / numpy tree.  DataFrames become
    {"__dataframe__": True, "index_name", "index", "columns": [...], "data": {col: array | categorical}}."""
    import h5py

    enc = _attr(obj, "encoding-type", "")
    if isinstance(obj, h5py.Dataset):
        if enc in ("string-array", "string") or obj.dtype.kind in "OS":
            return _strings(obj)
        val = obj[()]
        return val.item() if np.ndim(val) == 0 else val
    if enc == "dataframe":
        idx_key = _attr(obj, "_index")
        order = [c.decode() if isinstance(c, bytes) else c for c in obj.attrs["column-order"]]
        data = {}
        for c in order:
            node = obj
            # a "/" inside a column name nests groups on disk
            for part in c.split("/"):
                node = node[part]
            data[c] = _decode(node)
        return {"__dataframe__": True, "index_name": idx_key, "index": _decode(obj[idx_key]),
                "columns": order, "data": data}
    if enc == "categorical":
        return {"__categorical__": True, "categories": _decode(obj["categories"]),
                "codes": obj["codes"][()], "ordered": bool(obj.attrs.get("ordered", False))}
    if enc in ("csr_matrix", "csc_matrix"):
        return {"__sparse__": enc, "data": obj["data"][()], "indices": obj["indices"][()],
                "indptr": obj["indptr"][()], "shape": tuple(int(s) for s in obj.attrs["shape"])}
    if enc.startswith("nullable"):
        return {"__nullable__": True, "values": obj["values"][()], "mask": obj["mask"][()]}
    return {k: _decode(v) for k, v in obj.items()}


def decode_file(path: str) -> dict:
    import h5py

    with h5py.File(path, "r") as f:
        out = {"X": _decode(f["X"]) if "X" in f else None}
        for key in ("layers", "obs", "var", "uns"):
            out[key] = _decode(f[key]) if key in f else {}
    return out


# ------------------------------------------------------------- main-side helpers
def _densify(x):
    if isinstance(x, dict) and "__sparse__" in x:
        from scipy import sparse

        cls = sparse.csr_matrix if x["__sparse__"] == "csr_matrix" else sparse.csc_matrix
        return np.asarray(cls((x["data"], x["indices"], x["indptr"]), shape=x["shape"]).todense())
    return x


def to_pandas(tree):
    """Rebuild DataFrames / categoricals from the decoded tree."""
    import pandas as pd

    if isinstance(tree, dict) and tree.get("__dataframe__"):
        cols = {}
        for c in tree["columns"]:
            v = tree["data"][c]
            if isinstance(v, dict) and v.get("__categorical__"):
                cols[c] = pd.Categorical.from_codes(v["codes"], categories=list(v["categories"]), ordered=v["ordered"])
            elif isinstance(v, dict) and v.get("__nullable__"):
                vals = v["values"].astype(object)
                vals[v["mask"].astype(bool)] = None
                cols[c] = vals
            else:
                cols[c] = v
        name = tree["index_name"]
        idx = pd.Index(tree["index"], name=None if name == "_index" else name)
        return pd.DataFrame(cols, index=idx, columns=tree["columns"])
    if isinstance(tree, dict) and tree.get("__categorical__"):
        import pandas as pd  # noqa: F811

        return pd.Categorical.from_codes(tree["codes"], categories=list(tree["categories"]), ordered=tree["ordered"])
    if isinstance(tree, dict) and "__sparse__" in tree:
        return _densify(tree)
    if isinstance(tree, dict):
        return {k: to_pandas(v) for k, v in tree.items()}
    return tree


def read_tree(path: str) -> dict:
    """Decode ``path`` in-process if h5py imports, else through the helper interpreter."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    try:
        import h5py  # noqa: F401

        return decode_file(path)
    except ImportError:
        pass
    if not os.path.exists(HELPER_PYTHON):
        raise ImportError(
            "reading .h5ad needs h5py: install it, or point BEAN_H5PY_PYTHON at an interpreter that has it "
            f"(tried {HELPER_PYTHON})"
        )
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "tree.pkl")
        res = subprocess.run([HELPER_PYTHON, os.path.abspath(__file__), path, out], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"h5ad helper failed:\n{res.stderr}")
        with open(out, "rb") as fh:
            return pickle.load(fh)


if __name__ == "__main__":  # helper-interpreter entry: h5ad_io.py <in.h5ad> <out.pkl>
    with open(sys.argv[2], "wb") as fh:
        pickle.dump(decode_file(sys.argv[1]), fh, protocol=4)
