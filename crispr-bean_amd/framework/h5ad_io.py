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


# ------------------------------------------------------------------- writing
def _enc_frame(df) -> dict:
    """pandas DataFrame -> encode tree (the mirror image of ``_decode``'s dataframe node)."""
    import pandas as pd

    cols = {}
    for c in df.columns:
        v = df[c]
        if isinstance(v.dtype, pd.CategoricalDtype):
            cols[str(c)] = {"__categorical__": True, "categories": np.asarray(v.cat.categories.astype(str), dtype=object),
                            "codes": np.asarray(v.cat.codes), "ordered": bool(v.cat.ordered)}
        elif v.dtype.kind in "biuf":
            cols[str(c)] = v.to_numpy()
        else:
            cols[str(c)] = np.asarray(["" if x is None or (isinstance(x, float) and np.isnan(x)) else str(x)
                                       for x in v.tolist()], dtype=object)
    return {"__dataframe__": True, "index_name": "_index" if df.index.name is None else str(df.index.name),
            "index": np.asarray(df.index.astype(str), dtype=object), "columns": [str(c) for c in df.columns], "data": cols}


def _enc_value(v):
    import pandas as pd

    if isinstance(v, pd.DataFrame):
        return _enc_frame(v)
    if isinstance(v, dict):
        return {str(k): _enc_value(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        arr = np.asarray(v)
        return np.asarray([str(x) for x in v], dtype=object) if arr.dtype.kind in "OUS" else arr
    if isinstance(v, np.ndarray) and v.dtype.kind in "OUS":
        return np.asarray([str(x) for x in v.reshape(-1)], dtype=object).reshape(v.shape)
    return v


def encode_screen(screen) -> dict:
    """ReporterScreen -> encode tree of numpy arrays / builtins (picklable)."""
    return {"X": np.asarray(screen.X, dtype=np.float32),
            "layers": {k: np.asarray(v, dtype=np.float32) for k, v in screen.layers.items()},
            "obs": _enc_frame(screen.guides), "var": _enc_frame(screen.samples),
            "uns": {str(k): _enc_value(v) for k, v in screen.uns.items()}}


def _write_node(parent, name, v):
    import h5py

    str_dt = h5py.string_dtype(encoding="utf-8")

    def tag(obj, enc, ver):
        obj.attrs["encoding-type"] = enc
        obj.attrs["encoding-version"] = ver

    if isinstance(v, dict) and v.get("__dataframe__"):
        g = parent.create_group(name)
        tag(g, "dataframe", "0.2.0")
        g.attrs["_index"] = v["index_name"]
        g.attrs["column-order"] = np.asarray(v["columns"], dtype=str_dt) if v["columns"] else np.empty(0, dtype=str_dt)
        _write_node(g, v["index_name"], v["index"])
        for c in v["columns"]:
            _write_node(g, c, v["data"][c])  # a "/" in the name nests groups, as anndata writes it
        return
    if isinstance(v, dict) and v.get("__categorical__"):
        g = parent.create_group(name)
        tag(g, "categorical", "0.2.0")
        g.attrs["ordered"] = bool(v["ordered"])
        _write_node(g, "categories", v["categories"])
        _write_node(g, "codes", np.asarray(v["codes"]))
        return
    if isinstance(v, dict):
        g = parent.create_group(name)
        tag(g, "dict", "0.1.0")
        for k, x in v.items():
            _write_node(g, k, x)
        return
    if isinstance(v, str):
        d = parent.create_dataset(name, data=v, dtype=str_dt)
        tag(d, "string", "0.2.0")
        return
    if isinstance(v, (bool, int, float, np.bool_, np.integer, np.floating)):
        d = parent.create_dataset(name, data=v)
        tag(d, "numeric-scalar", "0.2.0")
        return
    arr = np.asarray(v)
    if arr.dtype.kind in "OUS":
        d = parent.create_dataset(name, data=arr.astype(object), dtype=str_dt)
        tag(d, "string-array", "0.2.0")
    else:
        d = parent.create_dataset(name, data=arr)
        tag(d, "array", "0.2.0")


def write_tree_file(tree: dict, path: str) -> None:
    """Encode tree -> AnnData ``.h5ad`` (encoding versions of anndata 0.8+); needs h5py."""
    import h5py

    with h5py.File(path, "w") as f:
        f.attrs["encoding-type"] = "anndata"
        f.attrs["encoding-version"] = "0.1.0"
        _write_node(f, "X", tree["X"])
        _write_node(f, "layers", tree["layers"])
        _write_node(f, "obs", tree["obs"])
        _write_node(f, "var", tree["var"])
        _write_node(f, "uns", tree["uns"])
        for k in ("obsm", "varm", "obsp", "varp"):
            _write_node(f, k, {})


def write_screen(screen, path: str) -> None:
    """Write a ReporterScreen as ``.h5ad`` (``ReporterScreen.write``, reference
    ``bean/framework/ReporterScreen.py:896-915``), in-process with h5py or through the helper."""
    tree = encode_screen(screen)
    try:
        import h5py  # noqa: F401

        write_tree_file(tree, path)
        return
    except ImportError:
        pass
    if not os.path.exists(HELPER_PYTHON):
        raise ImportError("writing .h5ad needs h5py: install it, or point BEAN_H5PY_PYTHON at an interpreter that "
                          f"has it (tried {HELPER_PYTHON})")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "tree.pkl")
        with open(src, "wb") as fh:
            pickle.dump(tree, fh, protocol=4)
        res = subprocess.run([HELPER_PYTHON, os.path.abspath(__file__), "--write", src, path],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"h5ad helper failed:\n{res.stderr}")


if __name__ == "__main__":  # helper-interpreter entry: h5ad_io.py <in.h5ad> <out.pkl> | --write <tree.pkl> <out.h5ad>
    if sys.argv[1] == "--write":
        with open(sys.argv[2], "rb") as fh:
            write_tree_file(pickle.load(fh), sys.argv[3])
    else:
        with open(sys.argv[2], "wb") as fh:
            pickle.dump(decode_file(sys.argv[1]), fh, protocol=4)
