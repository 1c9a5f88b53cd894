"""A light ReporterScreen: the parts of ``bean/framework/ReporterScreen.py`` (an
``anndata.AnnData`` subclass through ``perturb_tools.Screen``) that ``bean run``
touches - ``X`` (guides x samples), ``layers``, ``guides`` (obs), ``samples``
(var), ``uns``, 2-D slicing, ``get_guide_edit_rate`` and the per-replicate log
fold changes of the sgRNA table.
"""
from __future__ import annotations

import copy as _copy
from typing import Optional

import numpy as np
import pandas as pd

from . import h5ad_io


class ReporterScreen:
    def __init__(self, X, guides: pd.DataFrame, samples: pd.DataFrame, layers=None, uns=None):
        self.X = np.asarray(X)
        self.guides = guides
        self.samples = samples
        self.layers = dict(layers or {})
        self.uns = dict(uns or {})
        assert self.X.shape == (len(guides), len(samples)), (self.X.shape, len(guides), len(samples))

    # AnnData spellings used by the reference
    obs = property(lambda self: self.guides)
    var = property(lambda self: self.samples)
    n_obs = property(lambda self: self.X.shape[0])
    n_vars = property(lambda self: self.X.shape[1])
    shape = property(lambda self: self.X.shape)

    @property
    def tiling(self) -> bool:
        return bool(self.uns.get("tiling", False))

    def copy(self) -> "ReporterScreen":
        return ReporterScreen(self.X.copy(), self.guides.copy(), self.samples.copy(),
                              {k: v.copy() for k, v in self.layers.items()}, _copy.deepcopy(self.uns))

    @staticmethod
    def _positions(key, index: pd.Index, n: int) -> np.ndarray:
        if isinstance(key, slice):
            return np.arange(n)[key]
        if isinstance(key, pd.Series):
            key = key.values
        key = np.asarray(key) if not isinstance(key, pd.Index) else key
        if getattr(key, "dtype", None) is not None and key.dtype == bool:
            return np.nonzero(np.asarray(key))[0]
        if getattr(key, "dtype", None) is not None and key.dtype.kind in "iu":
            return np.asarray(key, dtype=np.int64)
        return index.get_indexer(pd.Index(key))  # labels

    def __getitem__(self, key) -> "ReporterScreen":
        """``screen[guides, samples]`` with slices, masks, positions or labels.  Tables in
        ``uns`` that carry a ``guide`` column / sample columns are sliced along, as the
        reference does (``ReporterScreen.py:274-330``)."""
        gk, sk = key if isinstance(key, tuple) else (key, slice(None))
        gi = self._positions(gk, self.guides.index, self.n_obs)
        si = self._positions(sk, self.samples.index, self.n_vars)
        guides = self.guides.iloc[gi].copy()
        samples = self.samples.iloc[si].copy()
        uns = {}
        dropped = set(self.samples.index) - set(samples.index)
        for k, v in self.uns.items():
            if isinstance(v, pd.DataFrame) and "guide" in v.columns:
                v = v.loc[v["guide"].isin(guides.index)]
                v = v[[c for c in v.columns if c not in dropped]]
            elif isinstance(v, pd.DataFrame) and k == "repguide_mask":
                v = v.loc[v.index.intersection(guides.index)].reindex(guides.index)
            uns[k] = v
        return ReporterScreen(self.X[np.ix_(gi, si)], guides, samples,
                              {k: v[np.ix_(gi, si)] for k, v in self.layers.items()}, uns)

    # ------------------------------------------------------------- derived columns
    def get_guide_edit_rate(self, bcmatch_thres=1, prior_weight: Optional[float] = None, return_result=False,
                            count_layer="X_bcmatch", edit_layer="edits", condition_col="condition",
                            unsorted_condition_label=None):
        """Posterior-mean editing rate per guide from the reporter counts of the
        unsorted (control) samples (``ReporterScreen.py:448-530``, variant screens:
        no per-base normalisation)."""
        if self.layers.get(count_layer) is None or self.layers.get(edit_layer) is None:
            raise ValueError("edits or barcode matched guide counts not available.")
        if unsorted_condition_label is not None:
            cond = self.samples[condition_col].astype(str)
            idx = np.where(cond.map(lambda s: unsorted_condition_label in s))[0]
            if len(idx) == 0:
                raise ValueError(f"'{unsorted_condition_label}' is not found in ReporterScreen.samples"
                                 f"['{condition_col}'] that has values {self.samples[condition_col].unique()}.")
        else:
            idx = np.arange(self.n_vars)
        w = 1 if prior_weight is None else prior_weight
        n_edits = self.layers[edit_layer][:, idx].sum(axis=1)
        n_counts = self.layers[count_layer][:, idx].sum(axis=1)
        rate = (n_edits + w / 2) / (n_counts + w / 2)
        rate = np.where(n_counts < bcmatch_thres, np.nan, rate)
        if return_result:
            return rate
        self.guides["edit_rate"] = rate

    def log_norm(self, pseudocount: float = 1.0) -> np.ndarray:
        """log2 counts-per-million + pseudocount (perturb-tools' ``Screen.log_norm``;
        that package is not vendored in the reference, so this follows its published
        behaviour and is not pinned by a fixture)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log2(self.X / self.X.sum(axis=0, keepdims=True) * 1e6 + pseudocount)

    def log_fold_change_reps(self, cond1, cond2, rep_col="replicate", compare_col="condition",
                             pseudocount: float = 5) -> pd.DataFrame:
        """Per-replicate log2 fold change ``cond1 - cond2`` of the log-normalised counts,
        columns ``{rep}.{cond1}_{cond2}.lfc`` (the sgRNA table's columns, cf.
        ``docs/example_run_output/variant/bean_sgRNA_result.MixtureNormal+Acc.csv``)."""
        ln = self.log_norm(pseudocount)
        out = {}
        reps = self.samples[rep_col].astype(str)
        cond = self.samples[compare_col].astype(str)
        for rep in sorted(reps.unique()):
            i1 = np.where((reps == rep) & (cond == str(cond1)))[0]
            i2 = np.where((reps == rep) & (cond == str(cond2)))[0]
            if len(i1) != 1 or len(i2) != 1:
                continue
            out[f"{rep}.{cond1}_{cond2}.lfc"] = ln[:, i1[0]] - ln[:, i2[0]]
        return pd.DataFrame(out, index=self.guides.index)


def _write(self, out_path: str) -> None:
    """Write ``.h5ad`` (``ReporterScreen.write``, reference ``ReporterScreen.py:896-915``)."""
    h5ad_io.write_screen(self, out_path)


ReporterScreen.write = _write


def read_h5ad(path: str) -> ReporterScreen:
    """``be.read_h5ad`` (``bean/framework/ReporterScreen.py:1009-1011``)."""
    tree = h5ad_io.read_tree(path)
    obs, var = h5ad_io.to_pandas(tree["obs"]), h5ad_io.to_pandas(tree["var"])
    layers = {k: np.asarray(h5ad_io._densify(v)) for k, v in (tree.get("layers") or {}).items()}
    uns = h5ad_io.to_pandas(tree.get("uns") or {})
    return ReporterScreen(np.asarray(h5ad_io._densify(tree["X"])), obs, var, layers, uns)
