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

    @property
    def target_base_changes(self) -> dict:
        """``{"A": "G"}`` from ``uns["target_base_changes"]`` (or the older ``target_base_change``) -
        ``ReporterScreen.py:165-170``."""
        changes = self.uns["target_base_changes"] if "target_base_changes" in self.uns else self.uns["target_base_change"]
        return {bc[0]: bc[-1] for bc in str(changes).split(",")}

    # ------------------------------------------------------------- edits from the allele table
    def get_edit_from_allele(self, allele_count_key="allele_counts", allele_key="allele", return_result=False):
        """``uns["edit_counts"]``: per (guide, edit) the summed counts of the alleles that contain the
        edit (``ReporterScreen.py:595-621``).  Edits stay strings (``pos:rel_pos:strand:ref>alt``); the
        reference wraps them in ``Edit`` objects, whose ``str()`` is that string."""
        if allele_count_key not in self.uns:
            raise ValueError(f"No allele information stored: {list(self.uns)}")
        df = self.uns[allele_count_key].copy()
        df = df.loc[df[allele_key].map(str) != "", :]
        df["edits"] = df[allele_key].map(lambda a: str(a).split(","))
        df = (df[["guide", "edits"] + self.samples.index.tolist()].explode("edits")
              .groupby(["guide", "edits"]).sum().reset_index().rename(columns={"edits": "edit"}))
        if return_result:
            return df
        self.uns["edit_counts"] = df

    def get_edit_mat_from_uns(self, target_base_edit=None, match_target_position=None, rel_pos_start=0,
                              rel_pos_end=np.inf, rel_pos_is_reporter=False, target_pos_col="target_pos",
                              edit_count_key="edit_counts", reporter_length: int = 32,
                              reporter_right_flank_length: int = 6):
        """``layers["edits"]`` from ``uns[edit_count_key]`` (``ReporterScreen.py:352-446``): counts of the
        edits of the screen's base change that sit at the guide's target position (variant screens) or
        inside ``[rel_pos_start, rel_pos_end)`` of the spacer / reporter (tiling).  Returns the old layer.
        As in the reference, ``uns["reporter_length"]`` overrides the argument, the right-flank length
        does not (the reference looks it up under a misspelt key)."""
        from ..preprocessing.alleles import parse_nt_edit

        if target_base_edit is None:
            target_base_edit = self.target_base_changes
        if match_target_position is None:
            match_target_position = not self.tiling
        if "reporter_length" in self.uns:
            reporter_length = int(self.uns["reporter_length"])
        if edit_count_key not in self.uns:
            raise ValueError(f"Edit count isn't calculated or not provided with specified key `{edit_count_key}`. "
                             "Call .get_edit_from_allele(allele_count_key, allele_key)")
        edits = self.uns[edit_count_key].copy()
        old = self.layers["edits"].copy() if "edits" in self.layers else None
        cols = self.samples.index.tolist()
        parsed = edits["edit"].map(lambda e: parse_nt_edit(str(e)))
        ok = parsed.map(lambda p: p is not None and p[1] in target_base_edit and target_base_edit[p[1]] == p[2])
        edits = edits.loc[ok.values].reset_index(drop=True)
        rel_pos = parsed[ok.values].map(lambda p: p[0]).to_numpy(dtype=np.int64) if len(edits) else np.zeros(0, np.int64)
        gidx = self.guides.index.get_indexer(edits["guide"])
        keep = gidx >= 0
        if match_target_position:
            tpos = self.guides[target_pos_col].to_numpy()[np.where(keep, gidx, 0)]
            good = keep & (rel_pos == tpos)
        else:
            guide_len = self.guides["sequence"].map(len).to_numpy()[np.where(keep, gidx, 0)]
            start = reporter_length - reporter_right_flank_length - guide_len
            shift = 0 if rel_pos_is_reporter else start
            good = keep & (rel_pos >= rel_pos_start + shift) & (rel_pos < rel_pos_end + shift)
        mat = np.zeros(self.X.shape, dtype=np.float64)
        if good.any():
            vals = edits.loc[good, cols].to_numpy(dtype=np.float64).astype(np.int64)
            np.add.at(mat, gidx[good], vals)
        self.layers["edits"] = mat
        return old

    # ------------------------------------------------------------- derived columns
    def get_guide_edit_rate(self, normalize_by_editable_base: Optional[bool] = None, edited_bases=None,
                            editable_base_start=3, editable_base_end=8, bcmatch_thres=1,
                            prior_weight: Optional[float] = None, return_result=False,
                            count_layer="X_bcmatch", edit_layer="edits", condition_col="condition",
                            unsorted_condition_label=None):
        """Posterior-mean editing rate per guide from the reporter counts of the unsorted (control)
        samples (``ReporterScreen.py:448-530``); tiling screens also get ``edit_rate_norm``, the rate per
        editable base of the spacer window."""
        if self.layers.get(count_layer) is None or self.layers.get(edit_layer) is None:
            raise ValueError("edits or barcode matched guide counts not available.")
        if normalize_by_editable_base is None:
            normalize_by_editable_base = self.tiling
        n_sites = None
        if normalize_by_editable_base:
            if edited_bases is None:
                edited_bases = list(self.target_base_changes.keys())
            if isinstance(edited_bases, str):
                edited_bases = [edited_bases]
            for b in edited_bases:
                if b not in ("A", "C", "T", "G"):
                    raise ValueError("Specify the correct edited_base")
            n_sites = sum(self.guides["sequence"].map(lambda q, b=b: q[editable_base_start:editable_base_end].count(b))
                          for b in edited_bases).to_numpy(dtype=np.float64)
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
        if normalize_by_editable_base:
            with np.errstate(all="ignore"):
                norm = (n_edits + w / 2) / (n_counts * n_sites + w / 2)
            self.guides["edit_rate_norm"] = np.where(n_sites == 0, np.nan, norm)

    def log_norm(self, pseudocount: float = 1.0) -> np.ndarray:
        """log2 counts-per-million + pseudocount (perturb-tools' ``Screen.log_norm``; that package is not
        vendored in the reference, so this follows its published behaviour and is not pinned by a fixture).
        A sample without any read (the dummy samples ``bean qc`` adds for missing conditions) has CPM 0,
        not 0 / 0: the reference's own tests require ``bean qc -b`` to succeed on its ``*_missing.h5ad``
        files (``tests/test_qc.py:42-63``), which with NaN there ends in "too small number of replicate"."""
        tot = self.X.sum(axis=0, keepdims=True).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            cpm = np.where(tot > 0, self.X / np.where(tot > 0, tot, 1.0) * 1e6, 0.0)
        return np.log2(cpm + pseudocount)

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
