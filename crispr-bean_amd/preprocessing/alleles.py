"""Allele tables of tiling screens -> tensors.

What ``TilingReporterScreenData._post_init`` and its helpers compute
(``bean/preprocessing/data_class.py:574-873``, ``bean/preprocessing/utils.py:149-174``,
``bean/framework/Edit.py:36-159``, ``bean/framework/AminoAcidEdit.py:43-75,180-186``):
the per-guide allele numbering, the edit index, the allele -> edit map, per-sample allele
count tensors and the allele mask.

By design the allele -> edit map is produced directly in CSR form (the reference fills a dense
``(G, A-1, E)`` 0/1 tensor cell by cell, ``data_class.py:656-699``).

Two deliberate differences, both stated to the user when they apply:
* the edits of one allele are visited in sorted order (the reference iterates a Python ``set`` of
  objects whose hash depends on the process' string-hash seed, so its edit numbering is not
  reproducible run to run; any numbering is equivalent up to a permutation of the rows of the
  result table);
* alleles per guide: up to 32 (the unedited one included) run in the register-resident kernels
  (``libbean_hip.so`` holds 8, ``libbean_hip_a16.so`` 16, ``libbean_hip_a32.so`` 32), up to ``MAX_ALLELES`` = 512 in the
  allele-parallel kernels (``csrc/bean_tiling_wide.hpp``: 256 in the default build, 512 in the 16-allele build, which the
  engine loads for the wider tables), so unfiltered tables fit as they are (the
  reference's own ``tests/data/tiling_mini_screen.h5ad`` has 230).  Beyond 512 the build stops with an
  error that names ``bean filter``; ``BEAN_MAX_ALLELES_PER_GUIDE=N`` opts in to keeping each guide's
  ``N - 1`` most abundant alleles instead (the reads of the dropped ones fall into the unedited allele
  exactly as they do for alleles ``bean filter`` removes, ``data_class.py:773-777``) - a different model
  from the reference's, hence never the default.
"""
from __future__ import annotations

import os
import re
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch

MAX_ALLELES = 512  # csrc/bean_tiling_wide.hpp: kWideMaxA of the wider builds (libbean_hip.so itself holds 256)
_REV = {"A": "T", "C": "G", "T": "A", "G": "C", "-": "-"}
_NT_EDIT = re.compile(r"(((chr)?\w+|nan):)?-?\d+:-?\d+:[+-]:[A-Z*-]>[A-Z*-]")
_NT_EDIT_UID = re.compile(r"[\w*]!-?\d+:-?\d+:[+-]:[A-Z*-]>[A-Z*-]")


def nt_edit_abs(edit_str: str, uid: Optional[str] = None) -> Tuple[str, int]:
    """``Edit.from_str(s).get_abs_edit()`` (``Edit.py:36-87``): sense-strand representation
    ``[chrom:]pos:ref>alt``; with a uid (control guides) ``uid![chrom:]rel_pos:ref>alt``.
    Also returns the position used for sorting."""
    s = edit_str
    if not (_NT_EDIT.fullmatch(s) or _NT_EDIT_UID.fullmatch(s)):  # (a uid inside the string is ONE character, Edit.py:71-72)
        raise ValueError(f"{edit_str} doesn't match with Edit string format.")
    if "!" in s:
        uid_in, s = s.split("!")
        uid = uid if uid is not None else uid_in
    parts = s.split(":")
    chrom = None
    if len(parts) == 5:
        chrom, pos, rel_pos, strand, change = parts
    else:
        pos, rel_pos, strand, change = parts
    ref, alt = change.split(">")
    if strand == "-":
        ref, alt = _REV[ref], _REV[alt]
    pre = f"{chrom}:" if chrom else ""
    if uid is not None:
        return f"{uid}!{pre}{int(rel_pos)}:{ref}>{alt}", int(pos)
    return f"{pre}{int(pos)}:{ref}>{alt}", int(pos)


def parse_nt_edit(edit_str: str) -> Optional[Tuple[int, str, str]]:
    """``(rel_pos, ref_base, alt_base)`` of a nucleotide edit string as ``Edit.from_str`` reads it
    (``Edit.py:36-67``: ``[uid!][chrom:]pos:rel_pos:strand:ref>alt``, bases as written, i.e. on the
    guide's strand), or None for anything else (amino-acid edits)."""
    s = edit_str.split("!")[1] if "!" in edit_str else edit_str
    if not _NT_EDIT.fullmatch(s):
        return None
    parts = s.split(":")
    rel_pos, change = parts[-3], parts[-1]
    ref, alt = change.split(">")
    return int(rel_pos), ref, alt


def aa_edit_abs(edit_str: str) -> Tuple[str, int]:
    """``AminoAcidEdit.from_str(s).get_abs_edit()`` (``AminoAcidEdit.py:52-72``):
    ``[gene:]A<pos>:ref>alt``."""
    parts = edit_str.split(":")
    gene = None
    if len(parts) == 2:
        pos, change = parts
    else:
        gene, pos, change = parts
    ref, alt = change.split(">")
    return f"{gene + ':' if gene else ''}A{int(pos)}:{ref}>{alt}", int(pos)


def allele_edits(allele_str: str, uid: Optional[str] = None) -> List[str]:
    """Edits of one allele as absolute edit strings: ``Allele`` (comma separated nucleotide edits)
    or ``CodingNoncodingAllele`` (``aa_allele|nt_allele``; amino-acid edits first, as
    ``get_allele_to_edit_tensor`` lists them, ``data_class.py:674-677``)."""
    s = str(allele_str).strip()
    out: List[Tuple[int, int, str]] = []
    if "|" in s:
        aa, nt = s.split("|")
        for e in filter(None, (x.strip() for x in aa.split(","))):
            name, pos = aa_edit_abs(e)
            out.append((0, pos, name))
        s = nt
    for e in filter(None, (x.strip() for x in s.split(","))):
        name, pos = nt_edit_abs(e, uid)
        out.append((1, pos, name))
    out.sort()
    seen, names = set(), []
    for _, _, name in out:
        if name not in seen:
            seen.add(name)
            names.append(name)
    return names


class AlleleTensors:
    """Result of :func:`build_allele_tensors`."""

    def __init__(self):
        self.n_max_alleles = 2
        self.edit_index: Dict[str, int] = {}
        self.a2e_ptr = self.a2e_idx = None
        self.allele_mask = None
        self.guide_alleles: Dict[str, List[str]] = {}
        self.reindexed: Optional[pd.DataFrame] = None
        self.n_dropped = 0


def build_allele_tensors(allele_df: pd.DataFrame, guide_index: Sequence[str], allele_col: Optional[str] = None,
                         control_guide_tag: Optional[str] = None, max_alleles: int = MAX_ALLELES) -> AlleleTensors:
    """Number the alleles of every guide, index the edits and build the CSR allele -> edit map.

    ``allele_df``: one row per (guide, allele) with per-sample count columns
    (``screen.uns[allele_df_key]``)."""
    df = allele_df.reset_index(drop=True)
    if allele_col is None:
        allele_col = "aa_allele" if "aa_allele" in df.columns else "allele"
    if allele_col not in df.columns or "guide" not in df.columns:
        raise ValueError(f"the allele table needs `guide` and `{allele_col}` columns; found {list(df.columns)}")
    df[allele_col] = df[allele_col].astype(str)
    sample_cols = [c for c in df.columns if c not in ("guide", allele_col, "index", "allele_id")]
    out = AlleleTensors()
    out.allele_col = allele_col
    # allele_id_for_guide = 1, 2, ... in table order within each guide (reindex_allele_df, 701-734)
    df["allele_id_for_guide"] = df.groupby("guide", sort=False).cumcount() + 1
    n_max = int(df["allele_id_for_guide"].max()) + 1 if len(df) else 1
    opt_in = os.environ.get("BEAN_MAX_ALLELES_PER_GUIDE")
    if opt_in is not None:
        max_alleles = min(max_alleles, max(int(opt_in), 2))
    if n_max > max_alleles:
        if opt_in is None:
            raise ValueError(
                f"the allele table has up to {n_max - 1} edited alleles per guide; the kernels hold {max_alleles - 1}. "
                "Filter the table first (`bean filter` writes filtered `allele_counts_*` tables; pass one with "
                "--allele-df-key), or set BEAN_MAX_ALLELES_PER_GUIDE=N to keep each guide's N - 1 most abundant "
                "alleles (the rest are counted as unedited: this changes the fitted model).")
        total = df[sample_cols].sum(axis=1)
        rank = total.groupby(df["guide"], sort=False).rank(method="first", ascending=False)
        keep = rank <= (max_alleles - 1)
        out.n_dropped = int((~keep).sum())
        warnings.warn(
            f"BEAN_MAX_ALLELES_PER_GUIDE={max_alleles}: the allele table has up to {n_max - 1} edited alleles per "
            f"guide; keeping each guide's {max_alleles - 1} most abundant alleles ({out.n_dropped} rows folded into "
            "the unedited allele).")
        df = df.loc[keep].reset_index(drop=True)
        df["allele_id_for_guide"] = df.groupby("guide", sort=False).cumcount() + 1
        n_max = int(df["allele_id_for_guide"].max()) + 1 if len(df) else 1
    out.n_max_alleles = max(n_max, 2)
    A1 = out.n_max_alleles - 1

    def uid_of(guide):
        return guide if (control_guide_tag is not None and control_guide_tag in guide) else None

    # the reference numbers edits over the table grouped by guide name (groupby sorts the guides)
    edits_of_row = [allele_edits(a, uid_of(g)) for g, a in zip(df["guide"], df[allele_col])]
    if control_guide_tag is not None:
        tagged = [control_guide_tag in g for g in df["guide"]]
        assert any(tagged) or not len(df), "uid not assinged."
    df["_edits"] = edits_of_row
    for names in df.sort_values("guide", kind="stable")["_edits"]:
        for name in names:
            if name not in out.edit_index:
                out.edit_index[name] = len(out.edit_index)
    # CSR over slots s = g * (A - 1) + (allele_id - 1), guides in screen order
    gpos = {g: i for i, g in enumerate(guide_index)}
    G = len(guide_index)
    per_slot: List[List[int]] = [[] for _ in range(G * A1)]
    n_valid = np.zeros(G, dtype=np.int64)
    for g, aid, names in zip(df["guide"], df["allele_id_for_guide"], df["_edits"]):
        if g not in gpos:
            continue
        gi = gpos[g]
        per_slot[gi * A1 + (aid - 1)] = [out.edit_index[n] for n in names]
        n_valid[gi] = max(n_valid[gi], aid)
    counts = np.fromiter((len(x) for x in per_slot), dtype=np.int64, count=G * A1)
    ptr = np.zeros(G * A1 + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    idx = np.fromiter((e for x in per_slot for e in x), dtype=np.int64, count=int(ptr[-1]))
    out.a2e_ptr = torch.as_tensor(ptr.astype(np.int32))
    out.a2e_idx = torch.as_tensor(idx.astype(np.int32))
    # allele mask: the unedited allele and every listed allele of the guide (get_allele_mask, 851-873)
    mask = np.zeros((G, out.n_max_alleles), dtype=bool)
    mask[:, 0] = True
    for gi in range(G):
        mask[gi, 1:n_valid[gi] + 1] = True
    out.allele_mask = torch.as_tensor(mask)
    out.reindexed = df.drop(columns=["_edits"])
    out.sample_cols = sample_cols
    return out


def allele_count_tensor(at: AlleleTensors, samples: pd.DataFrame, guide_index: Sequence[str], bcmatch: np.ndarray,
                        n_reps: int, id_col: str) -> torch.Tensor:
    """``transform_allele`` (``data_class.py:736-790``): (R, n_conditions, G, A) float32 counts of
    the numbered alleles; allele 0 = barcode-matched reads not assigned to a listed allele, floored
    at 0.  ``samples`` must carry ``replicate_id`` and ``id_col``; ``bcmatch`` is (G, n_samples)."""
    df = at.reindexed
    G, A = len(guide_index), at.n_max_alleles
    gpos = {g: i for i, g in enumerate(guide_index)}
    conds = list(pd.unique(samples[id_col]))
    out = torch.empty((n_reps, len(conds), G, A), dtype=torch.float32)
    rows = df["guide"].map(gpos)
    ok = rows.notna().values
    gi = rows[ok].astype(int).values
    ai = df["allele_id_for_guide"].values[ok]
    for i in range(n_reps):
        for j, cond in enumerate(conds):
            hit = np.where((samples["replicate_id"].values == i) & (samples[id_col].values == cond))[0]
            assert len(hit) == 1, (i, j, hit)
            name = samples.index[hit[0]]
            if name not in df.columns:
                raise ValueError(f"the allele table has no column for sample {name}")
            m = np.zeros((G, A), dtype=np.int64)
            m[gi, ai] = df[name].values[ok].astype(np.int64)
            rest = bcmatch[:, hit[0]].astype(np.int64) - m[:, 1:].sum(axis=1)
            m[:, 0] = np.maximum(rest, 0)
            out[i, j] = torch.as_tensor(m, dtype=torch.float32)
    return out
