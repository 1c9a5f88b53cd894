"""Variant (edit) table of a tiling screen: what ``bean/cli/run.py:155-206`` assembles from
``annotate_edit`` (``bean/annotate/translate_allele.py:629-708``) and the effective-editing-rate
helpers (``bean/preprocessing/utils.py:254-312``), computed from the CSR allele -> edit map instead
of the dense ``(G, A-1, E)`` tensor."""
from __future__ import annotations

from typing import Collection, Dict, List, Optional

import numpy as np
import pandas as pd
import torch


def strsplit_edit(edit_str: str):
    """``translate_allele.py:629-647``: ``[chrom:]pos:ref>alt`` -> (chrom, pos, ref, alt)."""
    parts = edit_str.split(":")
    if len(parts) == 3:
        chrom, pos, transition = parts
    elif len(parts) == 2:
        pos, transition = parts
        chrom = None
    else:
        raise ValueError(f"{edit_str} is not in the correct format.")
    ref, alt = transition.split(">")
    return chrom, pos, ref, alt


def annotate_edit(edit_info: pd.DataFrame, edit_col: str = "edit", control_tag: Optional[str] = "CONTROL",
                  splice_sites: Optional[Collection[int]] = None) -> pd.DataFrame:
    """Classify edit strings into coding / noncoding and syn / missense / trunc / splicing / negctrl
    (``translate_allele.py:650-708``)."""
    df = edit_info.copy()
    df["group"] = ""
    df["int_pos"] = -1
    if "pos" not in df.columns:
        cols = list(zip(*df[edit_col].map(strsplit_edit))) if len(df) else [(), (), (), ()]
        df["chrom"], df["pos"], df["ref"], df["alt"] = [list(c) for c in cols]
    df["coding"] = ""
    is_coding = df.pos.map(lambda s: s.startswith("A"))
    df.loc[is_coding, "coding"] = "coding"
    df.loc[~is_coding, "coding"] = "noncoding"
    if control_tag is not None:
        is_ctrl = df.pos.map(lambda s: control_tag in s)
        df.loc[is_ctrl, "group"] = "negctrl"
        df.loc[is_ctrl, "coding"] = "negctrl"
    sel = (df.coding == "noncoding") & (df.group != "negctrl")
    df.loc[sel, "int_pos"] = df.loc[sel, "pos"].map(int)
    df.loc[(df.alt != df.ref) & (df.coding == "coding"), "group"] = "missense"
    df.loc[df.alt == "*", "group"] = "trunc"
    df.loc[(df.alt == df.ref) & (df.coding == "coding"), "group"] = "syn"
    if splice_sites is not None:
        df.loc[df.pos.isin(np.asarray(splice_sites).astype(str)), "group"] = "splicing"
    df.loc[df.int_pos < -100, "group"] = "negctrl"
    df.loc[df.int_pos < -100, "coding"] = "negctrl"
    return df


def variant_table(ndata, guide_names, guide_target_group, control_tag=None, splice_sites=None,
                  count_thres: int = 0) -> pd.DataFrame:
    """``target_info_df`` of a tiling screen (``bean/cli/run.py:165-200``): one row per edit with
    its annotation, the guides producing it, per-guide and total effective editing rates, the number
    of guides and of co-occurring variants."""
    E, G, A1 = ndata.n_edits, ndata.n_guides, ndata.n_max_alleles - 1
    edits = pd.Series(ndata.edit_index)
    df = annotate_edit(pd.DataFrame(edits).reset_index().rename(columns={"index": "edit"}),
                       control_tag=control_tag, splice_sites=splice_sites)
    # mean rate of every allele in the control samples (_obtain_effective_edit_rate, utils.py:254-296)
    rates = ndata.allele_counts_control / ndata.X_bcmatch_control[:, :, :, None]
    low = (ndata.X_bcmatch_control < count_thres)[:, :, :, None].expand(rates.shape)
    rates = torch.where(low, torch.full_like(rates, float("nan")), rates)
    mean_rates = rates.nanmean(dim=(0, 1))[:, 1:].cpu().numpy()  # (G, A - 1)
    ptr = ndata.a2e_ptr.cpu().numpy().astype(np.int64)
    idx = ndata.a2e_idx.cpu().numpy().astype(np.int64)
    rate_ge: List[Dict[int, float]] = [dict() for _ in range(E)]  # edit -> {guide: rate}
    n_guides = [set() for _ in range(E)]
    coocc = [set() for _ in range(E)]
    for slot in range(G * A1):
        es = idx[ptr[slot]:ptr[slot + 1]]
        if not len(es):
            continue
        g, a = divmod(slot, A1)
        r = mean_rates[g, a] / len(es)
        for e in es:
            n_guides[e].add(g)
            coocc[e].update(int(x) for x in es)
            if not np.isnan(r):
                rate_ge[e][g] = rate_ge[e].get(g, 0.0) + float(r)
    guide_names = np.asarray(guide_names)
    tg = np.asarray(guide_target_group)
    hit = [sorted(g for g, v in d.items() if v != 0) for d in rate_ge]
    df["guide_target_group"] = [",".join(np.unique(tg[h].astype(str)).tolist()) if h else "" for h in hit]
    df["effective_edit_rate"] = [float(sum(d.values())) for d in rate_ge]
    df["editing_guides"] = [",".join(guide_names[h].tolist()) if h else "" for h in hit]
    df["per_guide_editing_rates"] = [",".join(f"{d[g]:.3g}" for g in h if d[g] > 0) for d, h in zip(rate_ge, hit)]
    df["n_guides"] = [len(s) for s in n_guides]
    df["n_coocc"] = [max(len(s) - 1, 0) for s in coocc]
    return df


def guide_to_variant_df(target_info_df: pd.DataFrame) -> pd.DataFrame:
    """``_get_guide_to_variant_df`` (``bean/model/run.py:311-344``): per guide, the variants it
    produces and their per-variant editing rates.  As in the reference, an edit that no guide produces
    (empty ``editing_guides``) is listed under the guide name ``""`` with a missing rate; ``bean run`` joins
    the table on the screen's guides, so that row goes nowhere.  Pinned by ``tests/test_edit_golden.py``."""
    rows = []
    for edit, guides, rates in zip(target_info_df["edit"], target_info_df["editing_guides"],
                                   target_info_df["per_guide_editing_rates"]):
        if guides and pd.isnull(guides):
            continue
        names = str(guides).strip(",").split(",")
        values = [float(x) if x else np.nan for x in str(rates).strip(",").split(",")]
        rows.extend((g, edit, r) for g, r in zip(names, values))
    if not rows:
        return pd.DataFrame(columns=["variants", "per_variant_edit_rate"])
    df = pd.DataFrame(rows, columns=["guide", "variants", "per_variant_edit_rate"])
    return df.groupby("guide").agg(list)
