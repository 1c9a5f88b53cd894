"""Allele tables of tiling screens -> tensors (crispr-bean_amd/preprocessing/alleles.py) on the
reference's tiling mini-screen data file and on small hand-made tables.  CPU."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.framework import h5ad_io, read_h5ad
from bean_amd.model.tiling_info import annotate_edit, guide_to_variant_df, variant_table
from bean_amd.preprocessing import alleles
from bean_amd.preprocessing.screen_data import DATACLASS_DICT

TILING = os.path.join(os.path.dirname(__file__), "golden", "tiling_mini_screen.h5ad")
needs_h5 = pytest.mark.skipif(not os.path.exists(h5ad_io.HELPER_PYTHON),
                              reason="no h5py interpreter")


def test_edit_strings_follow_the_reference_format():
    # Edit.from_str(...).get_abs_edit(): sense-strand bases, absolute position (Edit.py:36-87)
    assert alleles.nt_edit_abs("11200120:28:-:A>G")[0] == "11200120:T>C"
    assert alleles.nt_edit_abs("11200120:28:+:A>G")[0] == "11200120:A>G"
    assert alleles.nt_edit_abs("chr19:11200120:28:+:A>G")[0] == "chr19:11200120:A>G"
    # control guides carry a uid: relative position, `uid!` prefix
    assert alleles.nt_edit_abs("11200120:28:-:A>G", uid="CONTROL_1")[0] == "CONTROL_1!28:T>C"
    assert alleles.aa_edit_abs("35:V>A")[0] == "A35:V>A"
    assert alleles.aa_edit_abs("LDLR:35:V>A")[0] == "LDLR:A35:V>A"
    with pytest.raises(ValueError):
        alleles.nt_edit_abs("not-an-edit")
    # CodingNoncodingAllele: amino-acid edits first, then nucleotide edits, each by position
    assert alleles.allele_edits("40:T>I,35:V>A|11200140:8:-:A>G,11200120:28:-:A>G") == [
        "A35:V>A", "A40:T>I", "11200120:T>C", "11200140:T>C"]
    assert alleles.allele_edits("") == []


def _small_table():
    return pd.DataFrame({
        "guide": ["g2", "g1", "g2", "g1", "g1"],
        "allele": ["100:3:+:A>G", "50:2:+:A>G,52:4:+:A>G", "100:3:+:A>G,103:6:+:A>G", "52:4:+:A>G", "50:2:+:A>G"],
        "s1": [5, 7, 1, 2, 0],
        "s2": [3, 9, 0, 4, 1],
    })


def test_numbering_csr_and_mask_on_a_small_table():
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        at = alleles.build_allele_tensors(_small_table(), ["g1", "g2", "g3"])
    assert at.n_max_alleles == 4 and at.n_dropped == 0
    # edits numbered over the table grouped by (sorted) guide, alleles in table order
    assert at.edit_index == {"50:A>G": 0, "52:A>G": 1, "100:A>G": 2, "103:A>G": 3}
    ptr, idx = at.a2e_ptr.numpy(), at.a2e_idx.numpy()
    slots = [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(3 * 3)]
    assert slots == [[0, 1], [1], [0], [2], [2, 3], [], [], [], []]
    assert at.allele_mask.tolist() == [[True, True, True, True], [True, True, True, False],
                                       [True, False, False, False]]
    samples = pd.DataFrame({"replicate_id": [0, 1], "cond_id": [0, 0]}, index=["s1", "s2"])
    bc = np.array([[20, 10], [4, 30], [8, 8]])
    cnt = alleles.allele_count_tensor(at, samples, ["g1", "g2", "g3"], bc, 2, "cond_id")
    assert cnt.shape == (2, 1, 3, 4)
    assert cnt[0, 0].tolist() == [[11, 7, 2, 0], [0, 5, 1, 0], [8, 0, 0, 0]]  # unedited = bcmatch - listed, >= 0
    assert cnt[1, 0].tolist() == [[0, 9, 4, 1], [27, 3, 0, 0], [8, 0, 0, 0]]


def test_too_many_alleles_is_an_error_and_truncation_is_opt_in(monkeypatch):
    """The reference has no bound on alleles per guide; beyond what the kernels hold the build stops
    (naming `bean filter`) rather than silently fitting a different model; BEAN_MAX_ALLELES_PER_GUIDE
    opts in to keeping the most abundant alleles."""
    df = _small_table()
    monkeypatch.delenv("BEAN_MAX_ALLELES_PER_GUIDE", raising=False)
    with pytest.raises(ValueError, match="bean filter"):
        alleles.build_allele_tensors(df, ["g1", "g2"], max_alleles=3)
    monkeypatch.setenv("BEAN_MAX_ALLELES_PER_GUIDE", "3")
    with pytest.warns(UserWarning, match="most abundant alleles"):
        at = alleles.build_allele_tensors(df, ["g1", "g2"])
    assert at.n_max_alleles == 3 and at.n_dropped == 1  # g1's rarest allele (total 1) is folded away
    assert "50:A>G" in at.edit_index and len(at.reindexed) == 4


@needs_h5
def test_tiling_mini_screen_builds_consistent_tensors():
    s = read_h5ad(TILING)
    s.samples["replicate"] = s.samples["replicate"].astype(str)
    d = DATACLASS_DICT["sorting"]["MultiMixtureNormal"](
        s, sample_mask_column=None, allele_df_key="allele_counts", control_condition="bulk")
    # the unfiltered table: up to 230 edited alleles per guide, kept as they are
    assert (d.n_reps, d.n_condits, d.n_guides, d.n_max_alleles) == (2, 5, 30, 231) and d.n_alleles_dropped == 0
    assert d.n_edits == len(d.edit_index) == d.n_targets and d.n_edits > 20
    d.validate()
    # alleles partition the barcode-matched reads
    assert torch.equal(d.allele_counts.sum(-1), d.X_bcmatch)
    assert torch.equal(d.allele_counts_control.sum(-1), d.X_bcmatch_control)
    assert d.allele_counts_control.shape == (2, 1, 30, 231) and d.pi_a0.shape == (30,)
    # the dense view agrees with a direct parse of the kept rows
    dense = d.allele_to_edit_dense()
    assert dense.shape == (30, 230, d.n_edits) and set(np.unique(dense.numpy())) <= {0.0, 1.0}
    tbl = s.uns["allele_counts"]
    tot = tbl[[c for c in tbl.columns if c.startswith("rep")]].sum(axis=1)
    g0 = s.guides.index[0]
    kept = set()
    for a in tbl.loc[tbl.guide == g0, "allele"]:
        kept.update(alleles.allele_edits(a))
    got = {e for e, j in d.edit_index.items() if dense[0, :, j].sum() > 0}
    assert got == kept
    # variant table of the CLI
    vt = variant_table(d, s.guides.index.values, s.guides["target_group"].values)
    assert len(vt) == d.n_edits and (vt["n_guides"] >= 1).all() and (vt["coding"] == "noncoding").all()
    assert (vt["effective_edit_rate"] >= 0).all() and vt["editing_guides"].map(len).gt(0).any()
    g2v = guide_to_variant_df(vt)
    assert set(g2v.index) - {""} <= set(s.guides.index)  # ("" = edits no guide produces, as in the reference)
    assert {"variants", "per_variant_edit_rate"} <= set(g2v.columns)


def test_annotate_edit_groups():
    df = annotate_edit(pd.DataFrame({"edit": ["A35:V>A", "A36:V>V", "A40:Q>*", "11200120:T>C", "CONTROL_3!5:A>G"]}),
                       control_tag="CONTROL")
    assert df["coding"].tolist() == ["coding", "coding", "coding", "noncoding", "negctrl"]
    assert df["group"].tolist() == ["missense", "syn", "trunc", "", "negctrl"]
    assert df["int_pos"].tolist() == [-1, -1, -1, 11200120, -1]
