"""Golden vectors from the reference's own Pyro-free host classes and functions around `bean run`'s tiling path.

Run in the build container only (``/root/reference`` does not travel):

    python tests/golden/make_edit_golden.py

``import bean`` fails here (pyro-ppl is absent), but these pieces of the reference reach for nothing
outside the standard library, numpy and pandas:

* ``Edit`` and ``Allele`` (``bean/framework/Edit.py:8-159``; they call ``jaccard``, ``bean/utils/arithmetric.py``),
* ``AA_SET``, ``MutationType``, ``AminoAcidEdit``, ``AminoAcidAllele``, ``CodingNoncodingAllele``
  (``bean/framework/AminoAcidEdit.py:10-330``),
* ``strsplit_edit`` and ``annotate_edit`` (``bean/annotate/translate_allele.py:629-708``),
* ``_get_guide_to_variant_df`` (``bean/model/run.py:311-344``).

The script parses each file and takes exactly those ``ClassDef`` / ``FunctionDef`` / ``Assign`` nodes UNCHANGED,
compiles them (with ``from __future__ import annotations`` in force, as in their files) into a namespace that holds only
the third-party names they use, and evaluates them on every allele and edit string of the reference's three mini-screen
files (``tests/golden/*_mini_screen.h5ad``) plus a handful of typed cases (amino-acid alleles, uids, chromosomes).  No
stand-in for anything is written: a node that reached for something else would raise ``NameError``.  Only inputs and
outputs are stored (``edit_cases.json.gz``); no reference source is copied.  ``tests/test_edit_golden.py`` pins
``bean_amd.framework.Edit``, ``bean_amd.preprocessing.alleles`` and ``bean_amd.model.tiling_info`` to them.
"""
import __future__

import ast
import gzip
import json
import os
import re
import sys
import warnings
from enum import IntEnum
from typing import Collection, Iterable, Optional

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/bean"
FLAGS = __future__.annotations.compiler_flag


def take(path, names, ns, kinds=(ast.ClassDef, ast.FunctionDef)):
    """Compile the named top-level nodes of a reference file, unchanged, into ``ns``."""
    tree = ast.parse(open(path).read(), path)
    nodes = []
    for n in tree.body:
        if isinstance(n, kinds) and n.name in names:
            nodes.append(n)
        elif isinstance(n, ast.Assign) and len(n.targets) == 1 and getattr(n.targets[0], "id", None) in names:
            nodes.append(n)
    got = sorted(getattr(n, "name", None) or n.targets[0].id for n in nodes)
    assert got == sorted(names), (path, got, names)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec", flags=FLAGS, dont_inherit=True), ns)  # noqa: S102
    return ns


def reference_namespace():
    ns = {"np": np, "re": re, "Iterable": Iterable, "Optional": Optional}
    take(f"{REF}/utils/arithmetric.py", ["jaccard"], ns)
    take(f"{REF}/framework/Edit.py", ["Edit", "Allele"], ns)
    ns.update({"IntEnum": IntEnum, "warnings": warnings})
    take(f"{REF}/framework/AminoAcidEdit.py",
         ["AA_SET", "MutationType", "AminoAcidEdit", "AminoAcidAllele", "CodingNoncodingAllele"], ns)
    ns.update({"pd": pd, "Collection": Collection})
    take(f"{REF}/annotate/translate_allele.py", ["strsplit_edit", "annotate_edit"], ns)
    take(f"{REF}/model/run.py", ["_get_guide_to_variant_df"], ns)
    return ns


def mini_screen_strings():
    import bean_amd  # noqa: F401
    from bean_amd.framework import read_h5ad

    alleles, edits, control_guides = [], [], []
    for f in ("tiling_mini_screen", "survival_tiling_mini_screen", "var_mini_screen"):
        s = read_h5ad(os.path.join(HERE, f + ".h5ad"))
        alleles += s.uns["allele_counts"]["allele"].astype(str).tolist()
        edits += s.uns["edit_counts"]["edit"].astype(str).tolist()
        control_guides += [g for g in s.guides.index if "CONTROL" in g]
    uniq = lambda xs: list(dict.fromkeys(xs))  # noqa: E731
    return uniq(alleles), uniq(edits), uniq(control_guides)


def main():
    ns = reference_namespace()
    Edit, Allele = ns["Edit"], ns["Allele"]
    AAEdit, AAAllele, CNAllele = ns["AminoAcidEdit"], ns["AminoAcidAllele"], ns["CodingNoncodingAllele"]
    alleles, edits, control_guides = mini_screen_strings()
    rng = np.random.default_rng(20261005)
    out = {}

    # ---- Edit: every edit string of the three screens, plus chromosome / uid forms
    edit_strs = edits + ["chr19:11200120:28:+:A>G", "chr6:-5:3:-:C>T", "nan:7:7:+:G>-", "X!11200120:28:-:A>G",
                         "0:0:+:A>G", "-3:5:-:T>C"]
    rows = []
    for s in edit_strs:
        e = Edit.from_str(s)
        row = {"s": s, "repr": repr(e), "abs": e.get_abs_edit(), "pos": int(e.pos), "rel_pos": int(e.rel_pos),
               "strand": e.strand, "chrom": e.chrom, "uid": e.uid, "base_change": e.get_base_change(),
               "abs_base_change": e.get_abs_base_change()}
        e2 = Edit.from_str(s).set_uid("CONTROL_7_g2")
        row["abs_uid"], row["repr_uid"] = e2.get_abs_edit(), repr(e2)
        e3 = Edit.from_str(s).set_chrom("chr2")
        row["abs_chrom"], row["repr_chrom"] = e3.get_abs_edit(), repr(e3)
        rows.append(row)
    out["edits"] = rows
    # the constructor as bean count / the tests call it (Edit.py:12-34): rel_pos, bases, offset, strand
    ctor = []
    for rel_pos, ref, alt, chrom, offset, strand in [(3, "A", "G", None, None, 1), (3, "A", "G", None, 100, 1),
                                                      (3, "A", "G", "chr1", 100, -1), (0, "C", "T", None, -5, -1),
                                                      (28, "A", "G", "chr19", 11200148, -1)]:
        e = Edit(rel_pos, ref, alt, chrom=chrom, offset=offset, strand=strand)
        ctor.append({"args": [rel_pos, ref, alt, chrom, offset, strand], "repr": repr(e), "abs": e.get_abs_edit(),
                     "pos": int(e.pos)})
    out["edit_ctor"] = ctor
    # match_str on well- and ill-formed strings
    probes = edit_strs[:40] + ["", "not-an-edit", "12:3:+:A>G ", "12:3:+:a>g", "12:3:*:A>G", "chr1:12:3:+:A>GG",
                               "12:+:A>G", "g!12:3:+:A>G", "gg!12:3:+:A>G", "1:2:3:+:A>G", "chr1:2:3:+:*>-"]
    out["edit_match"] = [[s, bool(Edit.match_str(s))] for s in probes]
    bad = []
    for s in ["not-an-edit", "12:+:A>G", "", "CONTROL_1_g1!12:12:+:A>G"]:  # (its uid pattern admits ONE character)
        try:
            Edit.from_str(s)
            bad.append([s, None])
        except ValueError as exc:
            bad.append([s, str(exc)])
    out["edit_from_str_errors"] = bad
    # ordering / equality / hash of Edit objects: pools of edits sorted by the reference's __lt__
    pools = []
    for _ in range(40):
        pool = [edit_strs[i] for i in rng.choice(len(edit_strs), size=int(rng.integers(2, 12)), replace=True)]
        objs = [Edit.from_str(s) for s in pool]
        pools.append({"pool": pool, "sorted": [repr(e) for e in sorted(objs)], "n_distinct": len(set(objs)),
                      "eq_first": [bool(objs[0] == o) for o in objs], "lt_first": [bool(objs[0] < o) for o in objs],
                      "gt_first": [bool(objs[0] > o) for o in objs]})
    out["edit_pools"] = pools

    # ---- Allele: every allele string of the three screens
    rows = []
    for s in alleles + ["", " ", "chr19:11200120:28:-:A>G,chr19:11200140:8:-:A>G", "12:3:+:A>G,12:3:+:A>G"]:
        a = Allele.from_str(s)
        rng_ = a.get_range()
        row = {"s": s, "repr": repr(a), "n": len(a), "bool": bool(a), "match": bool(Allele.match_str(s)),
               "abs_sorted": [e.get_abs_edit() for e in sorted(a.edits)], "range": None if rng_ is None else list(rng_),
               "uid": a.get_uid()}
        b = Allele.from_str(s).set_uid("CONTROL_3_g1")
        row["repr_uid"], row["abs_uid_sorted"], row["get_uid_after"] = repr(b), [e.get_abs_edit() for e in sorted(b.edits)], b.get_uid()
        rows.append(row)
    out["alleles"] = rows
    # has_edit / has_other_edit / jaccard / equality / __lt__ / map_to_closest on random pairs and lists
    qs = []
    for _ in range(200):
        sa, sb = (alleles[i] for i in rng.choice(len(alleles), 2))
        a, b = Allele.from_str(sa), Allele.from_str(sb)
        e = next(iter(sorted(b.edits)))
        qs.append({"a": sa, "b": sb, "jaccard": float(a.get_jaccard(b)), "eq": bool(a == b), "lt": bool(a < b),
                   "q": [e.ref_base, e.alt_base, int(e.pos), int(e.rel_pos)],
                   "has_edit_pos": bool(a.has_edit(e.ref_base, e.alt_base, pos=e.pos)),
                   "has_edit_rel": bool(a.has_edit(e.ref_base, e.alt_base, rel_pos=e.rel_pos)),
                   "has_other_pos": bool(a.has_other_edit(e.ref_base, e.alt_base, pos=e.pos)),
                   "has_other_rel": bool(a.has_other_edit(e.ref_base, e.alt_base, rel_pos=e.rel_pos))})
    out["allele_queries"] = qs
    closest = []
    for _ in range(60):
        sa = alleles[int(rng.integers(len(alleles)))]
        cand = [alleles[i] for i in rng.choice(len(alleles), size=int(rng.integers(0, 9)))]
        if rng.random() < 0.5 and cand:
            # a near neighbour: the allele itself minus its last edit
            cand[int(rng.integers(len(cand)))] = ",".join(sa.split(",")[:-1]) or sa
        thr = float(rng.choice([0.2, 0.5, 0.8]))
        prio = pd.Series(rng.integers(0, 4, size=len(cand)).astype(float)) if (cand and rng.random() < 0.5) else None
        got = Allele.from_str(sa).map_to_closest([Allele.from_str(c) for c in cand], jaccard_threshold=thr,
                                                 merge_priority=prio)
        closest.append({"a": sa, "cand": cand, "thr": thr, "prio": None if prio is None else prio.tolist(),
                        "closest": repr(got)})
    out["allele_closest"] = closest

    # ---- AminoAcidEdit / AminoAcidAllele / CodingNoncodingAllele on typed strings (the mini-screens hold none)
    aa_strs = ["35:V>A", "LDLR:35:V>A", "40:Q>*", "7:L>L", "LDLR:100:T>I", "100:T>I", "9:A>/", "LDLR:9:*>Q"]
    out["aa_edits"] = [{"s": s, "repr": repr(AAEdit.from_str(s)), "abs": AAEdit.from_str(s).get_abs_edit(),
                        "severity": int(AAEdit.from_str(s)._severity()), "gene": AAEdit.from_str(s).gene}
                       for s in aa_strs]
    pools = []
    for _ in range(30):
        pool = [aa_strs[i] for i in rng.choice(len(aa_strs), size=int(rng.integers(2, 6)), replace=False)]
        # the reference compares positions as the strings from_str leaves them; genes by truthiness then order
        # (a mixed pool can make its __lt__ return None = falsy; sorted() still terminates)
        objs = [AAEdit.from_str(s) for s in pool]
        pools.append({"pool": pool, "sorted": [repr(e) for e in sorted(objs)],
                      "eq_first": [bool(objs[0] == o) for o in objs]})
    out["aa_pools"] = pools
    nt_some = [a for a in alleles if 0 < len(a) < 60][:12]
    cn_strs = (["|".join([aa, nt]) for aa, nt in zip(["35:V>A", "40:T>I,35:V>A", "", "LDLR:7:L>L", "40:Q>*,7:L>L", ""],
                                                      nt_some[:5] + [""])] +
               ["35:V>A|", "|" + nt_some[6], "LDLR:40:Q>*,LDLR:35:V>A|" + nt_some[7], "|"])
    rows = []
    for s in cn_strs:
        c = CNAllele.from_str(s)
        rows.append({"s": s, "repr": repr(c), "n": len(c), "bool": bool(c), "match": bool(CNAllele.match_str(s)),
                     "aa_abs": sorted(e.get_abs_edit() for e in c.aa_allele.edits),
                     "nt_abs": [e.get_abs_edit() for e in sorted(c.nt_allele.edits)],
                     "has_coding": bool(c.has_coding()), "most_severe": float(c.get_most_severe()), "uid": c.uid})
    out["cn_alleles"] = rows
    out["cn_match"] = [[s, bool(CNAllele.match_str(s))] for s in ["a|b|c", "35:V>A", "35:V>A|12:3:+:A>G"]]

    # ---- annotate_edit / strsplit_edit on the absolute edit strings these screens produce (+ coding / control forms)
    abs_edits = list(dict.fromkeys(e.get_abs_edit() for s in alleles for e in sorted(Allele.from_str(s).edits)))
    ctrl = [Edit.from_str(s).set_uid(g).get_abs_edit() for s, g in zip(edits[:25], (control_guides * 25)[:25])] \
        if control_guides else []
    typed = ["A35:V>A", "A36:V>V", "A40:Q>*", "LDLR:A35:V>A", "chr19:11200120:T>C", "-250:A>G", "-99:A>G", "CONTROL_3!5:A>G"]
    ann_in = abs_edits[:400] + ctrl + typed
    out["strsplit"] = [[s, list(ns["strsplit_edit"](s))] for s in ann_in[:50] + typed]
    splice = np.array(sorted({int(s.split(":")[0]) for s in abs_edits[:400:7]}))
    for tag, kw in (("default", {}), ("notag", {"control_tag": None}), ("splice", {"splice_sites": splice})):
        # (without a control tag the reference's int() raises on a uid'd position: those rows stay out of that case)
        rows_in = [e for e in ann_in if "!" not in e] if tag == "notag" else ann_in
        df = ns["annotate_edit"](pd.DataFrame({"edit": rows_in}), **kw)
        out[f"annotate_{tag}"] = {c: df[c].tolist() for c in ("edit", "chrom", "pos", "ref", "alt", "coding", "group", "int_pos")}
    out["annotate_splice_sites"] = splice.tolist()

    # ---- _get_guide_to_variant_df on target tables shaped as bean run assembles them (cli/run.py:165-200)
    g2v = []
    guide_pool = [f"g{i}" for i in range(12)]
    for case in range(6):
        n = 15
        eg, rates = [], []
        for _ in range(n):
            k = int(rng.integers(0, 4)) if case else 2
            gs = [guide_pool[i] for i in rng.choice(len(guide_pool), size=k, replace=False)]
            eg.append(",".join(gs))
            rates.append(",".join(f"{x:.3g}" for x in rng.random(k)))
        if case == 2:
            eg[3], rates[3] = "", ""          # an edit no guide produces
        if case == 3:
            eg[0], rates[0] = "g1,g2,", "0.5,0.25,"  # trailing separators are stripped
        tdf = pd.DataFrame({"edit": abs_edits[case * n:(case + 1) * n], "editing_guides": eg,
                            "per_guide_editing_rates": rates})
        res = ns["_get_guide_to_variant_df"](tdf)
        g2v.append({"edit": tdf["edit"].tolist(), "editing_guides": eg, "per_guide_editing_rates": rates,
                    "index": res.index.tolist(), "variants": res["variants"].tolist(),
                    "per_variant_edit_rate": [[None if (isinstance(x, float) and np.isnan(x)) else x for x in r]
                                              for r in res["per_variant_edit_rate"].tolist()]})
    out["g2v"] = g2v

    path = os.path.join(HERE, "edit_cases.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(out, sort_keys=True).encode())
    print(f"wrote {path}: {os.path.getsize(path)} bytes; {len(out['edits'])} edits, {len(out['alleles'])} alleles, "
          f"{len(ann_in)} annotated edits")


if __name__ == "__main__":
    main()
