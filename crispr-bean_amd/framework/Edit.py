"""Edit / allele value types of the allele tables (``screen.uns["allele_counts*"]``).

Mirror of the reference's public ``bean.Edit``, ``bean.Allele``, ``bean.AminoAcidEdit``, ``bean.AminoAcidAllele`` and
``bean.CodingNoncodingAllele`` (``bean/framework/Edit.py:8-159``, ``bean/framework/AminoAcidEdit.py:10-330``): same
constructor arguments, string forms, ordering, equality and hashing, so that allele tables written by ``bean count`` /
``bean filter`` parse to the same objects and print back to the same strings.  Every string form, sort order and
predicate below is pinned to the reference's own classes by ``tests/test_edit_golden.py`` (fixtures:
``tests/golden/make_edit_golden.py``) on every allele of the reference's mini-screen files.

String forms
    nucleotide edit      ``[chrom:]pos:rel_pos:strand:ref>alt``       (``uid!`` in front of ``pos`` when a uid is set)
    absolute edit        ``[chrom:]pos:ref>alt`` on the sense strand  (``uid![chrom:]rel_pos:ref>alt`` with a uid)
    amino-acid edit      ``[gene:]pos:ref>alt``; absolute ``[gene:]A<pos>:ref>alt``
    allele               edits joined by ``,`` in sorted order; coding/noncoding allele ``aa_allele|nt_allele``

Behaviours of the reference that are kept on purpose because tables in the wild depend on them: a parsed uid may be ONE
word character long (longer uids are set with ``set_uid``); ``AminoAcidEdit.from_str`` keeps the position as the string
it read, so amino-acid edits order by the position's TEXT; ``AminoAcidEdit.match_str`` asks for a strand field that
amino-acid edits do not have.
"""
from __future__ import annotations

import re
import warnings
from enum import IntEnum
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_COMPLEMENT = {"A": "T", "C": "G", "T": "A", "G": "C", "-": "-"}
_STRAND_SIGN = {"+": 1, "-": -1}
_SIGN_STRAND = {1: "+", -1: "-"}
_BODY = r"-?\d+:-?\d+:[+-]:[A-Z*-]>[A-Z*-]"
_RE_PLAIN = re.compile(r"(((chr)?\w+|nan):)?" + _BODY)
_RE_UID = re.compile(r"[\w*]!" + _BODY)

AA_SET = frozenset("ACDEFGHIKLMNPQRSTVWY*/")


def jaccard(a, b) -> float:
    """|a ∩ b| / |a ∪ b| of two collections; 1 when both are empty (``bean/utils/arithmetric.py``)."""
    both = len(set(a).intersection(b))
    either = len(a) + len(b) - both
    return 1 if either == 0 else float(both) / either


class Edit:
    """One base change seen in a reporter read: position relative to the guide (``rel_pos``), absolute position
    (``pos = offset + rel_pos * strand``), bases as written on the guide's strand."""

    reverse_map = _COMPLEMENT
    strand_map = _STRAND_SIGN

    def __init__(self, rel_pos: int, ref_base: str, alt_base: str, chrom: Optional[str] = None,
                 offset: Optional[int] = None, strand: int = 1, unique_identifier=None):
        if strand not in (1, -1):
            raise AssertionError(f"strand must be +1 or -1, not {strand!r}")
        self.chrom, self.rel_pos = chrom, rel_pos
        self.ref_base, self.alt_base = ref_base, alt_base
        self.uid = unique_identifier
        self.strand = _SIGN_STRAND[strand]
        self.pos = rel_pos if offset is None else offset + rel_pos * strand

    # ---- parsing
    @classmethod
    def match_str(cls, edit_str) -> bool:
        if isinstance(edit_str, Edit):
            return True
        return bool(_RE_PLAIN.fullmatch(edit_str) or _RE_UID.fullmatch(edit_str))

    @classmethod
    def from_str(cls, edit_str):
        if type(edit_str) is Edit:
            return edit_str
        if not cls.match_str(edit_str):
            raise ValueError(f"{edit_str} doesn't match with Edit string format.")
        uid, body = edit_str.split("!") if "!" in edit_str else (None, edit_str)
        fields = body.split(":")
        chrom = fields.pop(0) if len(fields) == 5 else None
        pos, rel_pos, sign = int(fields[0]), int(fields[1]), _STRAND_SIGN[fields[2]]
        ref, alt = fields[3].split(">")
        return cls(rel_pos, ref, alt, chrom=chrom, offset=pos - rel_pos * sign, strand=sign, unique_identifier=uid)

    # ---- string forms
    def _sense(self) -> Tuple[str, str]:
        if self.strand == "-":
            return _COMPLEMENT[self.ref_base], _COMPLEMENT[self.alt_base]
        return self.ref_base, self.alt_base

    def _chrom_prefix(self) -> str:
        return f"{self.chrom}:" if self.chrom else ""

    def get_abs_edit(self) -> str:
        """The edit on the sense strand at its absolute position; a uid'd edit (control guides, whose positions mean
        nothing on the genome) keeps its position relative to the guide."""
        ref, alt = self._sense()
        if self.uid is not None:
            return f"{self.uid}!{self._chrom_prefix()}{int(self.rel_pos)}:{ref}>{alt}"
        return f"{self._chrom_prefix()}{int(self.pos)}:{ref}>{alt}"

    def get_abs_base_change(self) -> str:
        return "{}>{}".format(*self._sense())

    def get_base_change(self) -> str:
        return f"{self.ref_base}>{self.alt_base}"

    def __repr__(self) -> str:
        tail = f"{int(self.pos)}:{int(self.rel_pos)}:{self.strand}:{self.ref_base}>{self.alt_base}"
        return self._chrom_prefix() + (tail if self.uid is None else f"{self.uid}!{tail}")

    def set_uid(self, uid):
        if "!" in uid:
            raise ValueError("Cannot use special character `!` in uid.")
        self.uid = uid
        return self

    def set_chrom(self, chrom):
        self.chrom = chrom
        return self

    # ---- identity and order: by string form; edits at different positions order by position
    def __eq__(self, other):
        return repr(self) == repr(other)

    def __hash__(self):
        return hash(repr(self))

    def __lt__(self, other):
        if isinstance(other, Edit) and self.pos != other.pos:
            return self.pos < other.pos
        return repr(self) < str(other)

    def __gt__(self, other):
        if isinstance(other, Edit) and self.pos != other.pos:
            return self.pos > other.pos
        return repr(self) > str(other)


class Allele:
    """A set of edits seen together in one read."""

    edit_type = Edit

    def __init__(self, edits: Optional[Iterable[Edit]] = None):
        self.edits = set() if edits is None else set(edits)
        self.chrom = next(iter(edits)).chrom if edits and len(edits) > 0 else None

    @classmethod
    def from_str(cls, allele_str):
        if type(allele_str) is cls:
            return allele_str
        try:
            return cls({cls.edit_type.from_str(s) for s in allele_str.split(",")})
        except ValueError:
            if allele_str.strip() == "":
                return cls(None)
            raise

    @classmethod
    def match_str(cls, allele_str) -> bool:
        if isinstance(allele_str, cls) or allele_str == "":
            return True
        return all(cls.edit_type.match_str(s) for s in allele_str.split(","))

    def get_range(self):
        """(chrom, lowest, highest position) of the allele's edits."""
        if not self.edits:
            return None
        at = [e.pos for e in self.edits]
        return (self.chrom, min(at), max(at))

    def set_uid(self, uid):
        self.edits = {e.set_uid(uid) for e in self.edits}
        return self

    def get_uid(self):
        if self.edits and all(e.uid is not None for e in self.edits):
            return next(iter(self.edits)).uid
        return None

    def set_chrom(self, chrom: str):
        self.edits = {e.set_chrom(chrom) for e in self.edits}

    @staticmethod
    def _hits(e, ref_base, alt_base, pos, rel_pos) -> bool:
        if e.ref_base != ref_base or e.alt_base != alt_base:
            return False
        if pos is None and rel_pos is None:
            return True
        return (pos is not None and e.pos == pos) or (rel_pos is not None and e.rel_pos == rel_pos)

    @staticmethod
    def _one_of(pos, rel_pos):
        # at most one of the two may be given (neither = any position)
        if pos is not None and rel_pos is not None:
            raise ValueError("Either pos or rel_pos should be specified")

    def has_edit(self, ref_base, alt_base, pos=None, rel_pos=None) -> bool:
        self._one_of(pos, rel_pos)
        return any(self._hits(e, ref_base, alt_base, pos, rel_pos) for e in self.edits)

    def has_other_edit(self, ref_base, alt_base, pos=None, rel_pos=None) -> bool:
        """Whether the allele has an edit of another base change than (ref_base > alt_base) - or, as the reference
        has it, that very edit at the given position."""
        if not self.edits:
            return False
        self._one_of(pos, rel_pos)
        return any(self._hits(e, ref_base, alt_base, pos, rel_pos) or (e.ref_base, e.alt_base) != (ref_base, alt_base)
                   for e in self.edits)

    def get_jaccard(self, other):
        if self.chrom != other.chrom:
            return 0
        return jaccard({str(e) for e in self.edits}, {str(e) for e in other.edits})

    def get_jaccards(self, allele_list: Iterable["Allele"]):
        return np.array([self.get_jaccard(o) for o in allele_list])

    def map_to_closest(self, allele_list: Sequence["Allele"], jaccard_threshold=0.5, merge_priority=None):
        """The allele of ``allele_list`` with the highest Jaccard index to this one when that index exceeds the
        threshold (ties: the highest ``merge_priority``, else the first), otherwise an empty allele."""
        if len(allele_list) == 0:
            return Allele()
        jac = self.get_jaccards(allele_list)
        top = np.nanmax(jac)
        if np.isnan(top):
            return Allele()
        best = np.flatnonzero(jac == top)
        if len(best) > 1 and merge_priority is not None:
            if len(merge_priority) != len(allele_list):
                raise ValueError(f"merge_priority length {len(merge_priority)} is not the same as allele_list length "
                                 f"{len(allele_list)}")
            pick = best[np.nanargmax(merge_priority.iloc[best])]
        else:
            pick = best[0]
        return allele_list[int(pick)] if jac[pick] > jaccard_threshold else Allele()

    def add(self, edit):
        self.edits.add(edit)

    def update(self, edits):
        self.edits.update(edits)

    def __bool__(self):
        return len(self.edits) > 0

    def __len__(self):
        return len(self.edits)

    def __repr__(self):
        return ",".join(str(e) for e in sorted(self.edits))

    def __eq__(self, other):
        return repr(self) == repr(other)

    def __hash__(self):
        return hash(repr(self))

    def __lt__(self, other):  # fewer edits first (what pandas needs to sort a column of alleles)
        return len(self.edits) < len(other.edits)


class MutationType(IntEnum):
    NO_CHANGE = -1
    SYNONYMOUS = 0
    MISSENSE = 1
    NONSENSE = 2


class AminoAcidEdit(Edit):
    """One residue change of a translated allele."""

    def __init__(self, pos, ref: str, alt: str, gene: Optional[str] = None):
        assert ref in AA_SET, f"Invalid ref aa: {ref}"
        assert alt in AA_SET, f"Invalid alt aa: {alt}"
        self.gene, self.pos, self.ref, self.alt = gene, pos, ref, alt

    @classmethod
    def from_str(cls, edit_str):
        fields = edit_str.split(":")
        gene = fields.pop(0) if len(fields) != 2 else None
        pos, change = fields
        ref, alt = change.split(">")
        return cls(pos, ref, alt, gene=gene)  # (pos stays text: see the module docstring)

    @classmethod
    def match_str(cls, edit_str) -> bool:
        if isinstance(edit_str, AminoAcidEdit) or edit_str == "":
            return True
        return bool(re.fullmatch(r"(\w+:)?-?\d+:[+-]:[A-Z*-]>[A-Z*-]", edit_str))

    def _gene_prefix(self) -> str:
        return f"{self.gene}:" if self.gene else ""

    def get_abs_edit(self) -> str:
        return f"{self._gene_prefix()}A{int(self.pos)}:{self.ref}>{self.alt}"

    def __repr__(self) -> str:
        return f"{self._gene_prefix()}{int(self.pos)}:{self.ref}>{self.alt}"

    def __hash__(self):
        return hash(repr(self))

    def _severity(self) -> MutationType:
        if self.ref == self.alt:
            return MutationType.SYNONYMOUS
        if self.alt == "*":
            return MutationType.NONSENSE
        if self.alt in AA_SET:
            return MutationType.MISSENSE
        raise ValueError(f"Alt base invalid:{self.alt}")

    def __eq__(self, other):
        return (self.gene, self.pos, self.ref, self.alt) == (other.gene, other.pos, other.ref, other.alt)

    def _order(self, other, before: bool):
        # edits without a gene first, then by gene, then by position
        mine, theirs = bool(self.gene), bool(other.gene)
        if mine != theirs:
            return theirs if before else mine
        if self.gene == other.gene:
            return self.pos < other.pos if before else self.pos > other.pos
        return (self.gene < other.gene) if before else (self.gene > other.gene)

    def __lt__(self, other):
        return self._order(other, True)

    def __gt__(self, other):
        return self._order(other, False)


class AminoAcidAllele(Allele):
    edit_type = AminoAcidEdit

    def __init__(self, edits: Optional[Iterable[AminoAcidEdit]] = None, gene=None):
        self.gene = None
        self.edits = set() if edits is None else set(edits)

    @classmethod
    def from_str(cls, allele_str):
        if type(allele_str) is cls:
            return allele_str
        edits = set()
        try:
            for s in allele_str.split(","):
                edits.add(AminoAcidEdit.from_str(s))
        except ValueError:
            if allele_str.strip() == "":
                return cls(None)
        return cls(edits)

    @classmethod
    def match_str(cls, allele_str) -> bool:
        if isinstance(allele_str, cls):
            return True
        return all(AminoAcidEdit.match_str(s) for s in allele_str.split(","))

    def _by_severity(self) -> List[Tuple[MutationType, AminoAcidEdit]]:
        return [(e._severity(), e) for e in self.edits]

    def get_most_severe(self):
        return max((s for s, _ in self._by_severity()), default=MutationType.NO_CHANGE)

    def get_most_severe_edit(self):
        ranked = self._by_severity()
        if not ranked:
            return None
        return ranked[int(np.argmax([s for s, _ in ranked]))][1]


class CodingNoncodingAllele(Allele):
    """An allele after translation: its residue changes plus the nucleotide edits outside coding sequence."""

    def __init__(self, aa_edits: Optional[Iterable[AminoAcidEdit]] = None,
                 base_edits: Optional[Iterable[Edit]] = None, unique_identifier=None):
        self.aa_allele = AminoAcidAllele(aa_edits)
        self.nt_allele = Allele(base_edits)
        self.uid = unique_identifier
        if self.uid is not None:
            self.set_uid(self.uid)

    @classmethod
    def from_str(cls, allele_str):
        if isinstance(allele_str, cls):
            return allele_str
        try:
            aa_str, nt_str = allele_str.split("|")
        except ValueError as exc:
            if allele_str.strip() == "":
                return cls(None)
            raise ValueError(f"{allele_str!r} is not an `aa_allele|nt_allele` string") from exc
        nt = Allele.from_str(nt_str)
        return cls(AminoAcidAllele.from_str(aa_str).edits, nt.edits, nt.get_uid())

    @classmethod
    def from_alleles(cls, aa_allele: Optional[AminoAcidAllele] = None, nt_allele: Optional[Allele] = None):
        aa_allele = aa_allele or AminoAcidAllele()
        nt_allele = nt_allele or Allele()
        return cls(aa_allele.edits, nt_allele.edits, nt_allele.get_uid())

    @classmethod
    def match_str(cls, allele_str) -> bool:
        if isinstance(allele_str, CodingNoncodingAllele):
            return True
        if allele_str.count("|") != 1:
            return False
        aa_str, nt_str = allele_str.split("|")
        return bool(AminoAcidAllele.match_str(aa_str) and Allele.match_str(nt_str))

    def get_most_severe(self):
        worst = self.aa_allele.get_most_severe()
        return max(worst, 0.1) if self.nt_allele.edits else worst

    def get_most_severe_edit(self):
        worst = self.aa_allele.get_most_severe()
        if self.nt_allele.edits and worst > 0.1:
            return next(iter(self.nt_allele.edits))
        return self.aa_allele.get_most_severe_edit()

    def set_uid(self, uid):
        self.uid = uid
        self.nt_allele.edits = {e.set_uid(uid) for e in self.nt_allele.edits}

    def has_coding(self) -> bool:
        return len(self.aa_allele.edits) > 0

    def get_jaccard(self, other):
        return (jaccard(self.aa_allele.edits, other.aa_allele.edits),
                jaccard(self.nt_allele.edits, other.nt_allele.edits))

    def get_jaccards(self, allele_list):
        pairs = [self.get_jaccard(o) for o in allele_list]
        return np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])

    def map_to_closest(self, allele_list, aa_jaccard_threshold=0.5, nt_jaccard_threshold=0.5, merge_priority=None):
        """The allele of the list that is closest in BOTH parts (a part this allele does not have matches the list's
        alleles without that part); failing that, closest in the amino-acid part, then in the nucleotide part."""
        if len(allele_list) == 0:
            return CodingNoncodingAllele()
        aa_jac, nt_jac = self.get_jaccards(allele_list)
        has_aa, has_nt = bool(self.aa_allele), bool(self.nt_allele)

        def best(jac, has, part):
            if has:
                return np.flatnonzero(jac == np.nanmax(jac))
            return np.flatnonzero([not getattr(o, part) for o in allele_list])

        def pick(idx):
            if len(idx) > 1 and merge_priority is not None:
                if len(merge_priority) != len(allele_list):
                    raise ValueError("merge_priority length {} is not the same as allele_list length {}".format(
                        len(merge_priority), len(allele_list)))
                return idx[np.argmax(merge_priority[idx])]
            return idx[0] if len(idx) else -1

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # all-NaN slices
            aa_best, nt_best = best(aa_jac, has_aa, "aa_allele"), best(nt_jac, has_nt, "nt_allele")
            both = pick(np.intersect1d(aa_best, nt_best))
            if both >= 0:
                if (has_aa and aa_jac[both] >= aa_jaccard_threshold) or (has_nt and nt_jac[both] >= nt_jaccard_threshold):
                    return allele_list[int(both)]
            elif has_aa:
                i = pick(aa_best)
                if i >= 0:
                    return allele_list[int(i)]
            elif has_nt:
                i = pick(nt_best)
                if i >= 0:
                    return allele_list[int(i)]
        return CodingNoncodingAllele()

    def __bool__(self):
        return bool(self.aa_allele) or bool(self.nt_allele)

    def __len__(self):
        return len(self.aa_allele) + len(self.nt_allele)

    def __repr__(self):
        return f"{self.aa_allele}|{self.nt_allele}"

    def __hash__(self):
        return hash(repr(self))

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return False
        return (self.uid, self.aa_allele, self.nt_allele) == (other.uid, other.aa_allele, other.nt_allele)

    def __lt__(self, other):
        if self.aa_allele < other.aa_allele:
            return True
        return self.aa_allele == other.aa_allele and self.nt_allele < other.nt_allele

    def __gt__(self, other):
        if self.aa_allele > other.aa_allele:
            return True
        return self.aa_allele == other.aa_allele and self.nt_allele > other.nt_allele
