"""Tensor pack consumed by the SVI hot path.

This is the attribute contract that the reference's ``ScreenData`` family
produces (``bean/preprocessing/data_class.py:35-1556``):
the model/guide code only ever reads these attributes, so an object that
carries them is a drop-in ``data`` argument for ``run_inference``.

Shapes: R = n_reps, B = n_condits (sorting bins incl. the control pseudo-bin, or
timepoints), G = n_guides, T = n_targets, A = n_max_alleles, E = n_edits.
dtypes follow the reference (counts f32, size factors / a0 / bounds f64,
``sample_mask`` int64, ``repguide_mask`` bool).

Difference by design: ``allele_to_edit`` is kept in CSR form
(``a2e_ptr``/``a2e_idx``) instead of the reference's dense ``(G, A-1, E)`` 0/1
tensor (``data_class.py:656-699``); ``allele_to_edit_dense()`` materialises the
dense view for small cases.
"""
from __future__ import annotations

import copy
from typing import Optional, Sequence

import numpy as np
import torch

_TENSOR_FIELDS = (
    "X",
    "X_masked",
    "X_bcmatch",
    "X_bcmatch_masked",
    "X_control",
    "X_control_masked",
    "X_bcmatch_control",
    "X_bcmatch_control_masked",
    "sample_mask",
    "control_sample_mask",
    "repguide_mask",
    "size_factor",
    "size_factor_bcmatch",
    "size_factor_control",
    "size_factor_bcmatch_control",
    "a0",
    "a0_bcmatch",
    "pi_a0",
    "allele_counts_control",
    "upper_bounds",
    "lower_bounds",
    "timepoints",
    "control_timepoint",
    "target_lengths",
    "allele_mask",
    "guide_accessibility",
    "a2e_ptr",
    "a2e_idx",
)

# fields indexed by guide on the given axis (used by __getitem__)
_GUIDE_AXIS = {
    "X": 2,
    "X_masked": 2,
    "X_bcmatch": 2,
    "X_bcmatch_masked": 2,
    "X_control": 2,
    "X_control_masked": 2,
    "X_bcmatch_control": 2,
    "X_bcmatch_control_masked": 2,
    "repguide_mask": 1,
    "a0": 0,
    "a0_bcmatch": 0,
    "pi_a0": 0,
    "allele_counts_control": 2,
    "allele_mask": 0,
    "guide_accessibility": 0,
}


class ScreenTensors:
    """Attribute container with the reference's ``ScreenData`` field names."""

    selection: str = "sorting"  # "sorting" | "survival"
    library_design: str = "variant"  # "variant" | "tiling"

    def __init__(self, **fields):
        self.n_edits = 0
        self.n_max_alleles = 2
        self.guide_accessibility = None
        self.a2e_ptr = None
        self.a2e_idx = None
        self.allele_mask = None
        self.timepoints = None
        self.control_timepoint = None
        self.target_names = None
        self.negctrl_guide_idx = None
        for k, v in fields.items():
            setattr(self, k, v)

    # ------------------------------------------------------------------ utils
    def tensor_items(self):
        for k in _TENSOR_FIELDS:
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                yield k, v

    def to(self, device) -> "ScreenTensors":
        out = copy.copy(self)
        for k, v in self.tensor_items():
            setattr(out, k, v.to(device))
        return out

    @property
    def device(self):
        return self.X.device

    @property
    def target_offsets(self) -> torch.Tensor:
        """(T+1,) int64 exclusive prefix sum of ``target_lengths``."""
        tl = self.target_lengths.to(torch.int64)
        off = torch.zeros(tl.numel() + 1, dtype=torch.int64, device=tl.device)
        off[1:] = torch.cumsum(tl, 0)
        return off

    @property
    def guide_to_target(self) -> torch.Tensor:
        """(G,) int32 target index of every guide (guides are target-sorted)."""
        tl = self.target_lengths.to(torch.int64)
        return torch.repeat_interleave(
            torch.arange(tl.numel(), device=tl.device), tl
        ).to(torch.int32)

    def allele_to_edit_dense(self) -> torch.Tensor:
        """Dense ``(G, A-1, E)`` f32 view of the CSR allele->edit map."""
        G, A, E = self.n_guides, self.n_max_alleles, self.n_edits
        dense = torch.zeros((G * (A - 1), E), dtype=torch.float32)
        ptr = self.a2e_ptr.cpu().numpy()
        idx = self.a2e_idx.cpu().numpy()
        rows = np.repeat(np.arange(G * (A - 1)), np.diff(ptr))
        dense[torch.as_tensor(rows), torch.as_tensor(idx.astype(np.int64))] = 1.0
        return dense.reshape(G, A - 1, E).to(self.X.device)

    # ---------------------------------------------- neg-ctrl / shard sub-setting
    def __getitem__(self, guide_idx: Sequence[int]) -> "ScreenTensors":
        """Guide subset, as ``ScreenData.__getitem__`` (data_class.py:207-218,
        399-414, 503-509): size factors, bounds and masks per sample are kept,
        per-guide tensors are sliced, ``target_lengths`` is recomputed from the
        kept guides (which must stay target-sorted)."""
        idx = torch.as_tensor(np.asarray(guide_idx), dtype=torch.int64)
        out = copy.copy(self)
        for k, ax in _GUIDE_AXIS.items():
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                setattr(out, k, v.index_select(ax, idx.to(v.device)))
        out.n_guides = int(idx.numel())
        if self.negctrl_guide_idx is not None:  # positions of the kept negative-control guides
            neg = np.zeros(self.n_guides, dtype=bool)
            neg[np.asarray(self.negctrl_guide_idx, dtype=np.int64)] = True
            out.negctrl_guide_idx = np.nonzero(neg[idx.numpy()])[0]
        if getattr(self, "target_lengths", None) is not None:
            g2t = self.guide_to_target.cpu()[idx]
            if idx.numel():
                change = torch.ones_like(g2t, dtype=torch.bool)
                change[1:] = g2t[1:] != g2t[:-1]
                starts = torch.nonzero(change).flatten()
                ends = torch.cat([starts[1:], torch.tensor([idx.numel()])])
                lengths = ends - starts
                if torch.unique(g2t).numel() != lengths.numel():
                    raise ValueError(
                        "Input Screen object not sorted for target identity."
                    )
            else:
                lengths = torch.zeros(0, dtype=torch.int64)
            out.target_lengths = lengths.to(self.target_lengths.device)
            out.n_targets = int(lengths.numel())
            if self.target_names is not None:
                out.target_names = [self.target_names[int(t)] for t in g2t[starts]]
        if self.a2e_ptr is not None:
            A1 = self.n_max_alleles - 1
            ptr = self.a2e_ptr.cpu().numpy().astype(np.int64)
            col = self.a2e_idx.cpu().numpy()
            rows = (idx.numpy()[:, None] * A1 + np.arange(A1)[None, :]).reshape(-1)
            counts = ptr[rows + 1] - ptr[rows]
            new_ptr = np.zeros(rows.size + 1, dtype=np.int64)
            np.cumsum(counts, out=new_ptr[1:])
            new_idx = np.concatenate(
                [col[ptr[r] : ptr[r + 1]] for r in rows] or [np.zeros(0, col.dtype)]
            )
            out.a2e_ptr = torch.as_tensor(new_ptr.astype(np.int32))
            out.a2e_idx = torch.as_tensor(new_idx.astype(np.int32))
        return out

    def validate(self) -> None:
        """Host-side shape/dtype checks run before any kernel launch."""
        R, B, G = self.n_reps, self.n_condits, self.n_guides
        assert tuple(self.X.shape) == (R, B, G), (self.X.shape, (R, B, G))
        assert tuple(self.X_masked.shape) == (R, B, G)
        assert tuple(self.sample_mask.shape) == (R, B)
        assert tuple(self.repguide_mask.shape) == (R, G)
        assert tuple(self.size_factor.shape) == (R, B)
        assert tuple(self.a0.shape) == (G,)
        if getattr(self, "a2e_ptr", None) is not None:
            A1 = self.n_max_alleles - 1
            assert self.a2e_ptr.numel() == G * A1 + 1
            assert int(self.a2e_ptr[-1]) == self.a2e_idx.numel()
            assert self.a2e_idx.numel() == 0 or int(self.a2e_idx.max()) < self.n_edits
            assert tuple(self.allele_mask.shape) == (G, self.n_max_alleles)
        if getattr(self, "target_lengths", None) is not None:
            assert int(self.target_lengths.sum()) == G, "target_lengths must sum to G"
            assert self.target_lengths.numel() == self.n_targets
        if getattr(self, "X_bcmatch", None) is not None:
            assert tuple(self.X_bcmatch.shape) == (R, B, G)
            assert tuple(self.size_factor_bcmatch.shape) == (R, B)
            assert tuple(self.a0_bcmatch.shape) == (G,)
        if self.selection == "survival":
            assert tuple(self.timepoints.shape) == (B,)
        if getattr(self, "allele_counts_control", None) is not None:
            acc = self.allele_counts_control
            assert acc.shape[0] == R and acc.shape[2] == G
            assert acc.shape[3] == self.n_max_alleles
            assert tuple(self.pi_a0.shape) == (G,)
