"""ReporterScreen -> ``ScreenTensors``: what the ``ScreenData`` family of
``bean/preprocessing/data_class.py`` computes once per screen for the variant
library design (``VariantSorting[Reporter]ScreenData``,
``VariantSurvival[Reporter]ScreenData``): sample ordering (replicate, condition),
bin edges / timepoints, masks, size factors, the trend-fitted Dirichlet-Multinomial
precisions ``a0`` / ``a0_bcmatch`` / ``pi_a0`` and control allele counts.
Tiling screens (``TilingSortingReporterScreenData``): the allele tables are turned into the CSR
``allele_to_edit`` map, allele count tensors and the allele mask by ``alleles.py``.

Reference quirks kept: the control condition always stays among the selected
samples (``bean/cli/run.py:114``); size factors are normalised column
means (``data_class.py:237-251``); the samples' ``replicate`` column is used
whatever ``--replicate-col`` says after ``prepare_bdata`` rewrote it.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch

from .alleles import allele_count_tensor, build_allele_tensors
from .alpha0 import fitted_alpha0, fitted_pi_alpha0, pred_alpha0, pred_pi_alpha0
from .data_class import ScreenTensors
from .utils import assign_rep_ids_and_sort, get_accessibility_guides


def _size_factor(X: np.ndarray) -> np.ndarray:
    sf = X.mean(axis=0)
    return sf / sf.mean()


def _target_lengths(guides: pd.DataFrame, target_col: str) -> np.ndarray:
    codes = guides[target_col].astype("category").cat.codes.values
    change = np.ones(len(codes), dtype=bool)
    change[1:] = codes[1:] != codes[:-1]
    starts = np.nonzero(change)[0]
    lengths = np.diff(np.append(starts, len(codes)))
    if len(lengths) != len(np.unique(codes)):
        raise ValueError(
            "Input Screen object not sorted for target identity. Sort the screen object so that guides targeting "
            f"the same object would occur as consecutive block by screen[screen.guides[{target_col}].argsort(),:]")
    return lengths


def build_variant_screen_data(
    screen,
    selection: str = "sorting",
    reporter: bool = True,
    *,
    repguide_mask: Optional[str] = None,
    sample_mask_column: Optional[str] = "mask",
    shrink_alpha: bool = False,
    condition_column: str = "condition",
    control_condition: str = "bulk",
    accessibility_col: Optional[str] = None,
    accessibility_bw_path: Optional[str] = None,
    popt: Optional[Tuple[float, float]] = None,
    pi_popt: Optional[Tuple[float, float]] = None,
    lower_quantile_column: str = "lower_quantile",
    upper_quantile_column: str = "upper_quantile",
    time_column: str = "time",
    target_col: str = "target",
    use_bcmatch: bool = True,
    negctrl_guide_idx: Optional[Sequence[int]] = None,
    library_design: str = "variant",
    allele_df_key: Optional[str] = None,
    allele_col: Optional[str] = None,
    control_guide_tag: Optional[str] = None,
    **_ignored,
) -> ScreenTensors:
    replicate_column = "replicate"
    screen = screen.copy()
    samples = screen.samples
    samples["size_factor"] = _size_factor(screen.X)
    # ScreenData.__init__ (data_class.py:75-92): with uns["sample_covariates"] a "replicate" is a
    # (replicate, covariates) combination, keyed by the joined string `_rc`
    sample_covariates = None
    if "sample_covariates" in screen.uns:
        sample_covariates = [str(c) for c in np.asarray(screen.uns["sample_covariates"]).reshape(-1)]
        missing = [c for c in sample_covariates if c not in samples.columns]
        if missing:
            raise ValueError(f"uns['sample_covariates'] names {missing}, which are not columns of screen.samples")
        samples["_rc"] = [".".join(str(v) for v in row)
                          for row in samples[[replicate_column] + sample_covariates].values.tolist()]
        replicate_column = "_rc"
    if reporter or use_bcmatch:
        if "X_bcmatch" not in screen.layers:
            raise ValueError("the screen has no X_bcmatch layer")
        samples["size_factor_bcmatch"] = _size_factor(screen.layers["X_bcmatch"])
    control = control_condition.split(",")
    if samples[condition_column].isnull().any():
        screen = screen[:, (~samples[condition_column].isnull()).values]
        samples = screen.samples
    cond = samples[condition_column].astype(str)

    # ---- condition ids and sample order
    if selection == "sorting":
        for col in (lower_quantile_column, upper_quantile_column):
            if ((samples[col] < 0.0) | (samples[col] > 1.0)).any():
                raise ValueError(f"Invalid quantile value({samples[col]}) in screen.samples[{col}]: check input.")
        if (samples[upper_quantile_column] - samples[lower_quantile_column] < 0).any():
            raise ValueError(f"Not all screen.samples[{upper_quantile_column}] larger than "
                             f"screen.samples[{lower_quantile_column}]: check input.")
        n_reps = samples[replicate_column].astype(str).nunique()
        if not (samples.groupby([upper_quantile_column, lower_quantile_column]).size() == n_reps).all():
            raise ValueError("Not all replicate share same quantile bin definition. If you have missing bin data, "
                             "add the sample and add 'mask' column in 'screen.samples' or run `bean-qc` that "
                             "automatically handles this.")
        bins = samples.sort_values([upper_quantile_column, lower_quantile_column])[
            [upper_quantile_column, lower_quantile_column]].drop_duplicates()
        ids = {(u, l): j for j, (u, l) in enumerate(zip(bins[upper_quantile_column], bins[lower_quantile_column]))}
        id_col = f"{condition_column}_id"
        samples[id_col] = [ids[(u, l)] for u, l in zip(samples[upper_quantile_column], samples[lower_quantile_column])]
        cond_values = None
        upper = bins[upper_quantile_column].values.astype(np.float64)
        lower = bins[lower_quantile_column].values.astype(np.float64)
    else:
        try:
            times = samples[time_column].astype(float)
        except ValueError as exc:
            raise ValueError(f"Invalid timepoint value({samples[time_column]}) in screen.samples[{time_column}]: "
                             "check input.") from exc
        samples[time_column] = times / times.max()
        n_reps = samples[replicate_column].astype(str).nunique()
        if not (samples.groupby(condition_column, observed=True).size() == n_reps).all():
            raise ValueError("Not all replicate share same timepoint definition. If you have missing bin data, add "
                             "the sample and add 'mask' column in 'screen.samples', or run `bean-qc`.")
        uniq = np.sort(samples[time_column].unique())
        if np.isnan(uniq).any():
            raise ValueError(f"NaN values in time points provided in input: {samples[time_column]}")
        id_col = f"{time_column}_id"
        samples[id_col] = samples[time_column].map({t: j for j, t in enumerate(uniq)})
        cond_values = uniq
    screen = assign_rep_ids_and_sort(screen, replicate_column, id_col)
    samples = screen.samples
    cond = samples[condition_column].astype(str)
    is_ctrl = cond.isin(control).values
    if not is_ctrl.any():
        raise ValueError(f"No sample has control label `{control_condition}`")

    R = samples[replicate_column].astype(str).nunique()
    B = samples[id_col].nunique()
    G = screen.n_obs
    C = len(control)
    if screen.n_vars != R * B:
        raise ValueError(f"expected {R} replicates x {B} conditions = {R * B} samples, found {screen.n_vars}")

    def rbg(mat, n_cond):  # (G, R * n_cond) -> (R, n_cond, G), as ScreenData.transform_data
        return torch.as_tensor(np.ascontiguousarray(mat)).T.reshape(R, n_cond, G).float()

    ctrl = screen[:, is_ctrl]
    if sample_mask_column is not None:
        sample_mask = torch.as_tensor(samples[sample_mask_column].to_numpy()).reshape(R, B)
        control_sample_mask = torch.as_tensor(ctrl.samples[sample_mask_column].to_numpy()).reshape(R, C)
    else:
        sample_mask = torch.ones((R, B), dtype=torch.bool)
        control_sample_mask = torch.ones((R, C), dtype=torch.bool)
    X = rbg(screen.X, B)
    X_control = rbg(ctrl.X, C)
    no_zero = ~(X == 0).any(dim=1)
    if repguide_mask is None:
        rg = no_zero
    else:
        assert repguide_mask in screen.uns, f"{repguide_mask} not in screen.uns"
        tbl = screen.uns[repguide_mask]
        assert tbl.shape == (G, R), (tbl.shape, (G, R))
        tbl = tbl.reindex(screen.guides.index)
        rg = torch.logical_and(torch.as_tensor(tbl.values.T.astype(float)) > 0, no_zero)
    sf = torch.as_tensor(samples["size_factor"].to_numpy()).reshape(R, B)
    sf_control = torch.as_tensor(ctrl.samples["size_factor"].to_numpy()).reshape(R, C)
    a0, popt_fit = fitted_alpha0(X, sf, sample_mask, shrink=shrink_alpha, popt=popt)

    data = ScreenTensors(
        n_reps=R, n_condits=B, n_guides=G, n_max_alleles=2,
        X=X, X_masked=X * sample_mask[:, :, None], X_control=X_control,
        X_control_masked=X_control * control_sample_mask[:, :, None],
        sample_mask=sample_mask, control_sample_mask=control_sample_mask, repguide_mask=rg,
        size_factor=sf, size_factor_control=sf_control, a0=torch.as_tensor(a0), popt=popt_fit,
    )
    tiling = library_design == "tiling"
    data.selection, data.library_design = selection, library_design
    data.screen = screen
    data.screen_control = ctrl
    data.condition_column, data.control_condition, data.target_col = condition_column, control, target_col
    if selection == "sorting":
        data.upper_bounds = torch.as_tensor(upper)
        data.lower_bounds = torch.as_tensor(lower)
        data.n_bins = B
    else:
        data.upper_bounds = data.lower_bounds = None
        data.timepoints = torch.as_tensor(np.asarray(cond_values, dtype=np.float64))
        ct = np.unique(ctrl.samples[time_column].values.astype(np.float64))
        if len(ct) != C:
            raise ValueError("All samples with --control-condition should have the same --time-col column in "
                             "ReporterScreen.samples[time_col]. Check your input ReporterScreen object.")
        data.control_timepoint = torch.as_tensor(ct)
        data.n_timepoints = B
    if not tiling:
        lengths = _target_lengths(screen.guides, target_col)
        data.target_lengths = torch.as_tensor(lengths)
        data.n_targets = int(len(lengths))
        data.target_names = list(pd.unique(screen.guides[target_col].astype(str)))
    else:
        data.target_lengths = None
    data.negctrl_guide_idx = negctrl_guide_idx
    if sample_covariates is not None:
        # (R, n_cov) integer design, rows in replicate-id order (data_class.py:972-979)
        data.sample_covariates = sample_covariates
        data.n_sample_covariates = len(sample_covariates)
        design = samples[["_rc"] + sample_covariates].drop_duplicates().set_index("_rc")
        data.rep_by_cov = torch.as_tensor(design.values.astype(int))
        if data.rep_by_cov.shape[0] != R:
            raise ValueError("sample covariates are not constant within a replicate")
    data.guide_accessibility = None
    if accessibility_col is not None:
        acc = screen.guides[accessibility_col].values.astype(np.float64)
        data.guide_accessibility = torch.as_tensor(acc)
    elif accessibility_bw_path is not None:  # ScreenData._post_init (data_class.py:126-133)
        data.guide_accessibility = get_accessibility_guides(accessibility_bw_path, screen.guides)

    if reporter or use_bcmatch:
        Xb = rbg(screen.layers["X_bcmatch"], B)
        Xb_control = rbg(ctrl.layers["X_bcmatch"], C)
        sf_b = torch.as_tensor(samples["size_factor_bcmatch"].to_numpy()).reshape(R, B)
        data.X_bcmatch, data.X_bcmatch_masked = Xb, Xb * sample_mask[:, :, None]
        data.X_bcmatch_control = Xb_control
        data.X_bcmatch_control_masked = Xb_control * control_sample_mask[:, :, None]
        data.size_factor_bcmatch = sf_b
        data.size_factor_bcmatch_control = torch.as_tensor(
            ctrl.samples["size_factor_bcmatch"].to_numpy()).reshape(R, C)
        data.a0_bcmatch = torch.as_tensor(pred_alpha0(Xb, sf_b, popt_fit, sample_mask))
    if tiling:
        # TilingReporterScreenData._post_init (data_class.py:574-654)
        if allele_df_key is None or allele_df_key not in screen.uns:
            raise ValueError(f"--allele-df-key `{allele_df_key}` is not in the screen's uns: {list(screen.uns)}")
        at = build_allele_tensors(screen.uns[allele_df_key], list(screen.guides.index), allele_col=allele_col,
                                  control_guide_tag=control_guide_tag)
        data.n_max_alleles = at.n_max_alleles
        data.edit_index = at.edit_index
        data.n_edits = len(at.edit_index)
        data.n_targets = data.n_edits
        data.a2e_ptr, data.a2e_idx, data.allele_mask = at.a2e_ptr, at.a2e_idx, at.allele_mask
        data.allele_counts = allele_count_tensor(at, samples, list(screen.guides.index),
                                                 np.asarray(screen.layers["X_bcmatch"]), R, id_col)
        data.allele_counts_control = allele_count_tensor(at, ctrl.samples, list(screen.guides.index),
                                                         np.asarray(ctrl.layers["X_bcmatch"]), R, id_col)
        data.n_alleles_dropped = at.n_dropped
    if reporter:
        if not tiling:
            if "edits" not in screen.layers:
                raise ValueError("the screen has no `edits` layer (reporter editing outcomes)")
            edited = rbg(ctrl.layers["edits"], C)
            nonedited = (data.X_bcmatch_control - edited).clamp(min=0)
            data.allele_counts_control = torch.stack([nonedited, edited], dim=-1)
        if pi_popt is not None:
            pi_a0 = pred_pi_alpha0(data.allele_counts_control, sf_control, pi_popt)
        else:
            pi_a0, data._pi_popt = fitted_pi_alpha0(data.allele_counts_control, sf_control, shrink=shrink_alpha)
        data.pi_a0 = torch.as_tensor(pi_a0)
    data.validate()
    return data


def _builder(selection, reporter, library_design="variant"):
    def make(screen, **kw):
        return build_variant_screen_data(screen, selection=selection, reporter=reporter,
                                         library_design=library_design, **kw)
    return make


# model label -> tensor builder (``DATACLASS_DICT`` of bean/preprocessing/data_class.py:1540-1556)
DATACLASS_DICT = {
    "sorting": {
        "Normal": _builder("sorting", False),
        "MixtureNormal": _builder("sorting", True),
        "_MixtureNormal": _builder("sorting", True),
        "MixtureNormal+Acc": _builder("sorting", True),
        "_MixtureNormal+Acc": _builder("sorting", True),
        "MixtureNormalConstPi": _builder("sorting", False),
        "MultiMixtureNormal": _builder("sorting", True, "tiling"),
        "MultiMixtureNormal+Acc": _builder("sorting", True, "tiling"),
    },
    "survival": {
        "Normal": _builder("survival", False),
        "MixtureNormal": _builder("survival", True),
        "_MixtureNormal": _builder("survival", True),
        "MixtureNormal+Acc": _builder("survival", True),
        "_MixtureNormal+Acc": _builder("survival", True),
        "MultiMixtureNormal": _builder("survival", True, "tiling"),
        "MultiMixtureNormal+Acc": _builder("survival", True, "tiling"),
    },
}
