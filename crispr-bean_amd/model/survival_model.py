"""Survival-screen model / guide descriptors (see ``model.py`` for the idea).

Names, arguments and defaults follow ``bean/model/survival_model.py``.  The
HIP engine implements all of them: ``Normal`` (``--uniform-edit``), ``ControlNormal``,
``MixtureNormal`` (+Acc) and the tiling ``MultiMixtureNormal`` (+Acc).
"""
from __future__ import annotations

from typing import Optional

from .model import ModelSpec


def _spec(family, role, **kw):
    kw.pop("data", None)
    return ModelSpec(family=family, role=role, selection="survival", kwargs=kw)


def NormalModel(data=None, mask_thres: int = 10, use_bcmatch: bool = True,
                prior_params: Optional[dict] = None, mu_negctrl: float = 0.0):
    """bean/model/survival_model.py:15-130."""
    return _spec("Normal", "model", mask_thres=mask_thres, use_bcmatch=use_bcmatch,
                 prior_params=prior_params, mu_negctrl=mu_negctrl)


def ControlNormalModel(data=None, mask_thres=10, use_bcmatch=True):
    """bean/model/survival_model.py:133-212."""
    return _spec("ControlNormal", "model", mask_thres=mask_thres, use_bcmatch=use_bcmatch)


def MixtureNormalModel(data=None, alpha_prior: float = 1, use_bcmatch: bool = True,
                       use_all_timepoints_for_pi: bool = True, sd_scale: float = 0.01,
                       scale_by_accessibility: bool = False, fit_noise: bool = False,
                       mask_thres: int = 10, prior_params: Optional[dict] = None,
                       mu_negctrl=(0.0, 0.1)):
    """bean/model/survival_model.py:215-424."""
    return _spec("MixtureNormal", "model", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 sd_scale=sd_scale, scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise,
                 mask_thres=mask_thres, prior_params=prior_params, mu_negctrl=mu_negctrl)


def MultiMixtureNormalModel(data=None, alpha_prior=1, use_bcmatch=True, use_all_timepoints_for_pi: bool = True,
                            sd_scale=0.01, norm_pi=False, scale_by_accessibility=False, fit_noise: bool = False,
                            prior_params: Optional[dict] = None, epsilon=1e-5, mu_negctrl=(0.0, 0.1)):
    """bean/model/survival_model.py:427-626."""
    return _spec("MultiMixtureNormal", "model", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise,
                 prior_params=prior_params, epsilon=epsilon, mu_negctrl=mu_negctrl)


def NormalGuide(data=None):
    """bean/model/survival_model.py:629-648."""
    return _spec("Normal", "guide")


def MixtureNormalGuide(data=None, alpha_prior: float = 1, use_bcmatch: bool = True,
                       scale_by_accessibility: bool = False, fit_noise: bool = False):
    """bean/model/survival_model.py:651-739."""
    return _spec("MixtureNormal", "guide", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise)


def ControlNormalGuide(data=None, mask_thres=10, use_bcmatch=True):
    """bean/model/survival_model.py:742-756."""
    return _spec("ControlNormal", "guide", mask_thres=mask_thres, use_bcmatch=use_bcmatch)


def MultiMixtureNormalGuide(data=None, alpha_prior=1, use_bcmatch=True, epsilon=1e-5,
                            scale_by_accessibility: bool = False, fit_noise: bool = False):
    """bean/model/survival_model.py:759-833."""
    return _spec("MultiMixtureNormal", "guide", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 epsilon=epsilon, scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise)
