"""Sorting-screen model / guide descriptors.

In the reference these names are Pyro programs (``bean/model/model.py``) that
``pyro.infer.SVI`` traces every step.  Here the per-step work is fixed-function
HIP, so each name is a *descriptor*: calling it (directly or through the
``functools.partial`` that ``identify_model_guide`` builds) returns a
``ModelSpec`` recording the family and the same keyword arguments the reference
accepts; ``run_inference`` turns the (model, guide) pair into a ``HipSVI``
engine.  Argument names, defaults and truthiness quirks follow the reference so
that call sites read the same.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional


@dataclass
class ModelSpec:
    family: str          # "Normal" | "ControlNormal" | "MixtureNormal" | "MultiMixtureNormal"
    role: str            # "model" | "guide"
    selection: str = "sorting"
    kwargs: Dict[str, Any] = field(default_factory=dict)

    def get(self, key, default=None):
        return self.kwargs.get(key, default)


def _spec(family, role, **kw):
    kw.pop("data", None)
    return ModelSpec(family=family, role=role, selection="sorting", kwargs=kw)


def NormalModel(data=None, mask_thres: int = 10, use_bcmatch: bool = True, sd_scale: float = 0.01,
                prior_params: Optional[dict] = None):
    """bean/model/model.py:19-165."""
    return _spec("Normal", "model", mask_thres=mask_thres, use_bcmatch=use_bcmatch, sd_scale=sd_scale,
                 prior_params=prior_params)


def ControlNormalModel(data=None, mask_thres=10, use_bcmatch=True):
    """bean/model/model.py:168-252."""
    return _spec("ControlNormal", "model", mask_thres=mask_thres, use_bcmatch=use_bcmatch)


def MixtureNormalConstPiModel(data=None, alpha_prior: float = 1, use_bcmatch: bool = True, sd_scale: float = 0.01):
    """bean/model/model.py:255-375.  Dead path in the reference (SURVEY.md
    Appendix C item 13: its data class carries no ``pi``); not implemented."""
    raise NotImplementedError(
        "MixtureNormalConstPi (--const-pi) is unreachable in the reference "
        "(VariantSortingScreenData has no `pi`) and is not implemented"
    )


def MixtureNormalModel(data=None, alpha_prior: float = 1, use_bcmatch: bool = True, sd_scale: float = 0.01,
                       scale_by_accessibility: bool = False, fit_noise: bool = False,
                       prior_params: Optional[dict] = None):
    """bean/model/model.py:378-547."""
    return _spec("MixtureNormal", "model", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch, sd_scale=sd_scale,
                 scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise, prior_params=prior_params)


def MultiMixtureNormalModel(data=None, alpha_prior=1, use_bcmatch=True, sd_scale=0.01,
                            scale_by_accessibility=False, fit_noise: bool = False,
                            prior_params: Optional[dict] = None, epsilon=1e-5):
    """bean/model/model.py:550-751."""
    return _spec("MultiMixtureNormal", "model", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 sd_scale=sd_scale, scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise,
                 prior_params=prior_params, epsilon=epsilon)


def NormalGuide(data=None):
    """bean/model/model.py:754-782."""
    return _spec("Normal", "guide")


def MixtureNormalGuide(data=None, alpha_prior: float = 1, use_bcmatch: bool = True,
                       scale_by_accessibility: bool = False, fit_noise: bool = False):
    """bean/model/model.py:785-858."""
    return _spec("MixtureNormal", "guide", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise)


def ControlNormalGuide(data=None, mask_thres=10, use_bcmatch=True):
    """bean/model/model.py:861-875."""
    return _spec("ControlNormal", "guide", mask_thres=mask_thres, use_bcmatch=use_bcmatch)


def MultiMixtureNormalGuide(data=None, alpha_prior=1, use_bcmatch=True, epsilon=1e-5,
                            scale_by_accessibility: bool = False, fit_noise: bool = False):
    """bean/model/model.py:878-962."""
    return _spec("MultiMixtureNormal", "guide", alpha_prior=alpha_prior, use_bcmatch=use_bcmatch,
                 epsilon=epsilon, scale_by_accessibility=scale_by_accessibility, fit_noise=fit_noise)
