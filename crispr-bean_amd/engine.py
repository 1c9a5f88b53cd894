"""Host-side driver of the HIP SVI step.

Owns the torch tensors (parameters, ClippedAdam moments, gradients, loss
history) whose device pointers are bound into a ``libbean_hip`` handle, and
exposes the three operations the reference gets from Pyro
(``bean/model/run.py:366-396``): one ELBO+gradient evaluation, one ClippedAdam
update, and the fused ``for t in range(num_steps): svi.step(data)`` loop.

PyTorch is used for device memory and streams only; all arithmetic of the step
happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

PI_NOISE_SD = 0.655  # bean/model/utils.py:133
PER_GUIDE = ("alpha_pi", "noise_loc", "noise_scale", "q0", "initial_abundance")  # parameters with a guide axis
POSITIVE = ("mu_scale", "sd_scale", "alpha_pi", "noise_scale", "q0", "initial_abundance", "mu_cov_scale")


def _quantile_edges(upper: torch.Tensor, lower: torch.Tensor):
    """z-scores of the bin edges; quantiles exactly 1.0 / 0.0 are open edges
    (``bean/model/utils.py:48-54``)."""
    std = torch.distributions.Normal(0.0, 1.0)
    uq, lq = upper.detach().cpu().double(), lower.detach().cpu().double()
    z_hi = torch.full_like(uq, float("inf"))
    z_lo = torch.full_like(lq, float("-inf"))
    z_hi[uq != 1.0] = std.icdf(uq[uq != 1.0])
    z_lo[lq != 0.0] = std.icdf(lq[lq != 0.0])
    return z_hi, z_lo


class HipSVI:
    """One model family bound to one screen on one GPU."""

    def __init__(
        self,
        family: str,
        data,
        *,
        use_bcmatch: bool = True,
        scale_by_accessibility: bool = False,
        fit_noise: bool = True,
        sd_scale: float = 0.01,
        prior_params: Optional[dict] = None,
        initial_lr: float = 0.01,
        gamma: float = 0.1,
        num_steps: int = 2000,
        loss_capacity: Optional[int] = None,
        mask_thres: int = 10,
        dump_noise: bool = False,
        device=None,
        guide_offset: int = 0,
        target_offset: int = 0,
        n_guides_total: int = 0,
        mu_negctrl=(0.0, 0.1),
        t0_totals: Optional[torch.Tensor] = None,
        loss_owner: bool = True,
        alpha_prior: float = 1.0,
        lib_variant: Optional[str] = None,
        guide_ids: Optional[torch.Tensor] = None,
    ):
        if family not in _lib.FAMILY:
            raise ValueError(f"unknown model family {family!r}")
        survival = getattr(data, "selection", "sorting") == "survival"
        surv_normal = survival and family == "Normal"
        self.surv_normal = surv_normal
        self.survival = survival
        if not torch.cuda.is_available():
            raise RuntimeError("crispr-bean_amd needs a ROCm GPU: there is no CPU fallback")
        # the register-resident tiling kernels hold 8 alleles per guide (and 8 conditions) in the default
        # build, 16 in the second one, 32 alleles (16 conditions) in the third; more alleles per guide run in
        # the allele-parallel kernels of any build (csrc/bean_tiling_wide.hpp), so only the condition count
        # decides then
        amax = 8
        n_al = int(getattr(data, "n_max_alleles", 2)) if family == "MultiMixtureNormal" else 2
        if (8 < n_al <= 16) or data.n_condits > 8 or n_al > 256:  # (> 256: the wider allele-parallel kernels)
            amax = 16
        if 16 < n_al <= 32:
            amax = 32
        if lib_variant is not None:
            # "ab": libbean_hip_ab.so, the default kernels plus the superseded / opt-in forms the BEAN_HIP_*
            # switches select (A/B measurements, bit-identity tests); 8 alleles / conditions only
            if lib_variant != _lib.AB or amax != 8:
                raise ValueError("lib_variant must be 'ab' (and the screen must fit the 8-allele / 8-condition build)")
            amax = _lib.AB
        self.lib = _lib.load(amax)
        self.device = torch.device(device if device is not None else "cuda:0")
        self.family = family
        self.data = data
        data.validate()
        dev = self.device
        R, B, G = data.n_reps, data.n_condits, data.n_guides
        tiling = family == "MultiMixtureNormal"
        mixture = family in ("MixtureNormal", "MultiMixtureNormal")  # families with a Dirichlet pi site
        acc = bool(scale_by_accessibility) and mixture
        self.scale_by_accessibility = acc
        # tiling: fit_noise=~args.dont_fit_noise is always truthy (bean/model/run.py:416)
        self.fit_noise = (bool(fit_noise) or tiling) and acc
        has_bc = getattr(data, "X_bcmatch_masked", None) is not None
        self.use_bcmatch = bool(use_bcmatch) and has_bc
        if mixture and not has_bc:
            raise ValueError("MixtureNormal needs barcode-matched counts (X_bcmatch)")
        T = 1 if family == "ControlNormal" else (data.n_edits if tiling else data.n_targets)
        self.T = T
        A = int(data.n_max_alleles) if tiling else 2
        self.A = A
        nnz = 0
        if tiling:
            toff = g2t = None
            max_len = 0
            a2e_ptr = data.a2e_ptr.cpu().to(torch.int32)
            a2e_idx = data.a2e_idx.cpu().to(torch.int32)
            nnz = int(a2e_idx.numel())
            # transpose: edit -> allele slots containing it (stable, so sums keep a fixed order)
            rows = torch.repeat_interleave(
                torch.arange(a2e_ptr.numel() - 1, dtype=torch.int64), (a2e_ptr[1:] - a2e_ptr[:-1]).to(torch.int64))
            order = torch.argsort(a2e_idx.to(torch.int64), stable=True)
            e2a_idx = rows[order].to(torch.int32)
            counts = torch.bincount(a2e_idx.to(torch.int64), minlength=T)
            e2a_ptr = torch.zeros(T + 1, dtype=torch.int32)
            e2a_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        elif family == "ControlNormal":
            toff = torch.tensor([0, G], dtype=torch.int32)
            g2t = torch.zeros(G, dtype=torch.int32)
            max_len = G
        else:
            off64 = data.target_offsets.cpu()
            toff = off64.to(torch.int32)
            g2t = data.guide_to_target.cpu()
            max_len = int(data.target_lengths.max())
        flags = 0
        if self.use_bcmatch:
            flags |= _lib.FLAG_USE_BCMATCH
        if acc:
            flags |= _lib.FLAG_SCALE_BY_ACC
        if self.fit_noise:
            flags |= _lib.FLAG_FIT_NOISE
        if dump_noise:
            flags |= _lib.FLAG_DUMP_PI
        if not loss_owner:
            flags |= _lib.FLAG_NOT_LOSS_OWNER
        # sample covariates (uns["sample_covariates"], data_class.py:75-92): only the sorting NormalModel
        # has the mu_cov site (model.py:73-91, 771-783); with any other family the reference fits with
        # the regrouped replicates and then fails in write_result_table (readwrite.py:88-90)
        covs = getattr(data, "sample_covariates", None)
        self.n_cov = 0
        if covs is not None and family == "ControlNormal":
            # the reference's ControlNormalModel / Guide have no mu_cov site (model.py:168-252, 861-875): the
            # negative-control fit of a screen with sample covariates just sees the regrouped
            # (replicate, covariate) replicates
            covs = None
        if covs is not None:
            if not (family == "Normal" and not survival):
                raise ValueError(
                    "this screen carries uns['sample_covariates']: only the sorting Normal model (--uniform-edit) "
                    f"models them (bean/model/model.py:73-91); the {family} family cannot write its result table "
                    "in the reference (readwrite.py:88-90) and is refused here")
            self.n_cov = int(data.n_sample_covariates)
        self.prior_params = prior_params
        if prior_params is not None and ("mu_loc" in prior_params or "mu_scale" in prior_params):
            flags |= _lib.FLAG_PRIOR_NORMAL_MU
        # survival NormalModel: prior_params["initial_abundance"] replaces the ones / G prior concentration of
        # the Dirichlet-over-guides site (survival_model.py:38-49); float32 values as the reference holds them
        prior_ia, prior_ia_total = None, 0.0
        if surv_normal and prior_params is not None and "initial_abundance" in prior_params:
            full = torch.as_tensor(prior_params["initial_abundance"]).detach().cpu().to(torch.float32).double().reshape(-1)
            g_all = int(n_guides_total) if n_guides_total else G
            if full.numel() != g_all:
                raise ValueError(f"prior_params['initial_abundance'] has {full.numel()} entries for {g_all} guides")
            if not bool((full > 0).all()):
                raise ValueError("prior_params['initial_abundance'] must be positive")
            prior_ia = full[int(guide_offset): int(guide_offset) + G]
            prior_ia_total = float(full.sum())
        self.num_steps = int(num_steps)
        self.lrd = float(gamma) ** (1.0 / self.num_steps)
        self.initial_lr = float(initial_lr)
        n_ctrl = int(data.allele_counts_control.shape[1]) if mixture else 0
        if survival and family == "ControlNormal":
            flags |= _lib.FLAG_PRIOR_NORMAL_MU  # mu_targets ~ Normal(0, 1) (survival_model.py:142)
        shape = _lib.bean_hip_shape(
            family=_lib.FAMILY[family], selection=1 if survival else 0, flags=flags, n_reps=R, n_condits=B, n_guides=G,
            n_targets=T, n_max_alleles=A, n_edits=T if tiling else 0, n_ctrl=n_ctrl, mask_thres=int(mask_thres),
            max_target_len=max_len, guide_offset=int(guide_offset), target_offset=int(target_offset),
            n_guides_total=int(n_guides_total), n_a2e_nnz=nnz,
            # the reference holds the prior scale in a float32 tensor (model.py:406)
            sd_prior_scale=1.0 if family == "ControlNormal" else float(np.float32(sd_scale)),
            initial_lr=self.initial_lr, lrd=self.lrd, clip_norm=10.0,
            negctrl_loc=float(mu_negctrl[0]), negctrl_scale=float(mu_negctrl[1]),
            n_sample_covariates=self.n_cov,
            prior_ia_total=prior_ia_total,
        )
        self._shape = shape
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            self._check(self.lib.bean_hip_create(ctypes.byref(shape), ctypes.byref(handle)), "create")
        self._h = handle
        self._keep: Dict[str, torch.Tensor] = {}
        self.stream = torch.cuda.Stream(device=dev)

        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        f64 = lambda t: t.to(dev, torch.float64).contiguous()
        self._bind("X", f32(data.X_masked))
        self._bind("REPGUIDE", data.repguide_mask.to(dev, torch.uint8).contiguous())
        self._bind("SIZE_FACTOR", f64(data.size_factor))
        self._bind("SAMPLE_MASK", f64(data.sample_mask))
        self._bind("A0", f64(data.a0))
        if survival:
            self._bind("TIMEPOINTS", f64(data.timepoints))
            if surv_normal:
                # `mu[data.negctrl_guide_idx, :] = 0.0` (survival_model.py:59-60); indexing with None
                # (no index given) zeroes every row there, which is kept
                negmask = torch.zeros(G, dtype=torch.uint8)
                idx = getattr(data, "negctrl_guide_idx", None)
                if idx is None:
                    negmask[:] = 1
                else:
                    negmask[torch.as_tensor(np.asarray(idx), dtype=torch.int64)] = 1
                self._bind("NEGCTRL_MASK", negmask.to(dev).contiguous())
            if mixture:
                if int(data.control_timepoint.numel()) != n_ctrl:
                    raise ValueError("control_timepoint must list one time per control condition")
                self._bind("CONTROL_TIME", f64(data.control_timepoint))
            if family == "MixtureNormal":
                # observed initial abundance (survival_model.py:310), formed in float32 as the reference does
                x_t0 = data.X[:, 0, :].to(torch.float32) + 1
                if n_guides_total and n_guides_total != G:
                    # guide shard: the normaliser is the whole screen's (R,) total of X[:, 0, :] + 1
                    if t0_totals is None:
                        raise ValueError("a guide-sharded survival fit needs t0_totals (all-reduced sums of X[:, 0, :] + 1)")
                    tot0 = torch.as_tensor(t0_totals).to(torch.float32).reshape(-1, 1).to(x_t0.device)
                else:
                    tot0 = x_t0.sum(-1, keepdim=True)
                obs0 = x_t0 / tot0
                self._bind("LOG_OBS0", f64(torch.log(obs0.double())))
        else:
            z_hi, z_lo = _quantile_edges(data.upper_bounds, data.lower_bounds)
            self._bind("Z_HI", f64(z_hi))
            self._bind("Z_LO", f64(z_lo))
        # tiling: `data` may hold the guides in another order than the caller's screen (parallel.order_by_alleles);
        # guide_ids[i] is then the index of this engine's guide i in the WHOLE screen.  It keys the guide's random
        # streams (same draws as the screen order), and constrained() hands per-guide values back in screen order.
        self.guide_order = None
        if guide_ids is not None:
            if not tiling:
                raise ValueError("guide_ids: only the tiling family takes its guides in another order")
            gi = torch.as_tensor(guide_ids).to(torch.int64).cpu().reshape(-1)
            local = gi - int(guide_offset)
            if gi.numel() != G or not torch.equal(torch.sort(local).values, torch.arange(G)):
                raise ValueError("guide_ids must be a permutation of guide_offset ... guide_offset + n_guides - 1")
            self.guide_order = local.to(dev)
            self._bind("GUIDE_IDS", gi.to(torch.int32).to(dev).contiguous())
        if tiling:
            self._bind("A2E_PTR", a2e_ptr.to(dev).contiguous())
            self._bind("E2A_PTR", e2a_ptr.to(dev).contiguous())
            if nnz:
                self._bind("A2E_IDX", a2e_idx.to(dev).contiguous())
                self._bind("E2A_IDX", e2a_idx.to(dev).contiguous())
            self._bind("ALLELE_MASK", data.allele_mask.to(dev, torch.uint8).contiguous())
        else:
            self._bind("TARGET_OFFSETS", toff.to(dev).contiguous())
            self._bind("GUIDE_TO_TARGET", g2t.to(dev).contiguous())
        if self.use_bcmatch:
            self._bind("X_BC", f32(data.X_bcmatch_masked))
            self._bind("SIZE_FACTOR_BC", f64(data.size_factor_bcmatch))
            self._bind("A0_BC", f64(data.a0_bcmatch))
        if mixture:
            self._bind("ALLELE_CTRL", f32(data.allele_counts_control))
            self._bind("PI_A0", f64(data.pi_a0))
        if prior_ia is not None:
            self._bind("PRIOR_IA", f64(prior_ia))
        if self.n_cov:
            self._bind("REP_BY_COV", f64(torch.as_tensor(data.rep_by_cov).reshape(R, self.n_cov)))
        if acc:
            if data.guide_accessibility is None:
                raise ValueError("scale_by_accessibility needs data.guide_accessibility")
            self._bind("ACCESSIBILITY", f64(data.guide_accessibility))
        if prior_params is not None:
            for key, slot in (("mu_loc", "PRIOR_MU_LOC"), ("mu_scale", "PRIOR_MU_SCALE"),
                              ("sd_loc", "PRIOR_SD_LOC"), ("sd_scale", "PRIOR_SD_SCALE")):
                if key in prior_params:
                    v = torch.as_tensor(prior_params[key], dtype=torch.float64).reshape(-1)
                    if v.numel() == 1:
                        v = v.expand(T)
                    elif v.numel() != T:
                        # guide shard of a target-aligned fit: the caller holds the whole screen's
                        # per-target priors (`--prior-params`); this rank's targets are
                        # [target_offset, target_offset + T).  (Tiling shards replicate the per-edit
                        # parameters, so T is the whole edit count there and nothing is cut.)
                        if tiling or v.numel() < int(target_offset) + T:
                            raise ValueError(f"prior_params[{key!r}] has {v.numel()} entries, expected {T}"
                                             + ("" if tiling else f" (or at least {int(target_offset) + T} "
                                                                  "for this shard)"))
                        v = v[int(target_offset): int(target_offset) + T]
                    self._bind(slot, f64(v))

        # ---- parameters (unconstrained), as pyro.param initialises them
        pshape = () if family == "ControlNormal" else ((T,) if tiling else (T, 1))
        init = {"mu_loc": torch.zeros(pshape), "mu_scale": torch.zeros(pshape)}
        if not survival:  # survival models have no sd latent (bean/cli/run.py:305)
            init["sd_loc"] = torch.zeros(pshape)
            init["sd_scale"] = torch.zeros(pshape)
        if survival and family == "MixtureNormal":  # q0 = ones(G) / G over the WHOLE screen (survival_model.py:660-664)
            g_all = int(n_guides_total) if n_guides_total else G
            init["q0"] = torch.full((G,), float(np.log(np.float32(1.0) / np.float32(g_all))))
        if surv_normal:  # initial_abundance = ones(G) / G over the WHOLE screen (survival_model.py:630-634)
            g_all = int(n_guides_total) if n_guides_total else G
            init["initial_abundance"] = torch.full((G,), float(np.log(np.float32(1.0) / np.float32(g_all))))
        if mixture:
            # pyro.param("alpha_pi", ones * alpha_prior) in the guide, which runs first
            # (model.py:820-830, 927-937)
            if not float(alpha_prior) > 0:
                raise ValueError("alpha_prior must be positive")
            init["alpha_pi"] = torch.full((G, A), float(np.log(np.float32(alpha_prior))))
            if tiling:  # alpha_pi0[~allele_mask] = epsilon (model.py:643)
                init["alpha_pi"][~data.allele_mask.cpu()] = float(np.log(1e-5))
        if self.fit_noise:
            init["noise_loc"] = torch.zeros(G)
            init["noise_scale"] = torch.full((G,), float(np.log(PI_NOISE_SD)))
        if self.n_cov:  # mu_cov_loc = 0, mu_cov_scale = 1 (model.py:773-780)
            init["mu_cov_loc"] = torch.zeros(self.n_cov)
            init["mu_cov_scale"] = torch.zeros(self.n_cov)
        self.unconstrained: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self._m: Dict[str, torch.Tensor] = {}
        self._v: Dict[str, torch.Tensor] = {}
        order = list(_lib.PARAM_ORDER)
        if surv_normal:
            order[order.index("q0")] = "initial_abundance"  # same slot: a positive (G,) Dirichlet concentration
        if self.n_cov:  # the noise slots are free in the Normal family (include/bean_hip.h)
            order[order.index("noise_loc")] = "mu_cov_loc"
            order[order.index("noise_scale")] = "mu_cov_scale"
        for i, name in enumerate(order):
            if name not in init:
                continue
            p = init[name].to(dev, torch.float32).contiguous()
            self.unconstrained[name] = p
            self.grads[name] = torch.zeros_like(p)
            self._m[name] = torch.zeros_like(p)
            self._v[name] = torch.zeros_like(p)
            self._bind_slot(_lib.BUF["P"] + i, p, f"P.{name}")
            self._bind_slot(_lib.BUF["G"] + i, self.grads[name], f"G.{name}")
            self._bind_slot(_lib.BUF["M"] + i, self._m[name], f"M.{name}")
            self._bind_slot(_lib.BUF["V"] + i, self._v[name], f"V.{name}")
        cap = int(loss_capacity) if loss_capacity is not None else self.num_steps + 8
        self.loss_hist = torch.zeros(max(cap, 1), dtype=torch.float64, device=dev)
        self._bind("LOSS_HIST", self.loss_hist)
        self._noise_out: Dict[str, torch.Tensor] = {}
        if dump_noise:
            self._noise_out["eps_mu"] = torch.zeros(T, dtype=torch.float64, device=dev)
            self._bind("EPS_MU_OUT", self._noise_out["eps_mu"])
            if not survival:
                self._noise_out["eps_sd"] = torch.zeros(T, dtype=torch.float64, device=dev)
                self._bind("EPS_SD_OUT", self._noise_out["eps_sd"])
            if survival and mixture:
                self._noise_out["eps_u"] = torch.zeros(G, dtype=torch.float64, device=dev)
                self._bind("EPS_U_OUT", self._noise_out["eps_u"])
            if survival and family == "MixtureNormal":
                self._noise_out["initial_abundance"] = torch.zeros((R, G), dtype=torch.float64, device=dev)
                self._bind("X0_OUT", self._noise_out["initial_abundance"])
            if surv_normal:
                self._noise_out["q_0"] = torch.zeros((R, G), dtype=torch.float64, device=dev)
                self._bind("X0_OUT", self._noise_out["q_0"])
            if mixture:
                self._noise_out["pi"] = torch.zeros((R, G, A), dtype=torch.float64, device=dev)
                self._bind("PI_OUT", self._noise_out["pi"])
            if acc:
                self._noise_out["eps_noise"] = torch.zeros(G, dtype=torch.float64, device=dev)
                self._bind("EPS_NOISE_OUT", self._noise_out["eps_noise"])
            if self.n_cov:
                self._noise_out["eps_cov"] = torch.zeros(self.n_cov, dtype=torch.float64, device=dev)
                self._bind("EPS_NOISE_OUT", self._noise_out["eps_cov"])
        self.steps_done = 0
        with self._on_stream():
            self._check(self.lib.bean_hip_prepare(self._h, self._sptr()), "prepare")

    # -------------------------------------------------------------- plumbing
    def _check(self, status: int, what: str = ""):
        _lib.check(status, what, self.lib)

    def _bind(self, slot_name: str, t: torch.Tensor):
        self._bind_slot(_lib.BUF[slot_name], t, slot_name)

    def _bind_slot(self, slot: int, t: Optional[torch.Tensor], key: str):
        if t is None:
            self._keep.pop(key, None)
            self._check(self.lib.bean_hip_bind(self._h, slot, None, 0), f"bind {key}")
            return
        assert t.is_cuda and t.is_contiguous(), key
        self._keep[key] = t
        self._check(
            self.lib.bean_hip_bind(self._h, slot, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size()),
            f"bind {key}",
        )

    def _sptr(self):
        return ctypes.c_void_p(self.stream.cuda_stream)

    class _StreamScope:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.eng.stream.wait_stream(torch.cuda.current_stream(self.eng.device))
            return self

        def __exit__(self, *exc):
            torch.cuda.current_stream(self.eng.device).wait_stream(self.eng.stream)
            return False

    def _on_stream(self):
        return HipSVI._StreamScope(self)

    # ------------------------------------------------- guide-sharded stepping
    def exchange_buffers(self) -> Dict[str, torch.Tensor]:
        """Allocate and bind the exchange buffers this family needs when its guides are sharded
        over ranks (``bean_hip_sharded_*``): ``gsum`` (survival MixtureNormal) and/or ``tgrad``
        (ControlNormal, tiling).  Empty for the families that need no exchange."""
        if getattr(self, "_xchg", None) is None:
            self._xchg = {}
            R = self.data.n_reps
            # what is exchanged BEHIND the guide kernel lives in one allocation, so a step's post-guide exchange is
            # ONE all-reduce whatever the family (tgrad | sq | cov are views of `_xchg_post`)
            sizes = []
            if self.family in ("ControlNormal", "MultiMixtureNormal"):
                sizes.append(("tgrad", 2 * self.T, "XCHG_TGRAD"))
            if self.surv_normal:
                sizes.append(("sq", R, "XCHG_SQ"))
            if self.n_cov and int(self._shape.n_guides_total) not in (0, self.data.n_guides):
                # guide shard of a screen with sample covariates: mu_cov is shared by every guide
                sizes.append(("cov", R, "XCHG_COV"))
            self._xchg_post = None
            if sizes:
                self._xchg_post = torch.zeros(sum(n for _, n, _ in sizes), dtype=torch.float64, device=self.device)
                at = 0
                for name, n, slot in sizes:
                    self._xchg[name] = self._xchg_post[at:at + n]
                    self._bind(slot, self._xchg[name])
                    at += n
            if self.survival and self.family in ("MixtureNormal", "Normal"):
                self._xchg["gsum"] = torch.zeros(R + 1, dtype=torch.float64, device=self.device)
                self._bind("XCHG_GSUM", self._xchg["gsum"])
        return self._xchg

    def init_native_comm(self, group=None) -> bool:
        """Give the library its own RCCL communicator over the ranks of ``group`` (default group when
        None), so that ``run_exchanged`` steps without the host in the loop
        (``bean_hip_svi_run_exchanged``: kernels and ``ncclAllReduce`` enqueued by the library on the
        engine's stream).  Collective: every rank of the group must call it.  Only with the ``nccl``
        backend (the unique id travels by a ``torch.distributed`` broadcast); returns False - and the
        Python stepping loop stays in use - on any other backend or if RCCL refuses."""
        import torch.distributed as dist

        if getattr(self, "_native_comm", False):
            return True
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
            return False
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        path = _lib.rccl_path().encode()
        idt = torch.zeros(128, dtype=torch.uint8, device=self.device)
        # EVERY rank loads RCCL now (asking for an id does that; only rank 0's id is used) and the ranks agree
        # on the outcome BEFORE anyone enters ncclCommInitRank: a rank that could not load the library would
        # otherwise return while its peers block inside that collective
        buf = (ctypes.c_uint8 * 128)()
        loaded = self.lib.bean_hip_comm_unique_id(path, buf) == 0
        if rank == 0 and loaded:
            idt.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
        ok = torch.tensor([1 if loaded else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            return False
        dist.broadcast(idt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(idt.cpu().numpy().tobytes())
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(raw)
        self.exchange_buffers()
        with torch.cuda.device(self.device):
            st = self.lib.bean_hip_comm_init(self._h, path, buf, rank, world)
        # all ranks agree on the outcome (a rank that failed alone would leave the others in a collective)
        flag = torch.tensor([1 if st == 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if st == 0:
                self.lib.bean_hip_comm_destroy(self._h)
            return False
        # ... and a fit depends on the communicator only after it has summed a known vector right: rank k contributes
        # (k + 1) * (1, 2, ..., n) - every entry of the sum is an exact integer in float64 - through the very call the
        # stepping loop issues (ncclAllReduce, sum, float64, in place, the engine's stream), over a length that takes
        # RCCL through its multi-channel path as the per-edit gradients of a tiling fit do; the ranks agree on the
        # outcome, and on any doubt all of them keep the Python stepping loop
        n = 4099
        probe = torch.arange(1, n + 1, dtype=torch.float64, device=self.device) * (rank + 1)
        with self._on_stream():
            rc = self.lib.bean_hip_comm_all_reduce(self._h, ctypes.c_void_p(probe.data_ptr()), n, self._sptr())
        self.stream.synchronize()
        want = torch.arange(1, n + 1, dtype=torch.float64, device=self.device) * (world * (world + 1) // 2)
        good = rc == 0 and bool(torch.equal(probe, want))
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            self.lib.bean_hip_comm_destroy(self._h)
            return False
        self._native_comm = True
        return True

    def run_exchanged(self, n_steps: int, all_reduce, seed: int = 101, first_step: Optional[int] = None,
                      graph_chunk: Optional[int] = None):
        """``n_steps`` SVI steps of one guide shard.  With a native communicator (``init_native_comm``)
        the whole loop, collectives included, is enqueued by the library; otherwise the step is driven
        from here and ``all_reduce(tensor)`` must sum the tensor over the ranks in place on the current
        stream (``torch.distributed.all_reduce``).  ``graph_chunk`` > 1 (native path; default from
        ``BEAN_HIP_XCHG_GRAPH``, else 0) replays hipGraphs of that many steps, collectives captured."""
        first = self.steps_done if first_step is None else int(first_step)
        if first + n_steps > self.loss_hist.numel():
            raise ValueError("loss history too small: raise num_steps / loss_capacity")
        x = self.exchange_buffers()
        sp = self._sptr()
        if getattr(self, "_native_comm", False):
            import os

            chunk = int(os.environ.get("BEAN_HIP_XCHG_GRAPH", "0")) if graph_chunk is None else int(graph_chunk)
            with self._on_stream():
                self._check(self.lib.bean_hip_svi_run_exchanged(self._h, int(seed), first, int(n_steps), chunk, sp),
                            "svi_run_exchanged")
            self.last_exchange_path = ("native: kernels + ncclAllReduce enqueued by the library"
                                       + (f", hipGraphs of up to {chunk} steps" if chunk > 1 else ", eager launches"))
            self.steps_done = first + n_steps
            return
        self.last_exchange_path = ("python loop: 3 ctypes calls + at most 2 torch.distributed.all_reduce per step "
                                   "(one in front of the guide kernel, one packed behind it)")
        lib, h = self.lib, self._h
        sums, guide, update = lib.bean_hip_sharded_sums, lib.bean_hip_sharded_guide, lib.bean_hip_sharded_update
        pre, post = x.get("gsum"), self._xchg_post
        with self._on_stream(), torch.cuda.stream(self.stream):
            self._check(lib.bean_hip_sharded_begin(h, int(seed), first, int(n_steps), sp), "sharded_begin")
            for i in range(n_steps):
                rc = sums(h, sp)
                if pre is not None:
                    all_reduce(pre)
                rc |= guide(h, sp)
                if post is not None:
                    all_reduce(post)
                rc |= update(h, 1 if i == n_steps - 1 else 0, sp)
                if rc:
                    self._check(rc, "sharded step (sums / guide / update)")
        self.steps_done = first + n_steps

    # the four phases, for drivers that interleave several engines in one process (tests)
    def phase(self, name: str, *args):
        fn = getattr(self.lib, "bean_hip_sharded_" + name)
        with self._on_stream():
            self._check(fn(self._h, *args, self._sptr()), "sharded_" + name)

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            if getattr(self, "_native_comm", False):
                self.lib.bean_hip_comm_destroy(self._h)
                self._native_comm = False
            self.lib.bean_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ operations
    def set_noise(self, noise: Optional[dict]):
        """Inject the draws of the next evaluation(s) (parity tests);
        ``None`` returns to the in-kernel generator."""
        dev = self.device
        names = {"eps_mu": "EPS_MU_IN", "eps_sd": "EPS_SD_IN", "pi": "PI_IN", "eps_noise": "EPS_NOISE_IN",
                 "eps_u": "EPS_U_IN"}
        # the draw of the Dirichlet-over-guides site: `initial_abundance` in the survival MixtureNormal
        # guide, `q_0` (site initial_guide_abundance) in the survival NormalModel
        names["q_0" if self.surv_normal else "initial_abundance"] = "X0_IN"
        if self.n_cov:
            names.pop("eps_noise")
            names["eps_cov"] = "EPS_NOISE_IN"
        if noise is not None and "mu_negctrl" in noise and "eps_u" not in noise:
            m0, s0 = float(np.float32(self._shape.negctrl_loc)), float(np.float32(self._shape.negctrl_scale))
            noise = dict(noise)
            noise["eps_u"] = (torch.as_tensor(noise["mu_negctrl"]).double() - m0) / s0
        for key, slot in names.items():
            t = None if noise is None else noise.get(key)
            if t is None:
                if slot in self._keep:
                    self._bind_slot(_lib.BUF[slot], None, slot)
                continue
            t = torch.as_tensor(t).to(dev, torch.float64)
            if key == "pi":
                t = t.reshape(self.data.n_reps, self.data.n_guides, self.A)
                t = self._to_engine_order(t, 1)
            elif slot == "X0_IN":
                t = self._to_engine_order(t.reshape(self.data.n_reps, self.data.n_guides), 1).reshape(-1)
            else:
                t = t.reshape(-1)
                if key in ("eps_noise", "eps_u"):
                    t = self._to_engine_order(t, 0)
            self._bind(slot, t.contiguous())

    # With guide_ids the engine holds its guides in another order than the caller's screen: engine guide i is the
    # screen's (shard's) guide guide_order[i].  Everything per-guide that crosses this class - injected and exported
    # draws, gradients, constrained() - is in SCREEN order.
    def _to_engine_order(self, t: torch.Tensor, axis: int) -> torch.Tensor:
        return t if self.guide_order is None else t.index_select(axis, self.guide_order)

    def _to_screen_order(self, t: torch.Tensor, axis: int) -> torch.Tensor:
        if self.guide_order is None:
            return t
        return torch.empty_like(t).index_copy_(axis, self.guide_order, t)

    def elbo_grad(self, step: int = 0, seed: int = 101, loss_index: int = 0):
        """One ELBO evaluation: returns ``(loss, {name: grad})`` w.r.t. the
        unconstrained parameters; parameters are not modified."""
        with self._on_stream():
            self._check(
                self.lib.bean_hip_elbo_grad(self._h, int(seed), int(step), int(loss_index), self._sptr()),
                "elbo_grad",
            )
        torch.cuda.synchronize(self.device)
        grads = {k: (self._to_screen_order(v, 0) if k in PER_GUIDE and self.guide_order is not None else v.clone())
                 for k, v in self.grads.items()}
        return float(self.loss_hist[loss_index]), grads

    def drawn_noise(self) -> Dict[str, torch.Tensor]:
        """Draws used by the last evaluation (needs ``dump_noise=True``)."""
        out = {k: v.clone() for k, v in self._noise_out.items()}
        if self.guide_order is not None:  # per-guide draws go back in screen order (see _to_screen_order)
            for k, axis in (("pi", 1), ("eps_noise", 0), ("eps_u", 0), ("initial_abundance", 1), ("q_0", 1)):
                if k in out and not (k == "eps_noise" and self.n_cov):
                    out[k] = self._to_screen_order(out[k], axis)
        if "pi" in out:
            out["pi"] = out["pi"].unsqueeze(1)  # (R, 1, G, A) as the reference shapes it
        if "eps_u" in out:  # the oracle takes the baseline draw itself: u = m0 + s0 * eps (float32 constants)
            m0, s0 = np.float32(self._shape.negctrl_loc), np.float32(self._shape.negctrl_scale)
            out["mu_negctrl"] = float(m0) + out.pop("eps_u") * float(s0)
        if self.family == "MultiMixtureNormal":
            pass  # per-edit draws are 1-D, as the reference shapes them
        elif self.family != "ControlNormal":
            for k in ("eps_mu", "eps_sd"):
                if k in out:
                    out[k] = out[k].reshape(self.T, 1)
        else:
            for k in ("eps_mu", "eps_sd"):
                if k in out:
                    out[k] = out[k].reshape(())
        return out

    def adam(self, t: int):
        with self._on_stream():
            self._check(self.lib.bean_hip_adam(self._h, int(t), self._sptr()), "adam")

    def run(self, n_steps: int, seed: int = 101, graph_chunk: int = 50, first_step: Optional[int] = None,
            resume: bool = False):
        """Enqueue ``n_steps`` fused SVI steps (no host synchronisation).

        ``resume=True`` (``bean_hip_svi_resume``): for a fit stepped in windows.  Same results; a call that
        continues exactly where the previous one ended starts stepping at once (the previous call has left
        the next step's draw and tables on the device).  The caller promises not to write the parameter /
        moment tensors between such calls - ``run_inference`` and ``bench.py`` do not."""
        first = self.steps_done if first_step is None else int(first_step)
        if first + n_steps > self.loss_hist.numel():
            raise ValueError("loss history too small: raise num_steps / loss_capacity")
        # a write to a parameter or moment tensor between two windows (warm start, clamping, a loaded checkpoint)
        # bumps the tensor's version counter - the kernels' own writes do not: the draw and the tables the previous
        # window left on the device belong to the OLD values then, so this window goes through the plain loop, as
        # include/bean_hip.h asks after such a write (the library cannot see it; a first_step / seed that does not
        # continue it does see)
        versions = self._tensor_versions()
        prev = getattr(self, "_resume_versions", None)
        if resume and ((prev is not None and versions != prev) or getattr(self, "_resume_broken", False)):
            resume = False
        self._resume_broken = False
        fn = self.lib.bean_hip_svi_resume if resume else self.lib.bean_hip_svi_run
        with self._on_stream():
            self._check(fn(self._h, int(seed), first, int(n_steps), int(graph_chunk), self._sptr()),
                        "svi_resume" if resume else "svi_run")
        self._resume_versions = versions
        self.steps_done = first + n_steps

    def _tensor_versions(self):
        return tuple(t._version for d in (self.unconstrained, self._m, self._v) for t in d.values())

    def invalidate_resume(self):
        """Break the chain of resumed windows after writing parameters or moments through raw pointers (writes
        through torch are seen by themselves): the next ``run(resume=True)`` is a plain ``bean_hip_svi_run``."""
        self._resume_broken = True

    def losses(self):
        torch.cuda.synchronize(self.device)
        return self.loss_hist[: self.steps_done].cpu().tolist()

    def set_profile(self, enable):
        """0 off; 1 time the dominant (guide) kernel; 2 time the fused k_param launches instead."""
        self._check(self.lib.bean_hip_set_profile(self._h, int(enable)), "set_profile")

    def get_profile(self):
        avg, n = ctypes.c_double(0.0), ctypes.c_uint64(0)
        self._check(self.lib.bean_hip_get_profile(self._h, ctypes.byref(avg), ctypes.byref(n)), "get_profile")
        return avg.value, n.value

    @property
    def step_bytes(self) -> int:
        return int(self.lib.bean_hip_step_bytes(self._h))

    @property
    def dominant_kernel(self) -> str:
        return self.lib.bean_hip_dominant_kernel(self._h).decode()

    @property
    def dominant_kernel_variant(self) -> str:
        """The template instantiation launched for this shape, as spelt in the code object."""
        return self.lib.bean_hip_dominant_kernel_variant(self._h).decode()

    @property
    def dominant_lds_bytes(self) -> int:
        """Dynamic LDS per workgroup the library requests for the dominant kernel at this shape."""
        return int(self.lib.bean_hip_dominant_lds_bytes(self._h))

    def snapshot(self) -> Dict[str, torch.Tensor]:
        """Device-side copy of the unconstrained parameters as they are when everything enqueued so far has
        run (``run_inference`` keeps the one taken at the start of a report window: if the window's loss
        turns non-finite, that is what ``tmp_result.pkl`` holds, not the NaN updates applied since)."""
        with self._on_stream():
            return {k: v.clone() for k, v in self.unconstrained.items()}

    def constrained(self, snapshot: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Constrained parameter values, as ``pyro.get_param_store()[name]`` (of ``snapshot`` if given)."""
        torch.cuda.synchronize(self.device)
        src = self.unconstrained if snapshot is None else snapshot
        out = {k: (v.exp() if k in POSITIVE else v.clone()) for k, v in src.items()}
        if self.survival and self.family == "MultiMixtureNormal":
            # the reference's tiling survival guide registers this parameter and never uses it
            # (survival_model.py:770-774): it keeps its initial value
            g_all = int(self._shape.n_guides_total) or self.data.n_guides
            out["initial_abundance"] = torch.full((self.data.n_guides,), 1.0 / g_all, dtype=torch.float32,
                                                  device=self.device)
        if self.guide_order is not None:
            # engine guide i is the screen's guide guide_order[i]: per-guide values go back in screen order
            for k in PER_GUIDE:
                if k in out:
                    out[k] = self._to_screen_order(out[k], 0)
        return out


def test_special(op: int, a, x=None, b=None, device="cuda:0"):
    """Evaluate the device special functions (``bean_hip_test_special``)."""
    lib = _lib.load()
    dev = torch.device(device)
    a = torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    x = torch.zeros_like(a) if x is None else torch.as_tensor(x, dtype=torch.float64, device=dev).contiguous()
    b = torch.zeros_like(a) if b is None else torch.as_tensor(b, dtype=torch.float64, device=dev).contiguous()
    o0, o1 = torch.zeros_like(a), torch.zeros_like(a)
    s = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(
        lib.bean_hip_test_special(
            int(op), a.numel(), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(x.data_ptr()),
            ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(o0.data_ptr()), ctypes.c_void_p(o1.data_ptr()),
            ctypes.c_void_p(s),
        ),
        "test_special",
    )
    torch.cuda.synchronize(dev)
    return o0, o1
