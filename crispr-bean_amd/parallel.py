"""Guide-sharded multi-GPU SVI: one process per GPU (``torch.distributed``).

The reference has no distributed code.  The scaling axis of its hot path is the
guide dimension, and guides are stored target-sorted
(``bean/preprocessing/data_class.py:511-532``), so the screen is cut on *target
boundaries*: in the variant families every parameter is per-target or per-guide
(``bean/model/model.py:800-830``), hence parameters, gradients and ClippedAdam
state are shard-local and no gradient crosses GPUs.  What is exchanged:

* every ``report_every`` steps (100, the reference's loss-print cadence,
  ``bean/model/run.py:378``): one all-reduce (RCCL over xGMI with the ``nccl``
  backend) of that window of the per-step loss vector;
* once at the end: an all-gather of the fitted parameters so every rank
  returns the whole-screen parameter store.

Random streams are keyed by global guide/target indices
(``bean_hip_shape.guide_offset`` ...), so an N-GPU fit reproduces the 1-GPU fit
bit for bit.

Families in which something IS shared across shards step through
``HipSVI.run_exchanged`` instead, with one small all-reduce at each exchange point of the step:

* tiling (``MultiMixtureNormal``): the per-edit parameters are replicated on every rank and the
  guides are cut anywhere; per step one all-reduce of the per-edit likelihood gradients
  ``(2, E)`` float64 between the guide kernel and the parameter update (every rank then applies the
  same ClippedAdam update); rank 0 counts the replicated parameters' prior/entropy terms;
* survival ``MixtureNormal``: target-aligned shards as above, plus per step one all-reduce of the
  ``R + 1`` normalisers of the Dirichlet-over-all-guides draw; the observed-abundance totals are
  formed once from the whole screen.
These fits agree with the single-GPU fit up to the regrouping of float64 sums.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

Shard = Tuple[int, int, int, int]  # (guide_start, guide_end, target_start, target_end)


def plan_shards(target_lengths: Sequence[int], world_size: int) -> List[Shard]:
    """Cut the target-sorted guide axis into ``world_size`` contiguous,
    target-aligned shards of near-equal guide count.  Shards may be empty when
    there are fewer targets than ranks."""
    tl = np.asarray(target_lengths, dtype=np.int64)
    T, G = tl.size, int(tl.sum())
    ends = np.cumsum(tl)  # guide index after each target
    shards: List[Shard] = []
    t0 = g0 = 0
    for k in range(world_size):
        left = world_size - 1 - k  # ranks after this one
        if k == world_size - 1 or t0 >= T:
            t1 = T
        else:
            want = (k + 1) * G / world_size
            # target boundary closest to the ideal cut ...
            j = int(np.searchsorted(ends, want, side="left"))
            if j + 1 < T and j >= t0 and abs(ends[j] - want) > abs(want - (ends[j - 1] if j > 0 else 0)):
                j -= 1
            t1 = j + 1
            # ... keeping at least one target here and one for every later rank
            t1 = max(t1, t0 + 1)
            t1 = min(t1, max(T - left, t0 + 1), T)
        g1 = int(ends[t1 - 1]) if t1 > 0 else 0
        shards.append((g0, g1, t0, t1))
        t0, g0 = t1, g1
    assert shards[-1][1] == G and shards[-1][3] == T
    return shards


def plan_guide_shards(n_guides: int, world_size: int, n_targets: int = 0) -> List[Shard]:
    """Near-equal contiguous guide ranges for families whose per-target parameters are replicated
    (tiling): no target alignment is needed; every shard sees all ``n_targets`` targets."""
    cuts = [round(k * n_guides / world_size) for k in range(world_size + 1)]
    return [(cuts[k], cuts[k + 1], 0, n_targets) for k in range(world_size)]


def order_by_alleles(data, guide_offset: int = 0):
    """Tiling screens: the guides ordered by their number of alleles, most first (stable), and the index of each
    in the caller's whole screen - ``(ordered_data, guide_ids)``, or ``(data, None)`` when there is nothing to
    reorder.

    The register-resident tiling kernels give every (replicate, guide) a lane and walk the allele SLOTS; a masked
    slot takes other branches of the implicit-gradient and count code than a slot that holds an allele, so a
    wave whose guides have 2 ... 8 alleles in screen order runs both sides of each.  With its guides' alleles
    in the same slots it runs one: 179 -> 159 us per step on BASELINE configs[2]'s shape
    (``scripts/micro/tiling_sorted.py``).  ``HipSVI(..., guide_ids=)`` keys the random streams by the screen
    index, so the draws are those of the screen order, and returns per-guide values in screen order.
    ``BEAN_HIP_ORDER_GUIDES=0`` keeps the screen order."""
    import os

    mask = getattr(data, "allele_mask", None)
    if mask is None or os.environ.get("BEAN_HIP_ORDER_GUIDES", "1") == "0":
        return data, None
    n_al = mask.sum(1).cpu().numpy()
    perm = np.argsort(-n_al, kind="stable")
    if np.array_equal(perm, np.arange(perm.size)):
        return data, None
    return data[perm], torch.as_tensor(perm + int(guide_offset), dtype=torch.int64)


def shard_screen(data, shard: Shard):
    """Per-rank view of the screen.  Per-sample tensors (size factors, masks,
    bin edges) are global and shared; per-guide tensors are sliced."""
    g0, g1, _, _ = shard
    return data[np.arange(g0, g1)]


class _Group:
    """torch.distributed helpers that also work single-process."""

    def __init__(self, group=None):
        self.on = dist.is_available() and dist.is_initialized()
        self.group = group

    @property
    def world(self):
        return dist.get_world_size(self.group) if self.on else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if self.on else 0

    def all_reduce_sum(self, t: torch.Tensor):
        # also with one rank: the collective is then a copy, but the call (RCCL stream interplay with
        # the engine's stream and graph replay) is the same one an N-rank fit makes
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather_rows(self, t: torch.Tensor, sizes: Sequence[int]) -> torch.Tensor:
        """Concatenate per-rank tensors with different leading sizes."""
        if not self.on:
            return t
        m = max(max(sizes), 1)
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)


def check_window_finite(window: torch.Tensor, first_step: int) -> None:
    """Raise ``FloatingPointError`` naming the first non-finite loss of a report window (one small
    device-to-host read per window).  In a sharded fit the window is the all-reduced one, identical on
    every rank: all ranks raise together, none is left waiting in a collective."""
    bad = ~torch.isfinite(window)
    if bool(bad.any()):
        raise FloatingPointError(f"non-finite loss at iteration {first_step + int(torch.nonzero(bad)[0])}")


PER_TARGET = ("mu_loc", "mu_scale", "sd_loc", "sd_scale")
from .engine import PER_GUIDE  # noqa: E402  (parameters with a guide axis)
REPLICATED = ("mu_cov_loc", "mu_cov_scale")  # shared by every guide: identical on every rank


def native_comm_enabled() -> bool:
    """Exchange families of a guide-sharded fit step through the library-owned RCCL communicator
    (``HipSVI.init_native_comm``) - no host in the per-step loop, the collectives on the engine's own stream
    (``torch.distributed`` runs its collectives on a stream of its own: the two cross-stream hand-overs per step
    idle the device ~10 us each, ``profiles/r05_exchange_kernel_timeline.txt``).  Default since round 5: the
    communicator is CHECKED when it is made - every rank loads RCCL before anyone enters ``ncclCommInitRank``, the
    ranks agree on the outcome, and a known vector must come back summed exactly through the very call the stepping
    loop issues - and any doubt leaves every rank on the Python stepping loop.  ``BEAN_HIP_NATIVE_COMM=0`` forces
    that loop."""
    import os

    return os.environ.get("BEAN_HIP_NATIVE_COMM", "1") != "0"


def run_sharded(
    engine_factory: Callable,
    data,
    num_steps: int,
    *,
    seed: int = 101,
    report_every: int = 100,
    group=None,
    on_report: Optional[Callable[[int, float], None]] = None,
):
    """Fit the whole screen with one engine per rank.

    ``engine_factory(shard_data, shard, n_guides_total)`` must return an object
    with ``run(k, seed=)``, ``loss_hist`` (tensor of per-step losses),
    ``steps_done``, ``constrained()`` and ``close()`` - ``HipSVI`` on a GPU, or a
    CPU stand-in in the gloo tests.  Returns ``(constrained_params, losses)`` for
    the WHOLE screen on every rank.
    """
    grp = _Group(group)
    replicated_targets = getattr(data, "target_lengths", None) is None  # tiling: per-edit parameters
    if replicated_targets:
        shards = plan_guide_shards(data.n_guides, grp.world, getattr(data, "n_targets", 0))
    else:
        shards = plan_shards(data.target_lengths.cpu().numpy(), grp.world)
    mine = shards[grp.rank]
    # `shards` is identical on every rank: all of them raise (before any engine or collective exists),
    # so no rank is left waiting in an all-reduce for one that has gone
    empty = [k for k, s in enumerate(shards) if s[1] - s[0] == 0]
    if empty:
        raise ValueError(
            f"rank(s) {empty} would receive no guides: {getattr(data, 'n_targets', 0)} targets / "
            f"{data.n_guides} guides cannot feed {grp.world} ranks"
        )
    extra = {}
    if replicated_targets or getattr(data, "sample_covariates", None) is not None:
        # replicated parameters (tiling per-edit parameters; mu_cov of a screen with sample covariates):
        # one rank counts their prior / entropy terms
        extra["loss_owner"] = grp.rank == 0
    if getattr(data, "selection", "sorting") == "survival":
        # observed initial abundance is normalised over the whole screen (survival_model.py:306-311)
        extra["t0_totals"] = (data.X[:, 0, :].to(torch.float32) + 1).sum(-1)
    eng = engine_factory(shard_screen(data, mine), mine, data.n_guides, **extra)
    exchanged = bool(eng.exchange_buffers()) if hasattr(eng, "exchange_buffers") else False
    if exchanged and grp.on and hasattr(eng, "init_native_comm") and native_comm_enabled():
        # the library steps with its own RCCL communicator, no host in the per-step loop - once the communicator has
        # passed its check (init_native_comm: False, and the Python stepping loop + torch.distributed.all_reduce
        # stays, on a non-nccl backend, if RCCL refuses, or if the probe sum comes back wrong on any rank)
        eng.init_native_comm(group)
    done = 0
    while done < num_steps:
        k = min(report_every, num_steps - done)
        if hasattr(eng, "snapshot"):
            eng.window_start = eng.snapshot()  # what a halt inside this window dumps (model/run.py)
        if exchanged:
            eng.run_exchanged(k, grp.all_reduce_sum, seed=seed)
        else:
            eng.run(k, seed=seed, resume=True)
        window = eng.loss_hist[done : done + k]
        stream = getattr(eng, "stream", None)
        if stream is not None:
            with torch.cuda.stream(stream):
                grp.all_reduce_sum(window)
        else:
            grp.all_reduce_sum(window)
        try:
            check_window_finite(window, done)
        except FloatingPointError:
            eng.steps_done = done + k
            raise
        if on_report is not None:
            on_report(done, float(window[0]))
        done += k
    losses = eng.loss_hist[:num_steps].detach().cpu().tolist()
    local = eng.constrained()
    g_sizes = [s[1] - s[0] for s in shards]
    t_sizes = [s[3] - s[2] for s in shards]
    whole: Dict[str, torch.Tensor] = {}
    for name, t in local.items():
        if (replicated_targets and name not in PER_GUIDE) or name in REPLICATED:
            whole[name] = t  # identical on every rank
            continue
        sizes = g_sizes if name in PER_GUIDE else t_sizes
        whole[name] = grp.all_gather_rows(t.contiguous(), sizes)
    eng.close()
    return whole, losses
