"""N > 1 path on CPU: world_size-2 ``gloo`` run of the sharded SVI driver.

The HIP engine needs a GPU, so the ranks use a CPU stand-in with the same
interface, built on the oracle and fed draws keyed by *global* guide/target
indices (as the HIP kernels key their Philox streams).  The assertions are the
ones the multi-GPU design rests on: target-aligned shards need no data-path
exchange, the all-reduced loss equals the unsharded loss, and the gathered
parameters equal the single-process fit.
"""
import os
import socket

import numpy as np
import pytest
import scipy.stats as st
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bean_amd  # noqa: F401
from bean_amd import parallel
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
from oracle import elbo, svi

N_STEPS = 7
FAMILY = "MixtureNormal"


class OracleEngine:
    """CPU stand-in for HipSVI (same methods), for tests only."""

    stream = None

    def __init__(self, data, shard, n_guides_total, whole_targets, num_steps=2000):
        self.data, self.shard = data, shard
        self.G_tot, self.T_tot = n_guides_total, whole_targets
        self.params = elbo.init_params(FAMILY, data)
        self.optim = svi.ClippedAdam(self.params, lr=0.01, lrd=0.1 ** (1 / num_steps))
        self.loss_hist = torch.zeros(num_steps, dtype=torch.float64)
        self.steps_done = 0

    def _noise(self, seed, step):
        g0, g1, t0, t1 = self.shard
        rng = np.random.default_rng([seed, step])
        R = self.data.n_reps
        eps_mu = rng.standard_normal((self.T_tot, 1))
        eps_sd = rng.standard_normal((self.T_tot, 1))
        u = rng.random((R, self.G_tot))
        with torch.no_grad():
            a = self.params["alpha_pi"].exp().double()
            conc = (a / a.sum(-1, keepdim=True) * self.data.pi_a0[:, None]).clamp(1e-5).numpy()
        p1 = st.beta.ppf(u[:, g0:g1], conc[None, :, 1], conc[None, :, 0]).clip(1e-12, 1 - 1e-12)
        pi = np.stack([1 - p1, p1], -1)[:, None]
        return {"eps_mu": torch.tensor(eps_mu[t0:t1]), "eps_sd": torch.tensor(eps_sd[t0:t1]),
                "pi": torch.tensor(pi)}

    def run(self, k, seed=101, resume=False):
        for _ in range(k):
            s = self.steps_done
            self.loss_hist[s] = svi.svi_step(elbo.LOSSES[FAMILY], self.data, self.params, self.optim,
                                             noise=self._noise(seed, s))
            self.steps_done += 1

    def constrained(self):
        return {k: v.detach().clone() for k, v in elbo.constrained(self.params).items()}

    def close(self):
        pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = make_sorting_variant_screen(90, 2, seed=5, guides_per_target=4)
        factory = lambda d, shard, gtot: OracleEngine(d, shard, gtot, data.n_targets)
        whole, losses = parallel.run_sharded(factory, data, N_STEPS, seed=3, report_every=3)
        torch.save({"params": whole, "losses": losses}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_fit_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    data = make_sorting_variant_screen(90, 2, seed=5, guides_per_target=4)
    single = OracleEngine(data, (0, data.n_guides, 0, data.n_targets), data.n_guides, data.n_targets)
    single.run(N_STEPS, seed=3)
    ref_params, ref_losses = single.constrained(), single.loss_hist[:N_STEPS].tolist()
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    for out in outs:  # every rank returns the whole screen
        # the oracle sums its float32 sites per shard: agreement to float32 rounding
        np.testing.assert_allclose(out["losses"], ref_losses, rtol=1e-6)
        assert set(out["params"]) == set(ref_params)
        for k, v in out["params"].items():
            assert v.shape == ref_params[k].shape, k
            np.testing.assert_allclose(v.numpy(), ref_params[k].numpy(), rtol=2e-6, atol=1e-7)
    for k in ref_params:
        assert torch.equal(outs[0]["params"][k], outs[1]["params"][k])


def test_plan_shards_properties():
    rng = np.random.default_rng(0)
    for _ in range(200):
        T = int(rng.integers(1, 40))
        tl = rng.integers(1, 9, T)
        W = int(rng.integers(1, 9))
        shards = parallel.plan_shards(tl, W)
        assert len(shards) == W
        assert shards[0][0] == 0 and shards[0][2] == 0
        assert shards[-1][1] == tl.sum() and shards[-1][3] == T
        off = np.concatenate([[0], np.cumsum(tl)])
        for (g0, g1, t0, t1), nxt in zip(shards, shards[1:] + [None]):
            assert g0 == off[t0] and g1 == off[t1]          # cuts fall on target boundaries
            if nxt is not None:
                assert nxt[0] == g1 and nxt[2] == t1        # contiguous, no overlap
        if T >= W:
            assert all(s[3] > s[2] for s in shards)         # nobody is left empty
    # balance on a regular library
    shards = parallel.plan_shards([5] * 10000, 8)
    sizes = [s[1] - s[0] for s in shards]
    assert max(sizes) - min(sizes) <= 5


def test_shard_screen_slices_per_guide_tensors_only():
    data = make_sorting_variant_screen(50, 2, seed=9, guides_per_target=3)
    shards = parallel.plan_shards(data.target_lengths.numpy(), 3)
    parts = [parallel.shard_screen(data, s) for s in shards]
    assert sum(p.n_guides for p in parts) == data.n_guides
    assert sum(p.n_targets for p in parts) == data.n_targets
    assert torch.equal(torch.cat([p.X for p in parts], dim=2), data.X)
    assert torch.equal(torch.cat([p.a0 for p in parts]), data.a0)
    assert torch.equal(torch.cat([p.target_lengths for p in parts]), data.target_lengths)
    for p in parts:
        assert torch.equal(p.size_factor, data.size_factor)      # per-sample: global
        assert torch.equal(p.upper_bounds, data.upper_bounds)
        p.validate()


def test_empty_rank_is_an_error():
    data = make_sorting_variant_screen(6, 2, seed=9, guides_per_target=3)  # 2 targets
    shards = parallel.plan_shards(data.target_lengths.numpy(), 4)
    assert [s[1] - s[0] for s in shards].count(0) == 2


def _empty_shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = make_sorting_variant_screen(6, 2, seed=9, guides_per_target=3)  # 2 targets for 3 ranks
    made = []

    def factory(shard_data, shard, n_total, **extra):
        made.append(shard)
        raise AssertionError("no engine may be built when a rank has no guides")

    try:
        parallel.run_sharded(factory, data, 3)
        outcome = "returned"
    except ValueError as exc:
        outcome = "ValueError" if "no guides" in str(exc) else f"other: {exc}"
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write(f"{outcome} {len(made)}")
    dist.destroy_process_group()


def test_every_rank_raises_when_one_shard_is_empty(tmp_path):
    """Fewer targets than ranks: ALL ranks raise before any engine or collective exists (a rank that
    raised alone would leave the others waiting in an all-reduce until the backend's timeout)."""
    world = 3
    mp.spawn(_empty_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ValueError 0"


# ------------------------------------------------- exchange families (tiling, survival)
class ExchangeEngine:
    """Minimal stand-in with the interface of a HipSVI whose family needs per-step exchanges: the
    "fit" is arithmetic whose result depends on every rank's data only through the all-reduced
    buffer, so the plumbing of run_sharded (shard plan, extras, per-step all-reduce, gathering) is
    what is checked."""

    stream = None

    def __init__(self, data, shard, n_total, replicated, **extra):
        self.data, self.shard, self.extra = data, shard, extra
        self.replicated = replicated
        self.xchg = {"tgrad" if replicated else "gsum": torch.zeros(3, dtype=torch.float64)}
        self.loss_hist = torch.zeros(64, dtype=torch.float64)
        self.steps_done = 0
        self.shared = torch.zeros(3, dtype=torch.float64)
        self.per_guide = torch.zeros(data.n_guides, 2)

    def exchange_buffers(self):
        return self.xchg

    def run_exchanged(self, k, all_reduce, seed=101):
        buf = next(iter(self.xchg.values()))
        for _ in range(k):
            s = self.steps_done
            buf[:] = torch.tensor([self.data.X.sum(), self.data.n_guides, s + 1.0], dtype=torch.float64)
            all_reduce(buf)
            self.shared += buf
            self.per_guide += self.data.a0[:, None].float() * (s + 1)
            self.loss_hist[s] = float(self.data.n_guides)
            self.steps_done += 1

    def run(self, *a, **k):
        raise AssertionError("exchange families must step through run_exchanged")

    def constrained(self):
        name = "mu_loc" if self.replicated else "q0"
        out = {"alpha_pi": self.per_guide.clone()}
        if self.replicated:
            out["mu_loc"] = self.shared.clone()
        else:
            out["q0"] = self.per_guide[:, 0].clone()
            out["mu_loc"] = torch.full((self.data.n_targets, 1), float(self.shared[0]))
        return out

    def close(self):
        pass


def _exchange_worker(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if kind == "tiling":
            from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen
            data = make_sorting_tiling_screen(61, 2, seed=5, n_max_alleles=4)
        else:
            from bean_amd.preprocessing.synthetic import make_survival_variant_screen
            data = make_survival_variant_screen(64, 2, seed=5)
        seen = {}

        def factory(d, shard, gtot, **extra):
            seen.update(extra=extra, shard=shard, n=d.n_guides)
            return ExchangeEngine(d, shard, gtot, kind == "tiling", **extra)

        whole, losses = parallel.run_sharded(factory, data, 5, seed=3, report_every=2)
        extra = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in seen["extra"].items()}
        torch.save({"params": whole, "losses": losses, "extra": extra, "shard": seen["shard"], "n": seen["n"]},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["tiling", "survival"])
def test_two_rank_exchange_families_plumbing(tmp_path, kind):
    world = 2
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    if kind == "tiling":
        from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen
        data = make_sorting_tiling_screen(61, 2, seed=5, n_max_alleles=4)
        # guides cut anywhere, every shard sees all edits, rank 0 owns the replicated loss terms
        assert [o["shard"] for o in outs] == [(0, 30, 0, data.n_targets), (30, 61, 0, data.n_targets)]
        assert [o["extra"]["loss_owner"] for o in outs] == [True, False]
    else:
        from bean_amd.preprocessing.synthetic import make_survival_variant_screen
        data = make_survival_variant_screen(64, 2, seed=5)
        want = (data.X[:, 0, :].float() + 1).sum(-1).tolist()
        for o in outs:  # whole-screen t0 totals on every rank, target-aligned cuts
            assert o["extra"]["t0_totals"] == want and "loss_owner" not in o["extra"]
        assert outs[0]["shard"][1] == outs[1]["shard"][0] and outs[1]["shard"][1] == 64
    assert sum(o["n"] for o in outs) == data.n_guides
    # the per-step all-reduce summed both ranks' buffers: X total, guide total, step counter x 2
    steps = 5
    shared = torch.tensor([float(data.X.sum()) * steps, data.n_guides * steps, 2.0 * sum(range(1, steps + 1))],
                          dtype=torch.float64)
    for o in outs:
        assert o["losses"] == [float(data.n_guides)] * steps  # loss windows all-reduced
        assert o["params"]["alpha_pi"].shape == (data.n_guides, 2)  # per-guide: gathered
        np.testing.assert_allclose(o["params"]["alpha_pi"][:, 0].numpy(),
                                   data.a0.float().numpy() * sum(range(1, steps + 1)), rtol=1e-6)
        if kind == "tiling":
            np.testing.assert_allclose(o["params"]["mu_loc"].numpy(), shared.numpy(), rtol=1e-12)  # replicated
        else:
            assert o["params"]["q0"].shape == (data.n_guides,)  # per-guide q0 gathered
            assert o["params"]["mu_loc"].shape == (data.n_targets, 1)  # per-target: gathered by target counts
            np.testing.assert_allclose(o["params"]["mu_loc"].numpy(), float(shared[0]), rtol=1e-12)


def test_order_by_alleles_is_a_stable_permutation_with_screen_indices(monkeypatch):
    from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen, make_sorting_variant_screen

    data = make_sorting_tiling_screen(257, 2, seed=4)
    ordered, ids = parallel.order_by_alleles(data, 1000)
    n = data.allele_mask.sum(1).numpy()
    perm = (ids - 1000).numpy()
    assert sorted(perm.tolist()) == list(range(257))
    assert np.all(np.diff(n[perm]) <= 0)
    for c in np.unique(n):  # stable: screen order inside a class
        assert np.all(np.diff(perm[n[perm] == c]) > 0)
    assert torch.equal(ordered.allele_mask, data.allele_mask[perm])
    assert torch.equal(ordered.X, data.X[:, :, perm])
    # nothing to reorder: variant screens, or switched off
    v = make_sorting_variant_screen(50, 2, seed=1)
    assert parallel.order_by_alleles(v)[1] is None
    monkeypatch.setenv("BEAN_HIP_ORDER_GUIDES", "0")
    assert parallel.order_by_alleles(data)[1] is None
