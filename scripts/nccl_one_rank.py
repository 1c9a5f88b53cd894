"""One-rank RCCL run of the guide-sharded fit paths (launched by tests/test_gpu_nccl.py).

`init_process_group("nccl", world_size=1)`, then `parallel.run_sharded` for (a) the variant sorting
MixtureNormal family (no data-path exchange: loss all-reduce per window + final all-gather) and (b) one
exchange family, tiling MultiMixtureNormal (`HipSVI.run_exchanged`: one all-reduce of the per-edit
gradients per step on the engine's stream, next to the HIP kernels) and (c) survival MixtureNormal
(normaliser exchange).  Each must equal the fit without a process group.  This is the first time RCCL,
its stream interplay with the engine's stream / graph replay and `all_gather_rows` run on hardware
before an 8-GPU run; with one rank every collective is a copy.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bean_amd  # noqa: F401,E402
from bean_amd import engine, parallel  # noqa: E402
from bean_amd.preprocessing import synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    steps = 130  # two report windows: 100 + 30

    def fit(family, data, **kw):
        def factory(shard_data, shard, n_total, **extra):
            return engine.HipSVI(family, shard_data.to(dev), num_steps=steps, device=dev, guide_offset=shard[0],
                                 target_offset=shard[2], n_guides_total=n_total, **kw, **extra)

        whole, losses = parallel.run_sharded(factory, data, steps, seed=101)
        ref = engine.HipSVI(family, data.to(dev), num_steps=steps, device=dev, **kw)
        ref.run(steps, seed=101)
        want, want_losses = ref.constrained(), ref.losses()
        ref.close()
        return whole, losses, want, want_losses

    # (a) no data-path exchange
    data = syn.make_sorting_variant_screen(6000, 3, seed=5, with_accessibility=True)
    whole, losses, want, want_losses = fit("MixtureNormal", data, scale_by_accessibility=True)
    for k in want:
        assert torch.equal(whole[k], want[k]), k
    assert losses == want_losses, "loss history differs"
    # the exchange families step through the Python loop + dist.all_reduce by default and through the
    # library-owned RCCL communicator with BEAN_HIP_NATIVE_COMM=1 (parallel.native_comm_enabled): both
    for native in ("0", "1"):
        os.environ["BEAN_HIP_NATIVE_COMM"] = native
        # (b) per-step exchange: tiling per-edit gradients
        data = syn.make_sorting_tiling_screen(3000, 3, seed=6)
        whole, losses, want, want_losses = fit("MultiMixtureNormal", data)
        for k in want:
            assert torch.allclose(whole[k], want[k], rtol=1e-5, atol=1e-7), (native, k)
        assert max(abs(a - b) / abs(b) for a, b in zip(losses, want_losses)) < 1e-9, native
        # (c) per-step exchange: survival normalisers
        data = syn.make_survival_variant_screen(4000, 3, seed=7)
        whole, losses, want, want_losses = fit("MixtureNormal", data)
        for k in want:
            assert torch.allclose(whole[k], want[k], rtol=1e-5, atol=1e-7), (native, k)
        assert max(abs(a - b) / abs(b) for a, b in zip(losses, want_losses)) < 1e-9, native
    os.environ.pop("BEAN_HIP_NATIVE_COMM", None)
    # the collective itself, on the engine-style side stream
    s = torch.cuda.Stream(device=dev)
    t = torch.arange(1000, dtype=torch.float64, device=dev)
    with torch.cuda.stream(s):
        dist.all_reduce(t)
    s.synchronize()
    assert float(t.sum()) == 999 * 1000 / 2
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_ONE_RANK_OK")


if __name__ == "__main__":
    main()
