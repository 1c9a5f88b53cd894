"""Tiling step time with the guides in screen order and ordered by their number of alleles (so that the lanes of
a wave have their alleles in the same slots).  Diagnostic.

    [BEAN_HIP_LIB=path] python scripts/micro/tiling_sorted.py [guides] [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bean_amd  # noqa: F401,E402
from bean_amd import engine  # noqa: E402
from bean_amd.preprocessing import synthetic as syn  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    # TILING_AMAX / TILING_AMEAN: allele slots per guide and mean edited alleles (defaults: BASELINE config 3's)
    amax, amean = int(os.environ.get("TILING_AMAX", "8")), float(os.environ.get("TILING_AMEAN", "3.0"))
    data = syn.make_sorting_tiling_screen(G, 5, seed=20240503, n_max_alleles=amax, alleles_mean=amean)
    n_al = data.allele_mask.sum(1).numpy()
    out = {}
    for name in os.environ.get("ORDERS", "screen,sorted,screen,sorted").split(","):
        d, kw = data, {}
        if name == "sorted":
            d = data[np.argsort(-n_al, kind="stable")]
        if name == "ordered":  # as run_inference does it: with the screen indices that key the random streams
            from bean_amd import parallel

            d, ids = parallel.order_by_alleles(data)
            kw = dict(guide_ids=ids)
        eng = engine.HipSVI("MultiMixtureNormal", d.to("cuda:0"), num_steps=steps + 100, **kw)
        eng.run(50)
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        eng.set_profile(1)
        eng.run(20)
        torch.cuda.synchronize()
        k_ms, n = eng.get_profile()
        out[name] = {"us_per_step": round(dt / steps * 1e6, 1), "kernel": eng.dominant_kernel, "kernel_us": round(k_ms * 1e3, 1),
                     "loss_last": float(eng.loss_hist[eng.steps_done - 1])}
        print(name, json.dumps(out[name]), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
