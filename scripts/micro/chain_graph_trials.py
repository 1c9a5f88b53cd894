"""How repeatable is a K-branch graph of independent launch chains?  (two_chains.py, CHAIN_MODE=graph: per trial a
fresh set of engines and a fresh graph; per replay its time.)  Diagnostic.

    python scripts/micro/chain_graph_trials.py [guides] [K] [trials]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bean_amd  # noqa: F401,E402
from bean_amd import engine, parallel  # noqa: E402
from bean_amd.preprocessing import synthetic as syn  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    chunk = 50
    data = syn.make_sorting_variant_screen(G, 5, seed=20240502)
    shards = parallel.plan_shards(data.target_lengths.numpy(), K)
    parts = [parallel.shard_screen(data, sh).to("cuda:0") for sh in shards]
    for trial in range(trials):
        engs = [engine.HipSVI("MixtureNormal", p, num_steps=2000, guide_offset=sh[0], target_offset=sh[2],
                              n_guides_total=G) for p, sh in zip(parts, shards)]

        def eager(n):
            for e in engs:
                e._check(e.lib.bean_hip_svi_resume(e._h, 101, e.steps_done, n, 0, e._sptr()), "svi_resume")
                e.steps_done += n

        for e in engs:
            e.stream.wait_stream(torch.cuda.current_stream())
        eager(20)
        torch.cuda.synchronize()
        main_s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s, capture_error_mode="thread_local"):
            for e in engs:
                e.stream.wait_stream(main_s)
            eager(chunk)
            for e in engs:
                main_s.wait_stream(e.stream)
        times = []
        for _ in range(6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            times.append(round((time.perf_counter() - t) / chunk * 1e6, 1))
        print(trial, "us/step per replay:", times, "streams", [hex(e.stream.cuda_stream)[-5:] for e in engs], flush=True)
        del g
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
