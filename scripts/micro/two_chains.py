"""Does one GPU step faster when the screen is fitted as K independent target-aligned parts on K streams?
(variant sorting families share no parameter across targets: the parts are what `torchrun` ranks would hold, here
all on one device - one part's latency-bound k_param under another part's guide kernel.)  Diagnostic.

    python scripts/micro/two_chains.py [guides] [steps]
"""
import ctypes
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
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    data = syn.make_sorting_variant_screen(G, 5, seed=20240502)
    out = {}
    ref_loss = None
    ks = [int(k) for k in os.environ.get("CHAINS", "1,2,3,4,8").split(",")]
    for K in ks:
        shards = parallel.plan_shards(data.target_lengths.numpy(), K)
        engs = []
        for sh in shards:
            part = parallel.shard_screen(data, sh).to("cuda:0")
            engs.append(engine.HipSVI("MixtureNormal", part, num_steps=steps + 400, guide_offset=sh[0],
                                      target_offset=sh[2], n_guides_total=G))

        def enqueue(n):
            # straight to the library: HipSVI.run orders its stream against the current one on both sides,
            # which would put the parts one after another
            for e in engs:
                e._check(e.lib.bean_hip_svi_resume(e._h, 101, e.steps_done, n, 50, e._sptr()), "svi_resume")
                e.steps_done += n

        for e in engs:
            e.stream.wait_stream(torch.cuda.current_stream())
        if os.environ.get("CHAIN_MODE", "streams") == "graph":
            # ONE graph with K parallel branches (captured from K streams that fork from and join the capturing
            # one): which hardware queue a branch runs on is then the graph executor's choice, made once
            def eager(n):
                for e in engs:
                    e._check(e.lib.bean_hip_svi_resume(e._h, 101, e.steps_done, n, 0, e._sptr()), "svi_resume")
                    e.steps_done += n

            eager(100)
            torch.cuda.synchronize()
            main = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            chunk = 50
            with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
                for e in engs:
                    e.stream.wait_stream(main)
                eager(chunk)
                for e in engs:
                    main.wait_stream(e.stream)
            g.replay()
            torch.cuda.synchronize()
            reps = steps // chunk
            t = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * steps / (reps * chunk)
            for e in engs:
                e.steps_done += chunk * reps  # (the device counters moved on with every replay)
        else:
            enqueue(100)
            torch.cuda.synchronize()
            t = time.perf_counter()
            enqueue(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        loss = sum(float(e.loss_hist[e.steps_done - 1]) for e in engs)
        if ref_loss is None:
            ref_loss = loss
        out[f"{K}_{len(out)}"] = {"us_per_step": round(dt / steps * 1e6, 2), "loss_last": loss, "loss_rel": abs(loss - ref_loss) / abs(ref_loss),
                  "guides": [s[1] - s[0] for s in shards]}
        print(K, json.dumps(list(out.values())[-1]), flush=True)
        for e in engs:
            e.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "two_chains.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
