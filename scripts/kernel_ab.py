"""A/B timing of libbean_hip.so variants on one GPU (diagnostic; not part of the product path).

    python scripts/kernel_ab.py [--guides 50000,62500] [--family MixtureNormal] name=path[,ENV=VAL...] ...

Each variant runs in its own process (BEAN_HIP_LIB selects the library): per screen size the dominant
kernel's mean duration (HIP events with the kernel's own timestamps) and the graph-replayed step time.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(guides, family, steps):
    import torch

    import bean_amd  # noqa: F401
    from bean_amd import engine
    from bean_amd.preprocessing import synthetic as syn

    out = {}
    for G in guides:
        if family == "survival":
            data = syn.make_survival_variant_screen(G, 3, seed=20240506).to("cuda:0")
            fam = "MixtureNormal"
        elif family == "tiling":
            data = syn.make_sorting_tiling_screen(G, 5, seed=20240503).to("cuda:0")
            fam = "MultiMixtureNormal"
        else:
            data = syn.make_sorting_variant_screen(G, 5, seed=20240502).to("cuda:0")
            fam = family
        eng = engine.HipSVI(fam, data, num_steps=steps + 100)
        eng.run(50)
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        losses = eng.losses()
        eng.close()
        prof = engine.HipSVI(fam, data, num_steps=100)
        prof.set_profile(True)
        prof.run(30, graph_chunk=0)
        torch.cuda.synchronize()
        ms, n = prof.get_profile()
        name = prof.dominant_kernel
        prof.close()
        prof = engine.HipSVI(fam, data, num_steps=100)
        prof.set_profile(2)
        prof.run(30, graph_chunk=0)
        torch.cuda.synchronize()
        pms, _ = prof.get_profile()
        prof.close()
        out[str(G)] = {"kernel": name, "kernel_us": round(ms * 1e3, 2), "param_us": round(pms * 1e3, 2),
                       "step_us": round(dt / steps * 1e6, 2), "loss_last": losses[-1]}
    print("AB_RESULT " + json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    guides, family, steps = [50000], "MixtureNormal", 300
    specs = []
    i = 0
    while i < len(args):
        if args[i] == "--guides":
            guides = [int(x) for x in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--family":
            family = args[i + 1]
            i += 2
        elif args[i] == "--steps":
            steps = int(args[i + 1])
            i += 2
        elif args[i] == "--child":
            return child(guides, family, steps)
        else:
            specs.append(args[i])
            i += 1
    results = {}
    for spec in specs:
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        env = dict(os.environ)
        if parts[0]:
            env["BEAN_HIP_LIB"] = os.path.join(ROOT, parts[0]) if not os.path.isabs(parts[0]) else parts[0]
        for kv in parts[1:]:
            k, v = kv.split("=", 1)
            env[k] = v
        cmd = [sys.executable, os.path.abspath(__file__), "--guides", ",".join(map(str, guides)), "--family", family,
               "--steps", str(steps), "--child"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("AB_RESULT ")]
        results[name] = json.loads(line[0][len("AB_RESULT "):]) if line else {"error": res.stderr[-800:]}
        print(name, json.dumps(results[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = os.environ.get("AB_TAG", "ab")
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}.json"), "w") as fh:
        json.dump(results, fh, indent=1)


if __name__ == "__main__":
    main()
