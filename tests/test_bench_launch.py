"""``bench.py --gpus N`` must be startable by the bare command the driver uses: without a launcher
around it, it starts its own ranks (child ``torch.distributed.run``) before touching a GPU and
relays rank 0's single JSON line.  Here (no GPU) the ranks run the ``dry`` rehearsal: rendezvous
over gloo on 127.0.0.1, no measurement; the GPU version of this test times real legs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                         text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, lines


@pytest.mark.timeout(300)
def test_bench_gpus2_starts_its_own_ranks_and_prints_one_json_line():
    res, lines = _run({"BEAN_BENCH_REHEARSAL": "dry"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["strong"]["guides"] == 500_000  # BASELINE configs[3]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"):
        assert key in out


@pytest.mark.timeout(300)
def test_bench_strong_legs_exist_for_survival_and_tiling():
    for cfg, guides in (("survival", 100_000), ("tiling", 50_000)):
        res, lines = _run({"BEAN_BENCH_REHEARSAL": "dry"}, "--gpus", "2", "--config", cfg, "--scaling", "strong",
                          "--steps", "2", "--warmup", "0")
        assert res.returncode == 0, res.stderr[-2000:]
        out = json.loads(lines[0])
        assert out["scaling"] == "strong" and out["strong"]["guides"] == guides


def test_bench_world_size_mismatch_is_an_error():
    res, _ = _run({"RANK": "0", "WORLD_SIZE": "1", "BEAN_BENCH_REHEARSAL": "dry"}, "--gpus", "2")
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_times_weak_and_strong_legs():
    """Two ranks on cuda:0 over gloo (BEAN_BENCH_REHEARSAL=1), started by bench.py itself: one JSON
    line with the weak value and the strong object of one screen cut in two."""
    res, lines = _run({"BEAN_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--steps", "30", "--warmup", "5", "--guides", "4000",
                      "--strong-guides", "8000", "--no-cpu-baseline")
    assert res.returncode == 0, res.stderr[-3000:]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    s = out["strong"]
    assert s["guides"] == 8000 and 3900 <= s["guides_this_rank"] <= 4100 and s["value"] > 0
    assert out["roofline"]["kernel"] == "k_guide_wave2" and out["roofline"]["kernel_ms"] > 0
    # exchange families: every step all-reduces inside the step
    res, lines = _run({"BEAN_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--config", "survival", "--steps", "20", "--warmup",
                      "3", "--guides", "3000", "--strong-guides", "6000", "--no-cpu-baseline")
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(lines[0])
    assert "all-reduce gsum" in out["strong"]["exchange"] and out["strong"]["value"] > 0
    # tiling: every rank orders its shard's guides by allele count; the per-edit gradients are exchanged
    res, lines = _run({"BEAN_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--config", "tiling", "--steps", "20", "--warmup",
                      "3", "--guides", "3000", "--strong-guides", "6000", "--no-cpu-baseline")
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(lines[0])
    assert "all-reduce tgrad" in out["strong"]["exchange"] and out["strong"]["value"] > 0
    assert "ordered by allele count" in out["config"]["workload"]
