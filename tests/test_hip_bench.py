"""GPU: `bench.py` as the driver runs it — the contract line of a SMALL workload (same code path as the headline: the step
replayed as a captured HIP graph, every N-th timed step enqueued eagerly with the HIP-event brackets of the roofline block) and
of the plain eager procedure, in fresh child processes.  Both must report the same loss (same kernels, same inputs), the
roofline blocks, the fused-path counts and the measurement hooks' fields.

What is timed replaces /root/reference/augmented_cyclegan/train.py:190-245 (the reference's loop around
`model.train_instance`, which reports seconds per image)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "7", "--batch", "2", "--size", "128", "--blocks", "2", "--no-cpu-baseline", "--timer-every", "3"]


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + list(extra), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_bench_line_graph_replay_and_eager_agree():
    # the graph path runs two untimed steps behind its capture (the third call): 3 warm-up calls = 5 steps in front of the timed ones
    g, err = _bench("--warmup", "3")
    e, _ = _bench("--warmup", "5", "--no-step-graph")
    assert "capture failed" not in err
    for d in (g, e):
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "roofline_hbm", "fused_paths"):
            assert k in d, k
        assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] in (3, 5) and d["unit"] == "images/s" and d["vs_baseline"] is None
        assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 0.02 * d["value"]
        r = d["roofline"]
        # 3 of the 7 timed steps (0, 3, 6) carry the brackets: 2 generators x 2 passes x 2 blocks x 2 convolutions = 16 forwards each
        assert r["launches_timed"] == 3 * 16 and r["frac"] is not None and 0.0 < r["frac"] < 1.0
        assert r["kernel"].startswith("igemm_conv_x3") and r["peak"] == 833.3 and r["unit"] == "TFLOP/s"
        assert r["passes"]["wgrad"]["split_k_reduce_ms"] > 0.0 and r["passes"]["wgrad"]["with_split_k_reduce_ms"] > r["passes"]["wgrad"]["avg_launch_ms"]
        assert r["peak_sustained"] is not None and 300.0 < r["peak_sustained"] < 834.0          # live probe: 0.9 .. 2.5 PFLOP/s executed
        assert d["roofline_hbm"]["launches_timed"] == 3 * 4 and d["roofline_hbm"]["bound"] == "hbm"
        assert d["config"]["world_size_seen"] == 1 and "backend" in d["config"]
        assert d["fused_paths"]["per_step"]["wgrad_s16"] == 16.0
    assert "captured HIP graph" in g["config"]["launch"] and e["config"]["launch"] == "eager"
    assert g["config"]["loss_G_A"] == e["config"]["loss_G_A"]      # the replay runs the same launches on the same inputs
