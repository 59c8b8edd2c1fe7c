"""bench.py's N > 1 code path on a one-GPU box: BENCH_FORCE_DIST=1 takes RCCL init, the per-fragment observation
all-gather, the MAX over ranks and the teardown with a world of one rank -- the 8-GPU run itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout: %r" % lines[:3]
    return json.loads(lines[0])


def test_forced_distributed_path_with_one_rank():
    out = run_bench({"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"},
                    "--gpus", "1", "--steps", "3000", "--warmup", "1500", "--no-extras")
    assert out["n_gpus"] == 1 and out["steps"] == 3000 and out["warmup"] == 1500
    assert out["metric"] == "env-steps/sec" and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["gather_check"] == {"gathered_rows": 4096, "rank_blocks_equal_local": True}
    assert out["roofline"]["frac"] > 0.05 and out["parity"]["within_1e-4"] is True
    assert out["value"] > 1e9 and abs(out["value"] - 4096 * 3000 / (out["ms_per_step"] * 3000 * 1e-3)) < 1e-3 * out["value"]
    assert out["cpu_baseline"] is None and "rollout_1500" not in out          # --no-extras
    assert out["fragment_latency"]["steps"] == 1500


def test_single_gpu_line_has_no_gather_and_strong_scaling_is_the_same_workload_at_one_gpu():
    out = run_bench({}, "--gpus", "1", "--steps", "1500", "--warmup", "1500", "--no-extras", "--scaling", "strong")
    assert out["gather_check"] is None and out["scaling"] == "strong" and out["n_gpus"] == 1
    assert out["config"]["replicas_total"] == 4096 and out["config"]["replicas_per_gpu"] == 4096
