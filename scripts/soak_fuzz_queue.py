"""One-off soak of the queue-order kernels (k_merge_queue, k_drop_queue) and of BottleneckAccelEnv's host path with random
configurations beyond the pinned test cases, each against the float32 oracle bit for bit:

    python scripts/soak_fuzz_queue.py [first seed] [count]
"""
import os
import sys
import traceback

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))


def merge_case(seed):
    import test_queue_gpu as tq
    from test_open_gpu import quiet, run_pair
    rng = np.random.default_rng(9000 + seed)
    spec = tq.ma_spec(R=int(rng.integers(1, 6)), cap_human=int(rng.integers(10, 50)), cap_rl=int(rng.integers(1, 9)),
                      num_rl=8, horizon=260, seed=seed, pre=float(rng.choice([150.0, 300.0, 500.0])),
                      q_highway=float(rng.integers(800, 2600)), q_rl=float(rng.integers(50, 400)),
                      q_merge=float(rng.integers(50, 700)), sims_per_step=int(rng.integers(1, 6)),
                      sim_step=float(rng.choice([0.1, 0.2, 0.5])), ma_apply_actions=bool(rng.integers(0, 2)))
    spec["num_rl"] = sum(1 for v in spec["vehicles"] if v["controller"] == 1)
    if rng.integers(0, 3) == 0:
        spec["vehicles"] = [dict(v, speed_mode=0) for v in spec["vehicles"]]         # collisions, re-sorts
    if rng.integers(0, 2):
        spec = quiet(spec)
    else:
        spec["noise_math"] = "exact"                                                 # noisy, and still oracle-exact
    R, A = spec["num_replicas"], spec["num_rl"]
    ora = run_pair(spec, "f32", 260, tq.nan_actions(R, A, seed), check_every=13)
    return "merge R=%d N=%d arrived>=%d" % (R, spec["num_vehicles"], int(ora.total_arrived.min()))


def drop_case(seed):
    import test_dropq_gpu as td
    from helpers import bottleneck_spec
    from oracle import opennet as O
    rng = np.random.default_rng(9500 + seed)
    dv = bool(rng.integers(0, 4))
    spec = bottleneck_spec(R=int(rng.integers(1, 4)), cap_human=int(rng.integers(36, 200)), cap_rl=int(rng.integers(2, 40)),
                           horizon=260, seed=seed, q=float(rng.integers(1500, 3800)),
                           zipper_distance=float(rng.choice([0.0, 25.0, 50.0, 80.0])),
                           sims_per_step=int(rng.integers(1, 4)), **({} if dv else {"env": O.ENV_BOTTLENECK}))
    ora = td.run(spec, 260, td.actions(spec, seed) if dv else None, check_every=13)
    return "drop R=%d N=%d max on a path %d" % (spec["num_replicas"], spec["num_vehicles"],
                                               int(max((ora.route[r] == p).sum() for r in range(ora.R) for p in range(4))))


def wide_case(seed):
    """k_steps_wide (more than 64 slots; lane changing on in half of the cases): tests/test_wide_gpu.py's fuzz case at other seeds."""
    import test_wide_gpu as tw
    os.environ["FLOWSIM_NO_QUEUE"] = "1"        # (without lane changing the case would run on k_drop_queue: drop_case's job)
    try:
        tw.test_wide_fuzz_random_lane_drop_configs_bit_exact(100 + seed)
    finally:
        del os.environ["FLOWSIM_NO_QUEUE"]
    return "wide seed %d" % (100 + seed)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = []
    for seed in range(first, first + count):
        cases = (("merge", merge_case), ("drop", drop_case), ("wide", wide_case))
        which = os.environ.get("SOAK_CASES")
        for name, fn in (c for c in cases if not which or c[0] in which.split(",")):
            try:
                print("seed", seed, fn(seed), flush=True)
            except Exception:
                bad.append((name, seed))
                traceback.print_exc()
    print("FAILED:" if bad else "all ok", bad)
    sys.exit(1 if bad else 0)
