"""The queue formulation of the open-network neighbour rules (oracle/queuenet.py: what the HIP kernel k_merge_queue keeps
instead of comparing pairs) equals the all-pairs statement of oracle/opennet.py -- leader, headway, sticky follower --
at every sub-step, through insertions, arrivals, merges and collisions (re-sorts).  CPU only."""
import numpy as np
import pytest

from helpers import merge_spec
from oracle import opennet as O
from oracle.queuenet import QueueMergeOracle


def quiet(spec):
    spec = dict(spec)
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    return spec


@pytest.mark.parametrize("seed", [1, 2, 7, 8, 11])
@pytest.mark.parametrize("env", [O.ENV_MERGE_MA, O.ENV_MERGE_PO])
def test_queue_structure_gives_the_all_pairs_neighbours(seed, env):
    kw = dict(R=3, cap_human=20 + seed % 5, cap_rl=4, num_rl=2, horizon=200, seed=seed, env=env,
              q_highway=1500 + 100 * seed, q_merge=200 + 80 * seed, sims_per_step=1 + seed % 3)
    if seed % 4 == 3:
        kw["sim_step"] = 0.5
    spec = quiet(merge_spec(**kw))
    if seed % 3 == 2:                                   # "aggressive": nobody obeys SUMO's safe speed -> collisions
        spec["vehicles"] = [dict(v, speed_mode=0) for v in spec["vehicles"]]
    q = QueueMergeOracle(spec, np.float32)
    q.reset()
    rng = np.random.default_rng(seed)
    for _ in range(200):
        q.step(rng.uniform(-1.0, 1.5, (3, spec["num_rl"])).astype(np.float32))      # (every sub-step asserts)
    assert q.checks >= 3 * 200 and q.joins > 0 and q.total_arrived.sum() > 0
    if seed in (8, 11) and env == O.ENV_MERGE_MA:
        assert q.resorts > 0                            # the collision runs went through the re-sort
