"""float32 against float64 on the open networks (no noise): the same merge / lane-drop configuration stepped by both kernels;
how long the discrete events (departures, arrivals) agree and how far positions are apart while they do."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from helpers import bottleneck_spec, merge_spec
from oracle import opennet as O
from flow_amd import _lib as L
from flow_amd.sim import FlowSim

for name, spec, K in (("merge (C5-like, 64 slots)", merge_spec(R=64, cap_human=56, cap_rl=8, num_rl=8, horizon=600, seed=32, env=O.ENV_MERGE_MA), 600),
                      ("lane drop (C4-like, 256 slots)", bottleneck_spec(R=16, cap_human=230, cap_rl=26, horizon=1000, seed=31), 600)):
    spec["vehicles"] = [dict(v, noise=0.0) for v in spec["vehicles"]]
    a, b = FlowSim(spec, "f32"), FlowSim(spec, "f64")
    a.reset(), b.reset()
    acts = np.zeros((spec["num_replicas"], max(a.act_dim, 1)), np.float32)[:, :a.act_dim]
    first_split = np.full(spec["num_replicas"], -1)
    worst = 0.0
    for k in range(K):
        a.step(acts if a.act_dim else None), b.step(acts if a.act_dim else None)
        ra, rb = a.get_state(L.FS_FIELD_ROUTE), b.get_state(L.FS_FIELD_ROUTE)
        same = (ra == rb).all(axis=1)
        first_split[(first_split < 0) & ~same] = k
        ok = first_split < 0
        if ok.any():
            alive = (ra >= 0) & ok[:, None]
            worst = max(worst, float(np.abs(a.pos.astype(np.float64) - b.pos)[alive].max()) if alive.any() else 0.0)
    print("%s: replicas whose departures / arrivals / lanes agree for all %d steps: %d of %d (first split at step: %s); max |dx| while they agree: %.2e m"
          % (name, K, int((first_split < 0).sum()), spec["num_replicas"], sorted(set(first_split[first_split >= 0].tolist()))[:6], worst))
    a.close(), b.close()
