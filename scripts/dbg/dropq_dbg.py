"""Debug aid: the first step at which k_drop_queue and the oracle differ on a fixed-lane inflow configuration."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import bottleneck_spec
from oracle import opennet as O
from flow_amd.sim import FlowSim
from flow_amd import _lib as L

spec = bottleneck_spec(R=3, cap_human=56, cap_rl=8, horizon=300, seed=11)
fl = spec["inflows"]
spec["inflows"] = [dict(fl[0], route=1, period=3.0), dict(fl[1], route=1, period=7.0), dict(fl[0], route=2, period=2.5),
                   dict(fl[0], route=-1, period=4.0)]
ora = O.MergeOracle(dict(spec, cell_sum="fixed"), np.float32)
sim = FlowSim(spec, precision="f32")
sim.reset(); ora.reset()
rng = np.random.default_rng(4)
for k in range(30):
    a = rng.uniform(-1, 1, (3, spec["num_rl"])).astype(np.float32)
    o_ref, r_ref, d_ref = ora.step(a)
    o_gpu, r_gpu, d_gpu = sim.step(a)
    route = sim.get_state(L.FS_FIELD_ROUTE)
    bad = np.argwhere(o_gpu != o_ref.astype(np.float32))
    if len(bad) or (route != ora.route).any():
        print("step", k, sim.last_kernel, "obs mismatches", bad[:12].tolist())
        for r in sorted(set(bad[:, 0].tolist()) | set(np.argwhere(route != ora.route)[:, 0].tolist())):
            al_g, al_o = np.flatnonzero(route[r] >= 0), np.flatnonzero(ora.route[r] >= 0)
            print(" replica", r, "gpu slots", al_g.tolist(), "route", route[r][al_g].tolist(), "x", sim.pos[r][al_g].round(3).tolist())
            print("          ora slots", al_o.tolist(), "route", ora.route[r][al_o].tolist(), "x", ora.x[r][al_o].round(3).tolist())
            print("  counters gpu", sim.get_state(L.FS_FIELD_COUNTERS)[r].tolist(), "ora dep/arr/drop", ora.total_departed[r], ora.total_arrived[r], ora.total_dropped[r], "seq", ora.seq_ctr[r])
        break
else:
    print("no mismatch in 30 steps")
