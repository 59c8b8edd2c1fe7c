import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import bottleneck_spec
from oracle import opennet as O
from flow_amd.sim import FlowSim
for fixed in (True, False):
    bspec = bottleneck_spec(R=2, cap_human=90, cap_rl=10, horizon=150, seed=2, q=3600.0)
    bsim, bora = FlowSim(bspec, "f32"), O.MergeOracle(dict(bspec, cell_sum="fixed") if fixed else bspec, np.float32)
    assert np.array_equal(bsim.reset(), bora.reset().astype(np.float32))
    rng = np.random.default_rng(1)
    bad = None
    for k in range(150):
        a = rng.uniform(-1.0, 1.0, (2, bspec["num_rl"])).astype(np.float32)
        o_gpu, r_gpu, d_gpu = bsim.step(a)
        o_ref, r_ref, d_ref = bora.step(a)
        if not np.array_equal(o_gpu, o_ref.astype(np.float32)):
            idx = np.argwhere(o_gpu != o_ref.astype(np.float32))
            bad = (k, idx[:6].tolist(), [float(o_gpu[tuple(i)]) for i in idx[:3]], [float(o_ref[tuple(i)]) for i in idx[:3]])
            break
    print("fixed" if fixed else "slot-order", bsim.last_kernel, "first mismatch:", bad, "pos equal:", np.array_equal(bsim.pos[bora.alive], bora.x[bora.alive]))
    bsim.close()
