import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from helpers import bottleneck_spec
from oracle import opennet as O
from flow_amd.sim import FlowSim
from flow_amd import _lib as L
C = 35
for prec, ch, cr, steps in (("f64", 20, 124, 30), ("f64", 130, 14, 300), ("f64", 100, 28, 300)):
    spec = bottleneck_spec(R=1, cap_human=ch, cap_rl=cr, horizon=400, seed=4)
    sim = FlowSim(spec, precision=prec)
    ora = O.MergeOracle(spec, np.float64 if prec == "f64" else np.float32)
    o = sim.reset(); oo = ora.reset()
    for k in range(steps):
        o, r, d = sim.step(None); oo, rr, dd = ora.step(None)
    bad = np.nonzero(np.abs(o[0] - oo[0]) > 1e-6)[0]
    print(prec, ch, cr, "N", ch + cr, "bad idx", bad, "kind", bad // C, "cell", bad % C)
    print("  gpu", o[0][bad], "\n  ora", oo[0][bad].astype(np.float32))
    al = np.nonzero(ora.alive[0])[0]
    print("  alive slots", al, " pos ok", np.allclose(sim.pos[0][al], ora.x[0][al]))
    sim.close()
