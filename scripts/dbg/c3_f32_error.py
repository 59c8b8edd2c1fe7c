import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
from helpers import figure_eight_spec, idm_vehicle
from oracle import refsim as S
from flow_amd.sim import FlowSim
R, N, K = 64, 14, 1500
spec = figure_eight_spec(R=R, N=N, horizon=K, seed=5, num_rl=1)
veh = [idm_vehicle(speed_mode=1, max_decel=1.5, noise=0.0) for _ in range(N - 1)]
veh.append(idm_vehicle(controller=S.CTRL_RL, rl_index=0, speed_mode=1))
spec["vehicles"] = veh
acts = np.random.default_rng(1).uniform(-1, 1, (K, R, 1)).astype(np.float32)
res = {}
for prec in ("f32", "f64"):
    sim = FlowSim(spec, prec)
    sim.reset()
    for k in range(K):
        sim.step(acts[k])
        if prec == "f32" and k in (199, 499, 999, 1499): res[("f32", k)] = (sim.pos.astype(np.float64).copy(), sim.vel.astype(np.float64).copy())
        if prec == "f64" and k in (199, 499, 999, 1499): res[("f64", k)] = (sim.pos.copy(), sim.vel.copy())
    print(prec, sim.last_kernel)
    sim.close()
for k in (199, 499, 999, 1499):
    a, b = res[("f32", k)], res[("f64", k)]
    L = float(np.asarray(spec["ring_length"])[0]) if "ring_length" in spec else 0
    dx = np.abs(a[0] - b[0]); 
    print(k + 1, "max |dx|", np.median(dx.max(axis=1)), dx.max(), "max |dv|", np.abs(a[1] - b[1]).max())
