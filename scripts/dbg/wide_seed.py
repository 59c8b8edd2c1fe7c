"""Replays tests/test_wide_gpu.py's fuzz case at one seed and reports the first step at which the kernel and the oracle part."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from helpers import bottleneck_spec
from oracle import opennet as O
from test_open_gpu import make

seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
R = int(rng.integers(1, 5)); cap_rl = int(rng.integers(2, 30)); N = int(rng.integers(65, 257))
spec = bottleneck_spec(R=R, cap_human=N - cap_rl, cap_rl=cap_rl, horizon=int(rng.integers(150, 420)), seed=seed,
                       q=float(rng.choice([2300, 3600, 5000])), av_frac=float(rng.choice([0.1, 0.3])),
                       zipper_distance=float(rng.choice([0.0, 20.0, 50.0, 120.0])),
                       warmup_steps=int(rng.choice([0, 0, 20])), lane_change_cooldown_steps=int(rng.choice([2, 8, 20])),
                       lane_change_min_gain=float(rng.choice([3.0, 10.0])), crash_gap=float(rng.choice([0.0, 1.0])),
                       track_followers=bool(rng.integers(0, 2)), sims_per_step=int(rng.choice([1, 1, 2])))
lc = bool(rng.integers(0, 2))
if lc:
    for v in spec["vehicles"][:N - cap_rl]:
        v["lane_change_mode"] = 1621
A = spec["num_rl"]
r = np.random.default_rng(seed)
print("R", R, "N", N, "lc", lc, {k: spec[k] for k in ("zipper_distance", "warmup_steps", "sims_per_step", "crash_gap", "track_followers")}, flush=True)
sim = make(spec, "f32"); ora = O.MergeOracle(spec, np.float32)
og = sim.reset(); orf = ora.reset().astype(np.float32)
print("reset equal", np.array_equal(og, orf), "pos equal", np.array_equal(sim.pos, ora.x))
for k in range(int(spec["horizon"])):
    a = r.uniform(-1.5, 1.5, (R, A)).astype(np.float32)
    o_gpu, r_gpu, d_gpu = sim.step(a); o_ref, r_ref, d_ref = ora.step(a)
    o_ref = o_ref.astype(np.float32)
    al = ora.alive
    st_eq = np.array_equal(sim.pos[al], ora.x[al]) and np.array_equal(sim.lane[al], ora.route[al]) if hasattr(sim, "lane") else np.array_equal(sim.pos[al], ora.x[al])
    if not np.array_equal(o_gpu, o_ref) or not st_eq:
        bad = np.argwhere(o_gpu != o_ref)
        print("step", k, "obs mismatches", bad[:8].tolist(), "gpu", o_gpu[o_gpu != o_ref][:6], "ref", o_ref[o_gpu != o_ref][:6])
        dx = np.argwhere((sim.pos != ora.x) & al)
        print("pos mismatches", dx[:8].tolist(), "kernel", sim.last_kernel)
        for (rr, i) in dx[:4]:
            print("  slot", rr, i, "gpu x", sim.pos[rr, i], "ref x", ora.x[rr, i], "ref route", ora.route[rr, i], "v", sim.vel[rr, i], ora.v[rr, i])
        C = len(spec["obs_cells"])
        for (rr, c) in bad[:4]:
            print("  obs col", c, "block", c // C, "cell", c % C, spec["obs_cells"][c % C] if c < 4 * C else "outflow")
        break
else:
    print("all steps equal")
