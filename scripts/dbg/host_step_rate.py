"""PCIe-inclusive rate of the host-buffer entry point fs_step (obs / rew / done copied back to host memory every call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from flow_amd.sim import FlowSim
for prec in ("f32", "mixed"):
    sim = FlowSim(bench.c2_spec(4096, seed=1), precision=prec)
    sim.reset()
    for _ in range(50):
        sim.step(None)
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        sim.step(None)
    dt = time.perf_counter() - t0
    print(prec, "fs_step: %.1f us per call, %.3f G env-steps/s (host buffers, 0.74 MB back per call)" % (dt / n * 1e6, 4096 * n / dt / 1e9))
    sim.close()
