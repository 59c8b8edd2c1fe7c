"""Where a sub-step of k_steps_open goes (C5 merge leg): builds a -DFS_PHASE_TIMERS variant of the seg64_f32 object,
runs one episode with it and prints the share of each section.  Run on the GPU box:

    python scripts/phase_open.py build      # here (CPU): compiles flow_amd/libflowsim_timers.so, which travels with gpurun
    FLOWSIM_LIB=flow_amd/libflowsim_timers.so python scripts/phase_open.py [replicas]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "flow_amd", "libflowsim_timers.so")
SECTIONS = ["controllers", "integration + right of way", "move + arrivals", "inflows", "neighbours + crash",
            "  of which full structure evaluations", "(structure evaluations, count)", "observation + reward"]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        from flow_amd import build
        print(build.build_variant(LIB, ["-DFS_PHASE_TIMERS"], names=["seg64_f32", "wide4_f32"]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "buildfine":
        from flow_amd import build
        print(build.build_variant(LIB.replace("timers", "timers2"), ["-DFS_PHASE_TIMERS=2"], names=["wide4_f32"]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("c4", "c4fine"):          # the wide kernel (C4): its own sections
        import numpy as np
        import torch
        import bench
        import flow_amd.sim as simmod
        from flow_amd import _lib as L
        keep = {}
        real_close = simmod.FlowSim.close

        def close(self):
            keep["cnt"] = self.get_state(L.FS_FIELD_COUNTERS).astype(np.float64)
            real_close(self)
        simmod.FlowSim.close = close
        # (lane changing on -- flow/benchmarks/bottleneck1 -- keeps the leg on k_steps_wide; off, it runs on k_drop_queue)
        res = bench.c4_leg(torch.device("cuda", 0), R=int(sys.argv[2]) if len(sys.argv) > 2 else 128,
                           lane_change_mode=int(os.environ.get("FLOWSIM_C4_LC", "1621")))
        mean = keep["cnt"].mean(axis=0) * 64.0 / res["env_steps"]
        names = ["-", "actions + integration + arbitration + move", "inflows", "neighbours: publish + rank",
                 "neighbours: masks", "neighbours: leader (+ lane-change wishes)", "neighbours tail + controllers",
                 "observation + reward"]
        first = 1
        if sys.argv[1] == "c4fine":
            names = ["outside the update and the observation", "update: first barrier", "update: old places, barrier, window",
                     "update: proof, scatter, flag barrier (+ count)", "update: masks", "update: leader search",
                     "update tail + cell lookup", "cell collection, stores, reward"]
            first = 0
        out = {"env_steps_per_s": res["value"], "cycles_per_sub_step (wave 0)": mean[first:].sum()}
        for q in range(first, 8):
            out[names[q]] = {"cycles_per_sub_step": mean[q], "share": mean[q] / mean[first:].sum()}
        print(json.dumps(out, indent=1))
        sys.exit(0)
    assert os.environ.get("FLOWSIM_LIB"), "run with FLOWSIM_LIB=flow_amd/libflowsim_timers.so"
    import numpy as np
    import torch
    import bench
    from flow_amd import _lib as L
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    keep = {}
    orig = L.load().fs_get_state

    import flow_amd.sim as simmod
    real_close = simmod.FlowSim.close

    def close(self):                      # the leg closes its env: grab the counters first
        keep["cnt"] = self.get_state(L.FS_FIELD_COUNTERS).astype(np.float64)
        real_close(self)
    simmod.FlowSim.close = close
    res = bench.c5_leg(torch.device("cuda", 0), R=R)
    cnt = keep["cnt"] * 64.0                               # cycles per wave over the 3000 sub-steps of the episode
    substeps = res["env_steps"] * res["sims_per_step"]
    mean = cnt.mean(axis=0)
    total = mean[[0, 1, 2, 3, 4, 7]].sum()
    out = {"sub_steps_per_s": res["value"], "cycles_per_sub_step": total / substeps,
           "structure_evaluations_per_sub_step": mean[6] / 64.0 / substeps}
    for q, name in enumerate(SECTIONS):
        if q == 6:
            continue
        out[name] = {"cycles_per_sub_step": mean[q] / substeps, "share": mean[q] / total}
    print(json.dumps(out, indent=1))
