"""Open-loop rate of the reference's multi-agent ring / figure-eight experiments (4096 replicas, 1500-step launches with an
action tape [K, R, num_rl]) on their rollout kernels, and forced onto the generic kernel.  Prints one JSON line each."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def leg(name, R=4096, K=1500, launches=10, generic=False):
    import numpy as np
    import torch
    import flow_amd
    from flow_amd.envs import VecFlowEnv
    flow_amd.install_as_flow()
    fp = importlib.import_module("exp_configs.rl.multiagent." + name).flow_params
    if generic:
        os.environ["FLOWSIM_FORCE_GENERIC"] = "1"
    vec = VecFlowEnv(fp, num_replicas=R, device=0)
    os.environ.pop("FLOWSIM_FORCE_GENERIC", None)
    dev = vec.device
    acts = (torch.rand((K, R, vec.act_dim), device=dev) * 2 - 1) * 0.5
    out = (torch.empty((K, R, vec.obs_dim), device=dev), torch.empty((K, R), device=dev),
           torch.empty((K, R), dtype=torch.uint8, device=dev))
    vec.reset()
    vec.rollout(K, actions=acts, out=out)
    torch.cuda.synchronize()
    ms = []
    for _ in range(launches):
        vec.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vec.rollout(K, actions=acts, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    res = {"experiment": name, "kernel": vec.sim.last_kernel, "replicas": R, "steps": K, "obs_dim": vec.obs_dim,
           "act_dim": vec.act_dim, "median_launch_ms": float(np.median(ms)), "value": R * K / (float(np.median(ms)) * 1e-3),
           "unit": "env-steps/s"}
    vec.close()
    return res


if __name__ == "__main__":
    for name in ("multiagent_ring", "multiagent_figure_eight"):
        print(json.dumps(leg(name)), flush=True)
        print(json.dumps(leg(name, launches=3, generic=True)), flush=True)
