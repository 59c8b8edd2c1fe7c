"""C5 with (almost) no inflow: the hot sub-step of k_merge_queue without its event handler (experiment)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if __name__ == "__main__":
    import torch
    import bench
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
    orig = bench.c5_flow_params

    def fp(precision="f32", noise=0.2):
        p = orig(precision, noise)
        for f in p["net"].inflows.get():
            f["vehsPerHour"] = f["vehsPerHour"] * scale
        return p
    bench.c5_flow_params = fp
    r = bench.c5_leg(torch.device("cuda", 0), R=1024)
    print(json.dumps({k: r[k] for k in ("kernel", "value", "departed_mean", "arrived_mean", "vehicles_in_network_mean")}))
