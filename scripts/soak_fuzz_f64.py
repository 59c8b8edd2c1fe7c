"""One-off soak of the float64 fuzz cases (tests/test_open_gpu.py, tests/test_wide_gpu.py) with seeds beyond the pinned ones:
python scripts/soak_fuzz_f64.py [first] [count]"""
import os
import sys
import traceback

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))

if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import test_open_gpu as to
    import test_wide_gpu as tw
    bad = []
    for seed in range(first, first + count):
        for name, fn in (("open_f64", to.test_fuzz_random_open_network_configs_float64),
                         ("wide_f64", tw.test_wide_fuzz_random_lane_drop_configs_float64)):
            try:
                fn(seed)
            except Exception:
                bad.append((name, seed))
                traceback.print_exc()
        print("seed", seed, "done", flush=True)
    print("FAILED:" if bad else "all ok", bad)
    sys.exit(1 if bad else 0)
