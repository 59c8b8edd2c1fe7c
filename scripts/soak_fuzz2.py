"""One-off soak of the rollout-kernel fuzz cases (tests/test_pair_gpu.py, tests/test_parity_gpu.py, tests/test_ringrl_gpu.py) with seeds beyond the
pinned ones: python scripts/soak_fuzz2.py [first] [count]"""
import os
import sys
import traceback

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))

if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import test_pair_gpu as tpair
    import test_parity_gpu as tpar
    import test_ringrl_gpu as tring
    bad = []
    for seed in range(first, first + count):
        for name, fn in (("pair", tpair.test_pair_hand_written_steps_fuzz_f32_and_mixed),
                         ("loop", tpar.test_loop_rollout_kernel_fuzz_against_generic_kernel),
                         ("ring_rl", tring.test_fuzz_ring_pair_equals_generic_kernel)):
            try:
                fn(seed)
            except Exception:
                bad.append((name, seed))
                traceback.print_exc()
        print("seed", seed, "done", flush=True)
    print("FAILED:" if bad else "all ok", bad)
    sys.exit(1 if bad else 0)
