#!/bin/bash
# replicas-per-GPU sweep of the headline rollout kernel (scripts/ubench/pair_bench: k_rollout_pair, C2 shape): how far
# the kernel is from the HBM roof once the chip holds several waves per SIMD (BASELINE's 4096 replicas are ~1 wave per
# SIMD).  Prints ms per launch and env-steps/s for float32 and FS_MIXED; algorithmic bytes = 181.2 B per env-step.
for cfg in "4096 1500" "8192 1500" "16384 750" "32768 375" "65536 200" "131072 100"; do
  set -- $cfg
  scripts/ubench/pair_bench $1 $2 256
done
