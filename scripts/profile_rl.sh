#!/bin/bash
# Profile the RL ring kernels (k_ring_pair open loop, k_ring_policy closed loop) on the GPU box (run through gpurun).
set -e
TAG=${1:-r03}
OUT=gpurun_out/prof_rl_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/dbg/ring_rl_rate.py > $OUT/rate_trace.txt 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 scripts/dbg/ring_rl_rate.py > $OUT/rate_sq.txt 2> $OUT/sq.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pol -- python3 scripts/dbg/policy_rate.py > $OUT/policy_trace.txt 2> $OUT/trace_pol.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq_pol -- python3 scripts/dbg/policy_rate.py > $OUT/policy_sq.txt 2> $OUT/sq_pol.err
tail -n 3 $OUT/rate_trace.txt; tail -n 3 $OUT/policy_trace.txt
