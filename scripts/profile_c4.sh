#!/bin/bash
# Profile the open-network kernel on the C4 bottleneck leg (run through gpurun).
set -e
TAG=${1:-r01}
R=${2:-128}
LEG=${3:-c4}          # c4: lane changing off (k_drop_queue); c4lc: on (k_steps_wide)
OUT=gpurun_out/prof_${LEG}_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_c5.py $R $LEG > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 scripts/bench_c5.py $R $LEG > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 scripts/bench_c5.py $R $LEG > $OUT/bench_sq2.json 2> $OUT/sq2.err
