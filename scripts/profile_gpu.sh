#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun).  Writes under gpurun_out/prof_$TAG_$PREC;
# scripts/summarize_profile.py turns the CSVs into the summaries committed under profiles/.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   scripts/profile_gpu.sh r02 mixed
set -e
TAG=${1:-r02}
PREC=${2:-mixed}
OUT=gpurun_out/prof_${TAG}_${PREC}
ARGS="--steps 7500 --warmup 1500 --no-extras --precision $PREC"
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python3 bench.py $ARGS > $OUT/bench_grbm.json 2> $OUT/grbm.err
python3 scripts/summarize_profile.py $OUT $TAG $PREC > $OUT/summary.txt
tail -5 $OUT/summary.txt
