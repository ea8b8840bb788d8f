#!/bin/bash
# Profiles bench.py (whole pipeline) on the GPU box: kernel trace/stats first, then PMC passes (separate runs).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 3 --warmup 1 --parity-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc2.log 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_pmc3.log 2>&1 || true
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
python3 scripts/summarize_pmc.py $OUT || true
