#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/build_trace
rm -rf $OUT && mkdir -p $OUT
BBIDX_BUILD_TIMERS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 scripts/exp_build_trace.py > $OUT/log.txt 2>&1 || echo "trace failed"
grep "build \|bbidx_build" $OUT/log.txt
for f in $(find $OUT/t -name "*kernel_stats.csv"); do head -16 $f | cut -c1-160; done
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
