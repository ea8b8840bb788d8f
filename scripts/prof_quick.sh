#!/bin/bash
# kernel-level timing of the bench (rocprofv3 kernel trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/profq
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --parity-sample 0 > $OUT/bench.log 2>&1 || true
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do grep -E "Name|bbidx|bbpipe|bbmsa" $f | cut -c1-150; done
tail -1 $OUT/bench.log | cut -c1-200
