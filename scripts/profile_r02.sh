#!/bin/bash
# Profiles the default bench.py workload on the GPU box: kernel trace + stats, then the two HBM-traffic PMC passes
# (separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Summaries go to gpurun_out/prof_r02/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 2 --warmup 1 --parity-sample 0 --stream-steps 0 $@"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.log 2>&1 || echo "fetch pass failed"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc2.log 2>&1 || echo "write pass failed"
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/trace -name "*kernel_trace.csv"); do python3 scripts/list_probe_launches.py $f $OUT/probe_launches.csv; done
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
grep "^{\"metric\"" $OUT/bench_trace.log > $OUT/bench_line.json
find $OUT -name "*.csv" -size +2M -delete
rm -rf $OUT/trace/*/*.db 2>/dev/null
ls -la $OUT
