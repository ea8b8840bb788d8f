#!/bin/bash
# dynamic instruction counts of the probe kernel on the default bench workload (one --pmc pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_insts
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --parity-sample 0 --stream-steps 0 > $OUT/log1.txt 2>&1 || echo "pass failed"
python3 scripts/summarize_pmc.py $OUT 2>&1 | grep -a "probe_wave_kernel"
find $OUT -name "*.csv" -size +2M -delete
