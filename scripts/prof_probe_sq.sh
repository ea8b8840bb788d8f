#!/bin/bash
# where the probe kernel's wave-cycles go (SQ counters, one --pmc pass per group) on the default bench workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 1 --warmup 1 --parity-sample 0 --stream-steps 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
grep -a "probe_wave_kernel\|^==" $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
