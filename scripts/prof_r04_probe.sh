#!/bin/bash
# Round 4, verdict item 2: which pipe bounds probe_wave_kernel on the 18-key hg38-shaped workload.  One --pmc pass per counter
# group (kernel-filtered), then the phase-timer builds.  Summaries land in gpurun_out/prof_r04_probe/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04_probe
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 1 --warmup 1 --parity-sample 0 --stream-steps 0 --default-set-steps 0"
i=0
for grp in "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 420 rocprofv3 --pmc $grp --kernel-include-regex "probe_wave" --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
grep -a "probe_wave_kernel\|^==" $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
bash scripts/exp_probe_phases.sh hg38 2000000 "1 2 3" > $OUT/phases.txt 2>&1
cat $OUT/phases.txt
