#!/bin/bash
# Round 4 profile of the default bench.py workload (the step now ends with the final alignment stage) and of --workload pacbio, on the
# GPU box; separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (never a --pmc pass combined with a trace).
# Summaries go to gpurun_out/prof_r04/; scripts/make_traffic_r04.py turns them into profiles/traffic_r04.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 2 --warmup 1 --parity-sample 0 --stream-steps 0 --default-set-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1 || echo "trace pass failed"
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.log 2>&1 || echo "fetch pass failed"
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc2.log 2>&1 || echo "write pass failed"
echo "write done"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-include-regex "probe_wave|msa_fill_fast" --output-format csv -d $OUT/sq$i -- python3 $ARGS > $OUT/bench_sq$i.log 2>&1 || echo "sq pass $i failed"
  echo "sq pass $i done"
done
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/trace -name "*kernel_trace.csv"); do python3 scripts/list_probe_launches.py $f $OUT/probe_launches.csv; done
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
grep "^{\"metric\"" $OUT/bench_trace.log > $OUT/bench_line.json
# ---- mapPacBio: one step (8,192 pieces), kernel stats + the same counter groups for the long-read probe and the strip kernel
PB=gpurun_out/prof_r04_pacbio
rm -rf $PB && mkdir -p $PB
PARGS="bench.py --workload pacbio --no-cpu-baseline --steps 1 --warmup 0 --parity-sample 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $PB/trace -- python3 $PARGS > $PB/bench_trace.log 2>&1 || echo "pacbio trace pass failed"
echo "pacbio trace done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $grp --kernel-include-regex "probe_long|msa_fill_strip" --output-format csv -d $PB/pmc$i -- python3 $PARGS > $PB/bench_pmc$i.log 2>&1 || echo "pacbio pmc pass $i failed"
  echo "pacbio pmc pass $i done"
done
for f in $(find $PB/trace -name "*kernel_stats.csv"); do cp $f $PB/kernel_stats.csv; done
python3 scripts/summarize_pmc.py $PB > $PB/pmc_summary.txt 2>&1
grep "^{\"metric\"" $PB/bench_trace.log > $PB/bench_line.json
find $OUT $PB -name "*.csv" -size +2M -delete
find $OUT $PB -name "*.db" -delete
ls -la $OUT $PB
