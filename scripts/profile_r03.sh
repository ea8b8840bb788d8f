#!/bin/bash
# Round 3 profile of the default bench.py workload on the GPU box (separate runs, as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes): kernel trace + stats, the two HBM-traffic PMC passes, SQ / TCC counter passes for the probe and the wavefront DP.
# Summaries go to gpurun_out/prof_r03/; scripts/make_traffic_r03.py turns them into profiles/traffic_r03.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 2 --warmup 1 --parity-sample 0 --stream-steps 0 --default-set-steps 0 $@"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1 || echo "trace pass failed"
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc1.log 2>&1 || echo "fetch pass failed"
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc2.log 2>&1 || echo "write pass failed"
echo "write done"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-include-regex "probe_wave|msa_fill_fast" --output-format csv -d $OUT/sq$i -- python3 $ARGS > $OUT/bench_sq$i.log 2>&1 || echo "sq pass $i failed"
  echo "sq pass $i done"
done
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/trace -name "*kernel_trace.csv"); do python3 scripts/list_probe_launches.py $f $OUT/probe_launches.csv; done
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
grep "^{\"metric\"" $OUT/bench_trace.log > $OUT/bench_line.json
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls -la $OUT
