#!/bin/bash
# Round 3: settle the probe kernel's bound.  SQ instruction/wait counters and L2 hit/miss of probe_wave_kernel on the default
# bench workload (one --pmc pass per group, kernel-trace/stats not combined), then an occupancy sweep (waves per SIMD).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03_probe
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --no-cpu-baseline --steps 1 --warmup 1 --parity-sample 0 --stream-steps 0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "probe_wave" --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
grep -a "probe_wave_kernel" $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
for occ in 3 4 5; do
  BBMSA_CXXFLAGS="-DBBIDX_LONG_SHORT_OCC=$occ" python -m bbmap_amd.build > $OUT/build_occ$occ.log 2>&1 || { tail -5 $OUT/build_occ$occ.log; exit 1; }
  echo "== occupancy $occ"
  BBMSA_CXXFLAGS="-DBBIDX_LONG_SHORT_OCC=$occ" timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('probe ms', d['ms_probe'])" | tee -a $OUT/occupancy.txt
done
python -m bbmap_amd.build > /dev/null 2>&1
