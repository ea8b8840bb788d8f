#!/bin/bash
# instruction mix / stall counters of the wavefront DP kernel on the bench workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_dp
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_READY SQ_WAVE_DEP_WAIT SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "msa_fill_fast" --output-format csv -d $OUT/p$i -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --parity-sample 0 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/summarize_pmc.py $OUT | grep -v "^==" | awk '{print $2, $4, $5}' | sort -u
