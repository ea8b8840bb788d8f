#!/bin/bash
# instruction mix / stall counters of the probe kernel on an hg38-sized synthetic reference
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/probe_big
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex probe_wave --output-format csv -d $OUT/p$i -- python3 scripts/exp_probe_big.py ${1:-24} ${2:-129000000} > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/summarize_pmc.py $OUT | grep -v "^==" | awk '{print $2, $5}' | sort -u
