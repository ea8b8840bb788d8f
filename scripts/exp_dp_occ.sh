#!/bin/bash
# DP wavefront kernel: waves per SIMD asked of the compiler (launch bounds) x lanes per job, on the hg38-shaped workload
cd $GRAFT_REPO_ROOT
run() {   # $1 = macro body, $2 = lanes per job of the plain context
  BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES(R)=$1" python -m bbmap_amd.build > gpurun_out/build_dpocc.log 2>&1 || { tail -5 gpurun_out/build_dpocc.log; return; }
  BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES(R)=$1" BBMSA_LANES_PER_JOB=$2 BBMAP_G2_LANES=$2 timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep -a wall_ms | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('min waves $1 lanes $2:', {k:d[k] for k in ('wall_ms','ms_slow','ms_dp_wave','ms_dp_gapped','ms_dp_wave_max')})"
}
run "((R)<=5?5:2)" 32
run "((R)<=5?6:2)" 32
run "((R)<=3?6:4)" 64
run "((R)<=3?8:4)" 64
python -m bbmap_amd.build > /dev/null 2>&1
