#!/bin/bash
# experiment: probe wave kernel at different register budgets (waves per SIMD)
cd $GRAFT_REPO_ROOT
for occ in 6 7 8; do
  BBMSA_CXXFLAGS="-DBBIDX_WAVE_OCC=$occ" python -m bbmap_amd.build --force > /dev/null 2>&1 || exit 1
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > gpurun_out/occ_$occ.log 2>&1 || exit 1
  echo "occ=$occ $(tail -1 gpurun_out/occ_$occ.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"]["probe_kernel"]["ms"])')"
done
