#!/bin/bash
cd $GRAFT_REPO_ROOT
for occ in 5; do
  BBMSA_CXXFLAGS="-DBBMSA_MIN_WAVES(R)=$occ" python -m bbmap_amd.build --force > gpurun_out/build_occ.log 2>&1 || { tail -5 gpurun_out/build_occ.log; exit 1; }
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dpocc_$occ.log 2>&1 || { tail -3 gpurun_out/dpocc_$occ.log; exit 1; }
  echo "occ=$occ $(tail -1 gpurun_out/dpocc_$occ.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernels"].items()}, d["config"]["parity"])')"
done
