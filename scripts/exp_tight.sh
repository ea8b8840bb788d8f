#!/bin/bash
cd $GRAFT_REPO_ROOT
for ts in 900 1200 1500 1800; do
  BBMSA_TIGHT_SLACK=$ts timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iterations > gpurun_out/tight_$ts.log 2>&1 || { tail -3 gpurun_out/tight_$ts.log; exit 1; }
  echo "tight=$ts $(tail -1 gpurun_out/tight_$ts.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["dp_jobs_by_kernel"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernels"].items()}, d["config"]["parity"])')"
done
