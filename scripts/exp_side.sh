#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 0 1; do
  BBPIPE_SIDE_STREAM=$g timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/side_$g.log 2>&1 || exit 1
  echo "side=$g $(tail -1 gpurun_out/side_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["parity"])')"
done
