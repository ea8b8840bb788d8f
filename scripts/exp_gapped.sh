#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 32 64; do
  BBPIPE_GAPPED_LANES=$g timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > gpurun_out/gap_$g.log 2>&1 || exit 1
  echo "G=$g $(tail -1 gpurun_out/gap_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"]["gapped_dp_kernels"])')"
done
