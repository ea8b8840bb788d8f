#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 448 544 608; do
  BBPIPE_GAP_FAST_COLS=$g timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > gpurun_out/gc_$g.log 2>&1 || exit 1
  echo "fast_cols=$g $(tail -1 gpurun_out/gc_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"]["gapped_dp_kernels"])')"
done
