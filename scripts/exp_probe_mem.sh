#!/bin/bash
cd $GRAFT_REPO_ROOT
BBMSA_CXXFLAGS="-DBBIDX_WAVE_OCC=6" python -m bbmap_amd.build --force > /dev/null 2>&1 || exit 1
for rl in 200000 4641652; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 --ref-len $rl > gpurun_out/mem_$rl.log 2>&1 || exit 1
  echo "reflen=$rl $(tail -1 gpurun_out/mem_$rl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"], d["config"]["dp_jobs_per_step"])')"
done
