#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_index_gpu.py -q -x 2>&1 | tail -1
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pf_a.log 2>&1 || exit 1
echo "ecoli $(tail -1 gpurun_out/pf_a.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"]["probe_wave_kernel"]["ms"], d["config"]["parity"])')"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ref-len 46709983 --repeat-frac 0.10 --max-sites 16 > gpurun_out/pf_b.log 2>&1 || exit 1
echo "chr21r $(tail -1 gpurun_out/pf_b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels"]["probe_wave_kernel"]["ms"], d["config"]["parity"])')"
