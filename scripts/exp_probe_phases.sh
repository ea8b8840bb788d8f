#!/bin/bash
# cycles per phase of the probe kernel (debug builds with cycle counters, see BBIDX_PHASE_TIMERS in index_probe_wave.hip) on the
# hg38-shaped workload: mode 1 = the read's phases, 2 = inside the walk, 3 = inside the prescan
cd $GRAFT_REPO_ROOT
for mode in ${3:-1 2 3}; do
  BBMSA_CXXFLAGS="-DBBIDX_PHASE_TIMERS=$mode" python -m bbmap_amd.build > gpurun_out/build_ph.log 2>&1 || { tail -5 gpurun_out/build_ph.log; exit 1; }
  echo "== BBIDX_PHASE_TIMERS=$mode"
  BBMSA_CXXFLAGS="-DBBIDX_PHASE_TIMERS=$mode" timeout -k 10 300 python scripts/exp_mapper.py ${1:-hg38} ${2:-2000000} 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); ps = d['probe_stats']; t = sum(ps)
print('probe ms', d['ms_probe'], 'slots %:', [round(100.0 * x / t, 1) for x in ps], 'sum (16-cycle units)', t)"
done
python -m bbmap_amd.build > /dev/null 2>&1
