#!/bin/bash
# cycles per phase of the probe kernel (debug build with phase timers) on the hg38-shaped workload, with and without the whole-cycle walk
cd $GRAFT_REPO_ROOT
for cyc in 1 0; do
  BBMSA_CXXFLAGS="-DBBIDX_PHASE_TIMERS -DBBIDX_CYCLE=$cyc" python -m bbmap_amd.build > gpurun_out/build_ph.log 2>&1 || { tail -5 gpurun_out/build_ph.log; exit 1; }
  echo "== BBIDX_CYCLE=$cyc"
  BBMSA_CXXFLAGS="-DBBIDX_PHASE_TIMERS -DBBIDX_CYCLE=$cyc" timeout -k 10 300 python scripts/exp_mapper.py ${1:-hg38} ${2:-2000000} 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); ps = d['probe_stats']; t = sum(ps)
print('probe ms', d['ms_probe'], 'phases (keys+lookup, trim/setup, prescan, walk, extend) %:', [round(100.0 * x / t, 1) for x in ps])"
done
python -m bbmap_amd.build > /dev/null 2>&1
