#!/bin/bash
# how the whole-cycle prescan fares on a workload: cycles offered / declined, entries and candidates per cycle (debug build)
cd $GRAFT_REPO_ROOT
BBMSA_CXXFLAGS="-DBBIDX_CYC_STATS" python -m bbmap_amd.build > gpurun_out/build_cs.log 2>&1 || { tail -5 gpurun_out/build_cs.log; exit 1; }
BBMSA_CXXFLAGS="-DBBIDX_CYC_STATS" timeout -k 10 300 python scripts/exp_mapper.py ${1:-hg38} ${2:-2000000} 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); ps = d['probe_stats']
print('probe ms', d['ms_probe'], 'prescan cycles offered', ps[0], 'declined', ps[1], 'entries/cycle', ps[2] / max(1, ps[0] - ps[1]), 'candidates/cycle', ps[3] / max(1, ps[0] - ps[1]), 'visited/cycle', ps[4] / max(1, ps[0] - ps[1]))"
python -m bbmap_amd.build > /dev/null 2>&1
