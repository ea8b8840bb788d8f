#!/bin/bash
# probe time on the hg38-shaped workload at different occupancies of the long-list / short-read variant (waves per SIMD)
cd $GRAFT_REPO_ROOT
for occ in ${1:-3 4 5}; do
  BBMSA_CXXFLAGS="-DBBIDX_LONG_SHORT_OCC=$occ" python -m bbmap_amd.build > gpurun_out/build_occ.log 2>&1 || { tail -5 gpurun_out/build_occ.log; exit 1; }
  echo "== occupancy $occ"
  BBMSA_CXXFLAGS="-DBBIDX_LONG_SHORT_OCC=$occ" timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('probe ms', d['ms_probe'])"
done
python -m bbmap_amd.build > /dev/null 2>&1
