#!/bin/bash
# round 3: probe time on the hg38-shaped workload for build variants: "<flags>" per argument (empty string = default build)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  BBMSA_CXXFLAGS="$flags" python -m bbmap_amd.build > gpurun_out/build_exp.log 2>&1 || { tail -5 gpurun_out/build_exp.log; exit 1; }
  echo "== flags: $flags"
  BBMSA_CXXFLAGS="$flags" timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('wall_ms','ms_probe','ms_score','ms_slow','ms_rescue','ms_dp_wave','ms_dp_gapped','ms_dp_narrow','fills','gapped_fills','rounds')})"
done
python -m bbmap_amd.build > /dev/null 2>&1
