#!/bin/bash
# probe time of one workload under different build flags: bash scripts/exp_probe_flags.sh <workload> <n_reads> "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
W=$1; N=$2; shift; shift
for fl in "$@"; do
  BBMSA_CXXFLAGS="$fl" python -m bbmap_amd.build > gpurun_out/build_fl.log 2>&1 || { tail -5 gpurun_out/build_fl.log; exit 1; }
  echo "== flags: $fl"
  BBMSA_CXXFLAGS="$fl" timeout -k 10 300 python scripts/exp_mapper.py $W $N 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('probe ms', d['ms_probe'], 'total', d['ms_total'])"
done
python -m bbmap_amd.build > /dev/null 2>&1
