#!/bin/bash
# step and stage times (ms per 2 M reads, hg38-shaped workload) of variant builds: bash scripts/exp_variants_step.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then unset BBMAP_AMD_SO; else export BBMAP_AMD_SO=$PWD/bbmap_amd/_variants/lib_$v.so; fi
  timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v: step %.1f ms  probe %.1f slow %.1f rescue %.1f final %.1f  dp_wave %.1f dp_gapped %.1f' % (d['wall_ms'], d['ms_probe'], d['ms_slow'], d['ms_rescue'], d['ms_final'], d['ms_dp_wave'], d['ms_dp_gapped']))" || { echo "$v failed"; exit 1; }
done
