#!/bin/bash
# probe time (ms per 2 M reads, hg38-shaped workload) of variant builds: bash scripts/exp_variants.sh name1 name2 ... ("default" = the tree's library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then unset BBMAP_AMD_SO; else export BBMAP_AMD_SO=$PWD/bbmap_amd/_variants/lib_$v.so; fi
  timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v: probe %.2f ms, final %.2f ms, step %.1f ms' % (d['ms_probe'], d.get('ms_final', 0.0), d['wall_ms']))" || { echo "$v failed"; exit 1; }
done
