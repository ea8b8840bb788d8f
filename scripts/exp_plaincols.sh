#!/bin/bash
# the first DP context's column limit (bbmap_config.fastCols): windows up to this width are "plain", wider ones go to the second context
cd $GRAFT_REPO_ROOT
for fc in "$@"; do
  BBMAP_FASTCOLS=$fc timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep wall_ms | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('plain columns $fc: step %.1f ms  slow %.1f rescue %.1f final %.1f  dp_wave %.1f dp_gapped %.1f' % (d['wall_ms'], d['ms_slow'], d['ms_rescue'], d['ms_final'], d['ms_dp_wave'], d['ms_dp_gapped']))" || { echo "$fc failed"; exit 1; }
done
