#!/bin/bash
# second DP context (gapped references, wide windows) on the hg38-shaped workload: job widths, then its time for lane / column geometries
cd $GRAFT_REPO_ROOT
python -m bbmap_amd.build > gpurun_out/build_exp.log 2>&1 || { tail -5 gpurun_out/build_exp.log; exit 1; }
for geo in "${@:-64:640:256}"; do
  IFS=: read lanes cols fast <<< "$geo"
  echo "== second context: $lanes lanes per job, first pass up to $cols columns; first context up to ${fast:-256} columns"
  BBMAP_FASTCOLS=${fast:-256} BBMAP_G2_LANES=$lanes BBMAP_G2_COLS=$cols BBMAP_G2_HIST=1 timeout -k 10 300 python scripts/exp_mapper.py hg38 2000000 2>&1 | grep -E "wall_ms|gapped widths" | tail -2 | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print({k: d[k] for k in ('wall_ms','ms_probe','ms_slow','ms_rescue','ms_dp_wave','ms_dp_gapped','gapped_fills','rounds')})
    else: print(ln.strip())"
done
