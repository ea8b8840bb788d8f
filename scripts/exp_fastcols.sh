#!/bin/bash
# the second DP context's first-pass column buffer (BBMAP_G2_COLS): LDS per
# block decides how many blocks share a CU, jobs wider than the buffer go to the one-job-per-block wide pass
cd $GRAFT_REPO_ROOT
for fc in ${@:-640 600 576 544}; do
  BBMAP_G2_COLS=$fc timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 --default-set-steps 0 --parity-sample 0 > gpurun_out/fc_$fc.json 2> gpurun_out/fc_$fc.err || { echo "fc $fc failed"; tail -3 gpurun_out/fc_$fc.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/fc_$fc.json").read().strip().splitlines()[-1]); s=d["config"]["stage_ms"]
print("fastCols $fc: %.1f ms/step  slow %.1f rescue %.1f final %.1f  dp_wave %.1f dp_gapped %.1f" % (d["ms_per_step"], s["slow"], s["rescue"], s["final"], s["dp_wave"], s["dp_gapped"]))
PY
done
