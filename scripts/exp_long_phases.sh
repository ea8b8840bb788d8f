#!/bin/bash
# where the long-read probe kernel's cycles go (debug build -DBBIDXL_TIMERS: probe_stats = cycles / 1024 in
# {refillAll, popSite incl. refills, quick scores, minHead + countWindow, whole prescan + walk loops}) on the pacbio bench workload
cd $GRAFT_REPO_ROOT
BBMSA_CXXFLAGS="-DBBIDXL_TIMERS" python -m bbmap_amd.build > gpurun_out/build_lt.log 2>&1 || { tail -5 gpurun_out/build_lt.log; exit 1; }
BBMSA_CXXFLAGS="-DBBIDXL_TIMERS" timeout -k 10 400 python bench.py --workload pacbio --reads ${1:-256} --steps 1 --warmup 1 --parity-sample 0 --no-cpu-baseline > gpurun_out/long_phases.json 2> gpurun_out/long_phases.err || tail -5 gpurun_out/long_phases.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/long_phases.json"))
ps = d["config"]["probe_stats_raw"]
print("probe ms", d["config"]["stage_ms"]["probe"], "kcycles: refill %d  popSite(incl refill) %d  qscore %d  minHead+count %d  loops total %d" % tuple(ps))
print(open("gpurun_out/long_phases.err").read()[-600:])
PY
python -m bbmap_amd.build > /dev/null 2>&1
