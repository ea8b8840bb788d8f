#!/bin/bash
# kernel trace of the mapper on the hg38-shaped workload (3 steps): every launch of the last step with its duration, in launch order
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/trace_step
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 scripts/exp_mapper.py hg38 2000000 > $OUT/log.txt 2>&1 || echo "trace failed"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_step/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last probe_wave_kernel launch with a large grid
idx = [i for i, r in enumerate(rows) if "probe_wave_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 40_000_000]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
with open("gpurun_out/trace_step/last_step.txt", "w") as out:
    for r in rows[start:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        line = "%9.3f ms  +%8.3f  q%-3s %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Queue_Id", "?"), r["Kernel_Name"][:70], r.get("Grid_Size", "?"))
        out.write(line + "\n")
        if d > 0.3: print(line)
PY
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
