"""Every probe_wave_kernel dispatch of a rocprofv3 --kernel-trace run (scripts/profile_r02.sh): python scripts/list_probe_launches.py <trace.csv> <out.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = ["# every probe_wave_kernel dispatch of the profiled run (rocprofv3 --kernel-trace): the long ones are the batch's probe (1 warm-up + 2 timed steps),",
       "# the short ones the overflow tier's probe over the few hundred reads whose site list did not fit (DESIGN 8)",
       "grid_x,workgroup_x,duration_ms"]
for r in rows:
    if "probe_wave_kernel" in r["Kernel_Name"]:
        out.append("%s,%s,%.3f" % (r["Grid_Size_X"], r["Workgroup_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
