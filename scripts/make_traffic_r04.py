"""profiles/traffic_r04.json from a scripts/profile_r04.sh run: HBM bytes per launch of the two big kernels on the profiled
workload.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; /opt/skills/guides/MI355X_MICROARCH.md: on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled (calibrated by the guide for wide streaming reads; the probe's
4-byte gathers are an uncalibrated pattern, so the raw figure is kept next to it)."""
import json
import re
import sys

src, workload, reads, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
vals = {}
for line in open(src):
    m = re.match(r"\s+(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+sum=([\d.e+]+)\s+dispatches=(\d+)\s+per_dispatch=([\d.e+]+)", line)
    if m:
        vals.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
res = {}
for short, pat in (("probe_wave_kernel", "probe_wave_kernel"), ("msa_fill_fast_kernel", "msa_fill_fast_kernel<5")):
    for name, v in vals.items():
        if pat in name and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            steps = 3          # the profiled command runs 1 warm-up + 2 timed steps; per step = per main launch of the probe (the
                               # overflow tier's own small probe launch, a few hundred reads, is counted in) and all DP launches of a step
            fetch = v["FETCH_SIZE"][0] * 1024 / steps
            write = v["WRITE_SIZE"][0] * 1024 / steps
            res[short] = {"reads_per_step": reads, "fetch_bytes_raw": int(fetch), "write_bytes": int(write),
                          "hbm_bytes_per_launch": int(2 * fetch + write),
                          "source": "profiles/r04_%s_pmc_summary.txt" % workload}
try:
    cur = json.load(open(out))
except Exception:
    cur = {}
cur[workload] = res
json.dump(cur, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
