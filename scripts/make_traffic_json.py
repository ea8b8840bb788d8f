"""profiles/traffic.json from the summary scripts/profile_bench.sh prints (gpurun_out/prof_final.log)."""
import json
import re
import sys

log = open(sys.argv[1]).read().splitlines()
sec = None
pmc = {}      # section -> kernel -> counter -> (sum, dispatches)
dur = {}      # section -> kernel -> (calls, avg_ms)
for ln in log:
    m = re.match(r"== (\S+)", ln)
    if m:
        path = m.group(1)
        sec = path.split("/")[2] if path.count("/") >= 2 else path
        continue
    m = re.match(r"\s+(.*?)\s+(\w+)\s+sum=([\d.e+]+)\s+dispatches=(\d+)", ln)
    if m and sec:
        pmc.setdefault(sec, {}).setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
        continue
    m = re.match(r"\s+(.*?)\s+calls=(\d+) avg_ms=([\d.]+)", ln)
    if m and sec:
        dur.setdefault(sec, {})[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))


def find(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return None


STEPS = 4     # warmup 1 + steps 3
names = {"probe_wave_kernel": "probe_wave_kernel", "msa_fill_fast_kernel<5>": "msa_fill_fast_kernel<5",
         "msa_fill_fast_kernel<3> (gapped-reference context, two dispatches per step)": "msa_fill_fast_kernel<3",
         "msa_fill_narrow_kernel": "msa_fill_narrow_kernel", "select_jobs_kernel": "select_jobs_kernel"}
per_step = {}
for label, key in names.items():
    f = find(pmc.get("pmc_fetch", {}), key)
    w = find(pmc.get("pmc_write", {}), key)
    per_step[label] = {"fetch": round(f["FETCH_SIZE"][0] / STEPS), "write": round(w["WRITE_SIZE"][0] / STEPS)}
out = {
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_*, separate passes, `bench.py --no-cpu-baseline --steps 3 "
               "--warmup 1 --parity-sample 0` (scripts/profile_bench.sh -> scripts/make_traffic_json.py), 1,000,000 reads per step; "
               "summaries: profiles/r01_final_pmc_summary.txt, profiles/r01_final_kernel_stats.csv (earlier states of the round: "
               "r01_pmc_summary.txt = DP-only bench, r01_pipeline_* = per-lane probe kernel)",
    "_correction": "MI355X_MICROARCH.md HBM section: counters are in KiB; FETCH_SIZE on gfx950 tallies 128-B requests at 64 B, so it is "
                   "doubled for the wide coalesced reads of the DP kernels (record read-back); the probe kernel's reads are scattered "
                   "4-32-byte gathers, a width the guide calls uncalibrated, so its FETCH_SIZE is taken as reported (a lower bound).  "
                   "WRITE_SIZE is taken as reported.  Per launch = per dispatch of that kernel in the ordinary (non-gapped) DP launch sequence.",
    "per_step_kib": per_step,
    "probe_wave_kernel_bytes_per_launch": int((per_step["probe_wave_kernel"]["fetch"] + per_step["probe_wave_kernel"]["write"]) * 1024),
    "msa_fill_fast_kernel_bytes_per_launch": int((2 * per_step["msa_fill_fast_kernel<5>"]["fetch"] + per_step["msa_fill_fast_kernel<5>"]["write"]) * 1024),
    "msa_fill_narrow_kernel_bytes_per_launch": int((2 * per_step["msa_fill_narrow_kernel"]["fetch"] + per_step["msa_fill_narrow_kernel"]["write"]) * 1024),
}
peak = 1024 * 2.4e9 / 2
vi = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/r01_final_pmc_summary.txt): VALU wave-instructions of all dispatches of the kernel "
                 "divided by their summed duration in the same run; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction "
                 "(MI355X_MICROARCH.md, SIMD-32) = 1.2288e12 wave-instructions/s",
      "peak_wave_inst_per_s": peak}
for label, key in (("msa_fill_fast_kernel", "msa_fill_fast_kernel<5"), ("msa_fill_narrow_kernel", "msa_fill_narrow_kernel"),
                   ("probe_wave_kernel", "probe_wave_kernel")):
    v = find(pmc["pmc_sq"], key)["SQ_INSTS_VALU"][0]
    calls, avg = find(dur["pmc_sq"], key)
    secs = calls * avg / 1e3
    vi[label] = {"valu_wave_inst": v, "seconds": round(secs, 6), "frac": round(v / secs / peak, 3)}
out["valu_issue"] = vi
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["per_step_kib"]), {k: v["frac"] for k, v in vi.items() if isinstance(v, dict)})
