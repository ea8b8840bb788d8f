#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the MI355X seed-and-extend hot path.

A "step" is one pass of the hot path over one batch of synthetic reads that is already resident in
HBM.  At N=1 the workload is BASELINE.json configs[1]: an E. coli K-12 sized reference
(4,641,652 bp, synthetic, seed 1) and 1,000,000 synthetic 150-bp single-end reads (seed 2, the
"mutated" read mix), one slow-align call (fillAndScoreLimited + traceback) per read at its candidate
site.  With N>1 every rank runs the same-sized shard (weak scaling, no collective on the data path;
the reference is replicated per GPU).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how each field is obtained.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def _emulate_read(oi, om, refb, bp, bm, offsets, key_scores, max_sites, L):
    """The reference's control flow for one read, from the oracle's pieces (probe -> scoreNoIndels -> scoreSlow)."""
    from oracle.oracle import score_no_indels, set_perfect
    maxSw = 70 + (L - 1) * 100
    maxImp = maxSw - 495
    minMsaLimit = -258 + int(np.float32(0.56) * np.float32(maxSw))
    try:
        sites = oi.find(bp, bm, [0] * L, key_scores, offsets, cap=max_sites)
    except RuntimeError:                      # more than max_sites sites: the probe must report the overflow (nsites = -1)
        return None, None, None
    near, force, sws = 0, False, []
    for e in sites:                                   # AbstractMapThread.scoreNoIndels, current/align2/AbstractMapThread.java:762-856
        bases = bm if e["strand"] else bp
        if e["perfect"]:
            sw = maxSw
            near += 1
            e["gaps"] = []
        else:
            old = e["score"]
            sw = score_no_indels(bases, refb, e["start"])
            if sw < old and old >= maxImp and e["stop"] - e["start"] + 1 != L:
                sw2 = score_no_indels(bases, refb, e["stop"] - L + 1)
                if sw2 >= maxImp:
                    sw = sw2
                    e["start"] = e["stop"] - L + 1
                    e["perfect"], e["semiperfect"] = set_perfect(bases, refb, e["start"], e["stop"])
            if sw >= maxImp:
                near += 1
                e["stop"] = e["start"] + L - 1
                e["gaps"] = []
                if sw >= maxSw:
                    e["perfect"] = e["semiperfect"] = 1
                else:
                    e["perfect"], e["semiperfect"] = set_perfect(bases, refb, e["start"], e["stop"])
            elif old >= maxImp:
                force = True
        sws.append(sw)
    dp = []
    if (-near if force else near) < 1:
        for s, e in enumerate(sites):
            if sws[s] < maxImp and not e["semiperfect"] and not e["gaps"]:
                bases = bm if e["strand"] else bp
                ms = max(sws[s], minMsaLimit)
                sv, mx = om.fillAndScoreLimited(bases, refb, e["start"] - 4, e["stop"] + 4, ms)
                tb = None
                if sv is not None:
                    tb = om.traceback(bases, refb, max(0, e["start"] - 4), e["stop"] + 4, mx[0], mx[1], mx[2])
                dp.append((s, sv, tb))
    return sites, sws, dp


def parity_sample(pipe, out, reads, ref, hi, offsets, key_scores, count, max_sites, max_cols):
    """Checks the first `count` reads of the last step end to end against the oracle."""
    from oracle.oracle import OracleIndex, OracleMSA, score_no_indels_match
    L = pipe.read_len
    oi = OracleIndex([ref], k=hi.k, chromBits=hi.chromBits)
    om = OracleMSA(160, max_cols)
    refb = ref.tobytes()
    comp = np.full(256, 255, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    by_src = {int(s): i for i, s in enumerate(out["src"])}
    bad = 0
    for r in range(count):
        bp_a = reads[r * L:(r + 1) * L]
        bp, bm = bp_a.tobytes(), comp[bp_a[::-1]].tobytes()
        sites, sws, dp = _emulate_read(oi, om, refb, bp, bm, offsets, key_scores, max_sites, L)
        if sites is None:
            bad += int(out["nsites"][r]) != -1
            continue
        ok = int(out["nsites"][r]) == len(sites)
        for s, e in enumerate(sites if ok else []):
            g = out["sites"][r, s]
            ok &= (int(g["chrom"]), int(g["strand"]), int(g["start"]), int(g["hits"])) == (e["chrom"], e["strand"], e["start"], e["hits"])
            ok &= int(out["no_indel"][r, s]) == sws[s]
        # reads finished without DP carry the ungapped match string of their best site
        st = int(out["read_state"][r])
        if sites and not dp and (st & 3) == 1:
            e = sites[st >> 2]
            sc, ms = score_no_indels_match(bm if e["strand"] else bp, refb, e["start"])
            ok &= (int(out["ungapped_len"][r]) == -1) if sc == -99999 else (out["ungapped_match"][r].tobytes() == ms)
        want = {s for s, _, _ in dp}
        have = {src % max_sites for src in by_src if src // max_sites == r}
        ok &= want == have
        for s, sv, tb in (dp if ok else []):
            i = by_src[r * max_sites + s]
            res = out["results"][i]
            gs = None if res["score_len"] == 0 else res["score"][: res["score_len"]].tolist()
            ok &= gs == sv
            if sv is not None:
                ok &= out["match"][i, : res["match_len"]].tobytes() == tb
        bad += (not ok)
    return {"checked_reads": count, "mismatches": bad}


def cpu_baseline(k, chrom_bits, reads, ref, L, offsets, key_scores, max_cols, target_seconds=15.0):
    """The same per-read pipeline on the host cores, built from the CPU oracle (a port of the reference's logic):
    probe + ungapped filter + DP + traceback (oracle/bench_oracle.c:orc_bench_map), one worker thread per usable
    core sharing one read-only index, on a bounded sample of the same read batch."""
    from oracle.oracle import OracleIndex
    cores = usable_cores()
    oi = OracleIndex([ref], k=k, chromBits=chrom_bits)
    L_ = oi.L
    L_.orc_bench_map.restype = C.c_double
    L_.orc_bench_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    offs = np.asarray(offsets, np.int32)
    ks = np.asarray(key_scores, np.int32)
    nreads = len(reads) // L
    mapped, jobs, cells = C.c_int64(), C.c_int64(), C.c_int64()

    def run(count):
        return L_.orc_bench_map(C.c_void_p(oi.h), reads.ctypes.data, count, L, offs.ctypes.data, ks.ctypes.data, len(offs),
                                max_cols, cores, C.byref(mapped), C.byref(jobs), C.byref(cells))
    probe_n = min(nreads, 2000 * cores)
    t = run(probe_n)
    count = int(min(nreads, max(probe_n, probe_n / max(t, 1e-6) * target_seconds)))
    t = run(count)
    return {"value": count / t, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the same batch through probe + ungapped filter + DP + traceback on the CPU oracle, "
                      "%d threads, %.1f s; %d DP jobs, %d reads mapped" % (count, cores, t, jobs.value, mapped.value)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step")
    ap.add_argument("--ref-len", type=int, default=0, help="reference length (default: E. coli K-12)")
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--max-sites", type=int, default=8)
    ap.add_argument("--scaffolds", type=int, default=1,
                    help="split the reference into this many chromosomes (large genomes: a chromosome must stay below 2^29 bases); "
                         "the CPU baseline and the parity sample are skipped when > 1")
    ap.add_argument("--repeat-frac", type=float, default=0.0, help="share of the reference drawn from repeat families (SURVEY 8d repeat model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=300)
    ap.add_argument("--no-iterations", action="store_true",
                    help="set BBMSA_NO_ITERATIONS on the DP jobs (scores and match strings unchanged, visited-cell counters not reported)")
    args = ap.parse_args()

    import torch
    from bbmap_amd import msa as M
    from bbmap_amd import workload as W
    from bbmap_amd.index import DeviceIndex
    from bbmap_amd.pipeline import MapPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    from bbmap_amd import dist as D0
    rank0 = rank
    read_len, k = 150, args.k
    ref_len = args.ref_len or W.ECOLI_K12_LEN
    if args.scaffolds > 1:
        # e.g. --ref-len 3100000000 --scaffolds 24: an hg38-sized reference; reads are drawn scaffold by scaffold
        per = ref_len // args.scaffolds
        chroms = [W.make_reference(per, seed=1000 + i, repeat_frac=args.repeat_frac, families=max(50, 2000 // args.scaffolds))
                  for i in range(args.scaffolds)]
        parts = [W.make_reads_and_jobs(c, args.reads // args.scaffolds + 1, read_len=150, seed=D0.shard_seed(2 + 7 * i, rank0))[0]
                 for i, c in enumerate(chroms)]
        reads_multi = np.concatenate(parts)[: args.reads * 150]
        ref = chroms[0]
    else:
        chroms, reads_multi = None, None
        ref = W.make_reference(ref_len, seed=1, repeat_frac=args.repeat_frac)
    # every rank draws its own shard of reads (same generator, different stream); the index is replicated per GPU
    from bbmap_amd import dist as D
    if reads_multi is not None:
        reads = reads_multi
        args.no_cpu_baseline = True
        args.parity_sample = 0
    else:
        reads, _, truth = W.make_reads_and_jobs(ref, args.reads, read_len=read_len, seed=D.shard_seed(2, rank))
    offsets = W.make_offsets(read_len, k, 1.9)
    key_scores = [100 * k] * len(offsets)          # GENERATE_KEY_SCORES_FROM_QUALITY needs qualities; synthetic reads have none
    max_sites, max_cols = args.max_sites, 256
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        # host-only work, done before this process touches the GPU
        chrom_bits = min(16, (32 - int(len(ref)).bit_length()) - 1)
        cpu = cpu_baseline(k, chrom_bits, reads, ref, read_len, offsets, key_scores, max_cols)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.reads
    t_ix = time.perf_counter()
    di = DeviceIndex.build(chroms if chroms is not None else [ref], k=k, device=local_rank)          # IndexMaker4 + analyzeIndex on the device (bbidx_build)
    torch.cuda.synchronize()
    t_ix = time.perf_counter() - t_ix
    hi = di.host
    pipe = MapPipeline(di, n, read_len, offsets, key_scores, device=local_rank, max_sites=max_sites, max_columns=max_cols,
                       no_iterations=args.no_iterations)
    pipe.load_reads(reads)

    for _ in range(args.warmup):
        pipe.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step(sync=False)                   # everything of a step is enqueued on the stream: no host round trip inside
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, pipe.dev)

    # per-kernel durations of the last timed step (HIP events the library recorded on the launch stream) and its counters
    njobs = pipe.counts()[0]
    probe_stats, pms = pipe.probe_stats()
    probe_ms = [pms]
    k3 = pipe.msa.last_kernel_ms3()
    narrow_ms, wave_ms, dp_ms = [k3[0]], [k3[1]], [k3[0] + k3[1] + k3[2]]
    gapped_ms = [sum(pipe.msa_gapped.last_kernel_ms3()) if pipe.last_counters[2] else 0.0]

    # ---- outside the timed region: checks and bookkeeping
    out = pipe.fetch(njobs)
    cnt = pipe.last_counters
    res = out["results"]
    dp_ok = np.zeros(n, bool)
    if njobs:
        np.logical_or.at(dp_ok, out["src"] // max_sites, res["score_len"] > 0)
    mapped = int(((out["nsites"] > 0) & ((out["no_indel"].max(axis=1) >= 70 + 149 * 100 - 495) | dp_ok)).sum())
    cells = int(res["iterations"].sum()) if njobs else 0
    ngapped = int(cnt[2])
    gcells = int(out["gresults"]["iterations"].sum()) if ngapped else 0
    split = pipe.msa.last_counts() if njobs else {"narrow": 0, "narrow_left": 0, "wave": 0, "generic": 0}
    parity = None
    if rank == 0 and args.parity_sample > 0:
        parity = parity_sample(pipe, out, reads, ref, hi, offsets, key_scores, min(n, args.parity_sample), max_sites, max_cols)
        if parity["mismatches"]:
            raise SystemExit("parity check failed: %s" % parity)

    if rank == 0:
        total_reads = n * world * args.steps
        value = total_reads / elapsed
        d_ms, p_ms = float(np.mean(dp_ms)), float(np.mean(probe_ms))
        n_ms, w_ms, g_ms = float(np.mean(narrow_ms)), float(np.mean(wave_ms)), float(np.mean(gapped_ms))
        jobs = out["jobs"]
        dp_bytes = W.algorithmic_bytes(jobs) if njobs else 0
        # the wavefront kernel's own share of those bytes: the jobs the narrow-window kernel did not finish
        wave_bytes = int(dp_bytes * (split["wave"] / max(1, njobs)))
        nkeys = len(offsets)
        # 2 strands x nkeys x (8 + 8) + 2 x 4 x (list entries streamed) + ref bytes compared + 64 x sites out
        probe_bytes = n * 2 * nkeys * 16 + 4 * (probe_stats[0] + probe_stats[1]) + probe_stats[3] + 64 * probe_stats[4]
        if w_ms >= p_ms:
            dom, dom_ms, dom_bytes = "msa_fill_fast_kernel", w_ms, wave_bytes
        else:
            dom, dom_ms, dom_bytes = "probe_wave_kernel", p_ms, probe_bytes
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom + "_bytes_per_launch")
                vi = tj.get("valu_issue", {})
                if dom in vi:       # the bound that actually binds these integer kernels (from the committed PMC profile, not live)
                    valu = {"source": "profiles/traffic.json (rocprofv3 SQ_INSTS_VALU)", "frac_of_valu_issue_peak": vi[dom]["frac"],
                            "peak_wave_inst_per_s": vi.get("peak_wave_inst_per_s")}
            except Exception:
                traffic, valu = None, None
        out_json = {
            "metric": "aligned_reads_per_sec", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("configs[1]: synthetic E. coli K-12 sized reference" if ref_len == W.ECOLI_K12_LEN and args.repeat_frac == 0
                                    else "synthetic reference, repeat fraction %g" % args.repeat_frac) + " (%d bp, seed 1), %d x %d-bp SE reads per GPU "
                                   "(seed 2, mutated mix), k=%d index resident in HBM; per step: reverse complement -> index probe "
                                   "(BBIndex.findAdvanced) -> ungapped site filter -> slow-align DP + traceback for the sites that "
                                   "need it (gapped-reference DP for sites with gap arrays)" % (ref_len, n, read_len, k),
                       "reads_per_gpu_per_step": n, "read_len": read_len, "keys_per_read": nkeys,
                       "dp_jobs_per_step": njobs, "dp_jobs_by_kernel": split, "gapped_dp_jobs_per_step": ngapped,
                       "gapped_dp_cells_per_step": gcells,
                       "reads_finished_without_dp": int(cnt[1]), "reads_without_site": int(cnt[3]),
                       "mapped_fraction": mapped / n, "dp_cells_per_step": cells,
                       "dp_gcups": (cells / (d_ms * 1e-3) / 1e9) if d_ms > 0 else 0.0,
                       "probe_list_entries_per_step": int(probe_stats[0] + probe_stats[1]),
                       "probe_extend_calls_per_step": int(probe_stats[2]),
                       "index_build_s_gpu": t_ix, "parity": parity},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom, "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": int(dom_bytes), "valu_issue": valu,
                         "kernels": {"probe_wave_kernel": {"ms": p_ms, "algorithmic_bytes": int(probe_bytes)},
                                     "msa_fill_fast_kernel": {"ms": w_ms, "algorithmic_bytes": int(wave_bytes)},
                                     "msa_fill_narrow_kernel": {"ms": n_ms, "algorithmic_bytes": int(dp_bytes - wave_bytes)},
                                     "gapped_dp_kernels": {"ms": g_ms}}},
        }
        if cpu is not None:
            out_json["cpu_baseline"] = cpu
        print(json.dumps(out_json))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
