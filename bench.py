#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the MI355X seed-and-extend hot path.

A "step" is one pass of the hot path over one batch of synthetic reads that is already resident in HBM:
index probe (BBIndex.findAdvanced) for both mates -> mate pairing and list trimming -> ungapped scores ->
scoreSlow (affine-gap DP, traceback) in rounds -> rescue of the unpaired mate (quickRescue scan + slowRescue DP),
everything on the device (bbmap_map_batch_device).

Default workload (BASELINE.json `metric`: 2x150 bp vs hg38; configs[3], one GPU's shard): a synthetic hg38-shaped
reference (24 chromosomes with the GRCh38 lengths, 3.09 Gbp, 10 % repeat families, seed 38+i), k=13 index built on the
device and resident in HBM, 1,000,000 synthetic read pairs (2 x 150 bp, opposite strands, mutated mix) per GPU and step.
With N > 1 every rank maps its own shard of pairs against its own replica of the index (weak scaling, no collective on
the data path).  `--workload ecoli` is configs[1] (single-ended), `--workload chr21` configs[2].

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how each field is obtained.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

# GRCh38 primary assembly chromosome lengths (chr1..22, X, Y)
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]

LEAD_N = {"chr21": [6600000]}      # undefined bases at the start of each chromosome (chr21's 6.6 Mbp leading N-run)

WORKLOADS = {
    # name: (chromosome lengths, paired, description)
    "hg38": (HG38, True, "configs[3], one GPU's shard: synthetic hg38-shaped reference (24 chromosomes with the GRCh38 lengths, "
                         "%d bp, 10 %% repeat families, seeds 38+i)" % sum(HG38)),
    "chr21": ([46709983], True, "configs[2]: synthetic chr21-sized reference (46709983 bp of which the first 6.6 Mbp are N as on the real "
                                "chromosome, 10 % repeat families, seed 38)"),
    "ecoli": ([4641652], False, "configs[1]: synthetic E. coli K-12 sized reference (4641652 bp, seed 38)"),
}


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


def log(msg):
    if os.environ.get("BENCH_VERBOSE"):
        print("[bench %6.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def shared_reference(name, lens, repeat_frac, local_rank, world_local):
    """The synthetic chromosomes.  With several ranks on a node, local rank 0 generates them once into /dev/shm and the
    others map that file (a 3 Gbp reference per rank would be 8 x the host memory and 8 x the time)."""
    from bbmap_amd import workload as W

    def generate():
        fam = max(50, 2000 // len(lens))
        lead = LEAD_N.get(name, [0] * len(lens))
        return [W.make_reference(n, seed=38 + i, repeat_frac=repeat_frac, families=fam, lead_n=lead[i]) for i, n in enumerate(lens)]
    if world_local <= 1:
        return generate(), None
    total = sum(n + 2 * W.START_PAD for n in lens)
    path = "/dev/shm/bbmap_amd_ref_%s_%d_%d.u8" % (name, total, os.getppid())
    if local_rank == 0:
        chroms = generate()
        mm = np.lib.format.open_memmap(path + ".tmp", mode="w+", dtype=np.uint8, shape=(total,))
        o = 0
        for c in chroms:
            mm[o:o + len(c)] = c
            o += len(c)
        mm.flush()
        del mm
        os.rename(path + ".tmp", path)
        return chroms, path
    t = time.time()
    while not os.path.exists(path):
        if time.time() - t > 900:
            raise SystemExit("timed out waiting for rank 0's reference in " + path)
        time.sleep(0.5)
    mm = np.load(path, mmap_mode="r")
    chroms, o = [], 0
    for n in lens:
        ln = n + 2 * W.START_PAD
        chroms.append(np.asarray(mm[o:o + ln]))
        o += ln
    return chroms, None


DEL_MODEL = "randomreads"      # deletions as sh/randomreads.sh draws them (1..400 bases); "short" = rounds 1-2's cap of 40


def make_batch(chroms, n_reads, paired, seed, read_set="mutated", lead=None, with_truth=False):
    """n_reads reads (n_reads / 2 pairs) drawn from the chromosomes in proportion to their lengths.  read_set: "mutated" = the
    reference generator's commented-out stress mix (snp .4 / ins .2 / del .2 / n .2 of the imperfect half, perfect .5,
    current/align2/RandomReads3.java:74-79) plus 3 % hard mates; "default" = sh/randomreads.sh's defaults, every rate 0: reads are
    exact copies of the reference (its quality-driven substitution errors are not modelled: the reads carry no qualities).
    with_truth: also the generator's coordinates per unit, in batch order -- {"chrom" (1-based), "start1", "strand1"[, "start2",
    "strand2"]} (single-ended reads are drawn from the plus strand: strand1 = 0)."""
    from bbmap_amd import workload as W
    L = 150
    units = n_reads // 2 if paired else n_reads
    lens = np.array([len(c) for c in chroms], np.float64)
    share = np.floor(units * lens / lens.sum()).astype(np.int64)
    share[0] += units - share.sum()
    parts, tparts = [], []
    pf, hard = (0.5, 0.03) if read_set == "mutated" else (1.0, 0.0)
    for i, (c, m) in enumerate(zip(chroms, share)):
        if m <= 0:
            continue
        lo = lead[i] if lead else 0
        if paired:
            rd, t = W.make_pairs(c, int(m), read_len=L, seed=seed + 7 * i, del_model=DEL_MODEL, perfect_frac=pf, hard_frac=hard, lo=lo)
            parts.append(rd.reshape(-1, 2 * L))
            tparts.append(np.stack([np.full(int(m), i + 1, np.int64), t["start1"], t["strand1"], t["start2"], t["strand2"]], axis=1))
        else:
            rd, _, t = W.make_reads_and_jobs(c, int(m), read_len=L, seed=seed + 7 * i, del_model=DEL_MODEL, perfect_frac=pf, lo=lo)
            parts.append(rd.reshape(-1, L))
            tparts.append(np.stack([np.full(int(m), i + 1, np.int64), t["start"], np.zeros(int(m), np.int64)], axis=1))
    allp = np.concatenate(parts)
    perm = np.random.Generator(np.random.PCG64(seed)).permutation(len(allp))      # mix the chromosomes within the batch
    reads = np.ascontiguousarray(allp[perm]).reshape(-1)
    if not with_truth:
        return reads
    tt = np.concatenate(tparts)[perm]
    truth = {"chrom": tt[:, 0], "start1": tt[:, 1], "strand1": tt[:, 2]}
    if paired:
        truth.update(start2=tt[:, 3], strand2=tt[:, 4])
    return reads, truth


def oracle_index(di, chroms, k):
    """The CPU oracle's probe over the device-built index arrays, exported block by block (tests/test_index_gpu.py shows the
    device build equal to the oracle's own builder array by array)."""
    from oracle.oracle import OracleIndexView
    blocks = [di.export_block(b) for b in range(di.host.nblocks)]
    return OracleIndexView(chroms, k, di.host.chromBits, di.host.params, blocks)


def cpu_baseline(oi, reads, L, paired, offsets, key_scores, target_seconds=20.0):
    """The same per-read flow on the host cores, from the CPU oracle (oracle/mapper_oracle.c: a port of the reference's
    logic): probe + pairing + ungapped scores + scoreSlow DP + rescue + the final alignment stage (genMatchString -> realign_new,
    match strings), one worker thread per usable core sharing one read-only index, on a bounded sample of the same batch."""
    from oracle.oracle import map_batch
    cores = usable_cores()
    r = reads.reshape(-1, L)
    units = len(r) // 2 if paired else len(r)

    def run(m):
        if paired:
            return map_batch(oi, r[0:2 * m:2].copy(), r[1:2 * m:2].copy(), L, offsets, key_scores, cap=64, want_log=False, threads=cores)
        return map_batch(oi, r[:m].copy(), None, L, offsets, key_scores, cap=64, want_log=False, threads=cores)
    probe_n = min(units, 4000 * cores)            # (a first, short run sizes the sample: about 10-20 s of work on all cores)
    out = run(probe_n)
    count = int(min(units, max(probe_n, probe_n / max(out["seconds"], 1e-6) * target_seconds)))
    out = run(count)
    nreads = count * (2 if paired else 1)
    return {"value": nreads / out["seconds"], "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d %s of the same batch through probe + %sungapped scores + scoreSlow DP%s + final alignment stage (realign_new fills, match strings) on the CPU oracle "
                      "(index arrays shared with the device build), %d threads, %.1f s; %d fills, %d visited cells, %d rescue scans, "
                      "%d reads with a site" % (count, "pairs" if paired else "reads", "pairing + " if paired else "",
                                                " + rescue" if paired else "", cores, out["seconds"], out["stats"][0], out["stats"][1],
                                                out["stats"][2], out["stats"][3])}


def parity_sample(mp, out, oi, reads, L, paired, offsets, key_scores, count):
    """Checks the first `count` reads of the last step end to end against the oracle (site lists, fills, match strings)."""
    from oracle.oracle import map_batch
    from tests.mapper_check import compare
    r = reads.reshape(-1, L)
    if paired:
        count -= count % 2
        orc = map_batch(oi, r[0:count:2].copy(), r[1:count:2].copy(), L, offsets, key_scores, cap=1024)
    else:
        orc = map_batch(oi, r[:count].copy(), None, L, offsets, key_scores, cap=1024)
    over = [i for i in range(count) if out["nsites"][i] in (-1, -2)]        # fitted neither max_sites nor the overflow tier
    good = [i for i in range(count) if out["nsites"][i] >= 0 or out["nsites"][i] == -3]
    bad = compare(out, orc, count, paired, reads_range=good)
    return {"checked_reads": count, "mismatches": len(bad), "overflowed_in_sample": len(over), "first": bad[:3]}


_HIP = None


def _hip_copy(dst, src, nbytes, kind, stream):
    """hipMemcpyAsync on raw pointers (kind 1 = host to device, 2 = device to host, 3 = device to device)."""
    global _HIP
    import ctypes as C
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _HIP.hipMemcpyAsync.restype = C.c_int
    if nbytes > 0:
        rc = _HIP.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), nbytes, kind, C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("hipMemcpyAsync failed (%d)" % rc)


def streaming_region(mp, batches, steps, warmup):
    """The same step with a DIFFERENT batch every time and the PCIe traffic a host would have inside the timed region: batch
    i + 1 is uploaded (pinned host memory, second stream) while batch i is mapped, and batch i's results -- the site lists without
    their empty slots (bbmap_pack_sites_device), the per-read counts, both fill logs with their traceback strings, the final records
    and their match strings -- are copied to pinned host memory while batch i + 1 is mapped.  Returns (seconds for `steps` steps, bytes up per step, bytes down per step)."""
    import torch
    dev, n, total = mp.dev, mp.n, mp.total_bytes
    pin_in = [torch.from_numpy(np.ascontiguousarray(b).reshape(-1)).pin_memory() for b in batches]
    dev_in = [torch.zeros(2 * total, dtype=torch.uint8, device=dev) for _ in range(2)]
    cap_rec = int(n * 5)
    counts = [torch.zeros(n + 1, dtype=torch.int32, device=dev) for _ in range(2)]
    offsets = [torch.zeros(n + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    packed = [torch.zeros(cap_rec * 128, dtype=torch.uint8, device=dev) for _ in range(2)]
    o = None
    mp.step()                                                     # sizes of the log staging buffers
    o = mp.output_pointers()
    job_cap, gjob_cap = int(o.n_jobs * 1.3) + 4096, int(o.n_gapped_jobs * 1.5) + 4096
    per_job, per_gjob = 40 + 80 + 16 + o.match_stride, 40 + 80 + 16 + 68 + o.gmatch_stride
    final_cap = (n * 64 + int(o.final_match_bytes * 1.3) + 65536) if o.final else 0
    stage = [torch.zeros(job_cap * per_job + gjob_cap * per_gjob + final_cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    host_out = [torch.zeros(4 * (n + 1) + cap_rec * 128 + stage[0].numel(), dtype=torch.uint8).pin_memory() for _ in range(2)]
    main = torch.cuda.current_stream()
    copy = torch.cuda.Stream(device=dev)
    h2d_done = [torch.cuda.Event() for _ in range(2)]
    out_ready = [torch.cuda.Event() for _ in range(2)]
    up = down = 0

    def upload(slot, b):
        _hip_copy(dev_in[slot].data_ptr(), pin_in[b % len(pin_in)].data_ptr(), total, 1, copy.cuda_stream)
        h2d_done[slot].record(copy)

    upload(0, 0)
    upload(1, 1)
    t0 = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            up = down = 0
        s = i % 2
        main.wait_event(h2d_done[s])
        mp.step(bases=dev_in[s])
        mp.pack_sites(counts[s], offsets[s], packed[s])
        o = mp.output_pointers()
        nj, ng = int(o.n_jobs), int(o.n_gapped_jobs)
        if nj > job_cap or ng > gjob_cap:
            raise RuntimeError("streaming_region: log staging buffers too small")
        # the logs live in the mapper's buffers, which the next step overwrites: staged device to device, then sent from the stage
        pieces = [(o.jobs, nj * 40), (o.results, nj * 80), (o.jobinfo, nj * 16), (o.match, nj * o.match_stride),
                  (o.gjobs, ng * 40), (o.gresults, ng * 80), (o.gjobinfo, ng * 16), (o.ggaps, ng * 68), (o.gmatch, ng * o.gmatch_stride)]
        if o.final:                                                # the final records (what BBMap prints) and their match strings
            if n * 64 + int(o.final_match_bytes) > final_cap:
                raise RuntimeError("streaming_region: final staging buffer too small")
            pieces += [(o.final, n * 64), (o.final_match, int(o.final_match_bytes))]
        off = 0
        for ptr, nb in pieces:
            _hip_copy(stage[s].data_ptr() + off, ptr, nb, 3, main.cuda_stream)
            off += nb
        total_sites = int(offsets[s][n].item())                    # (waits for the pack; the step itself is over)
        if total_sites > cap_rec:
            raise RuntimeError("streaming_region: packed site buffer too small")
        out_ready[s].record(main)
        copy.wait_event(out_ready[s])
        h = host_out[s].data_ptr()
        _hip_copy(h, counts[s].data_ptr(), 4 * n, 2, copy.cuda_stream)
        _hip_copy(h + 4 * (n + 1), packed[s].data_ptr(), total_sites * 128, 2, copy.cuda_stream)
        _hip_copy(h + 4 * (n + 1) + cap_rec * 128, stage[s].data_ptr(), off, 2, copy.cuda_stream)
        down += 4 * n + total_sites * 128 + off
        upload(s, i + 2)                                           # the batch after next goes into the buffer this step has freed
        up += total
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, up // max(1, steps), down // max(1, steps)


def pacbio_dp_main(args):
    """--workload pacbio_dp: the DP half of BASELINE.json configs[4] (mapPacBio: 10 kb reads pre-split into <= 6,000-base pieces,
    fastareadlen=6000, current/align2/BBMapPacBio.java:47-69).  A step = every piece of the batch aligned with the
    MultiStateAligner9PacBio scheme against its window +- padding (fillAndScoreLimited + traceback) by the strip-tiled wavefront
    kernel.  The PacBio INDEX probe (BBIndexPacBio's constants, 6,000-base reads) is not built, so this is not that configuration
    end to end and its metric is its own (pieces per second), never the headline's."""
    import threading
    import torch
    from bbmap_amd import msa as M
    n = args.reads if args.reads != 2000000 else 2000
    Lp = 6000
    rng = np.random.Generator(np.random.PCG64(5))
    BASES = np.frombuffer(b"ACGT", np.uint8)
    genome = BASES[rng.integers(0, 4, size=4_000_000, dtype=np.uint8)]
    reads, jobs, off = [], np.zeros(n, M.JOB_DTYPE), 0
    for i in range(n):
        s_ = int(rng.integers(1000, len(genome) - Lp - 2000))
        err = rng.uniform(0.13, 0.17)                                          # pbmin / pbmax (RandomReads3.java:1714-1715)
        x = rng.random(Lp)
        keep = x >= err * 0.35
        sub = (x >= err * 0.35) & (x < err * 0.55)
        piece = genome[s_:s_ + Lp].copy()
        piece[sub] = BASES[rng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]
        piece = piece[keep]
        ins_at = np.nonzero(rng.random(len(piece)) < err * 0.45)[0]
        piece = np.insert(piece, ins_at, BASES[rng.integers(0, 4, size=len(ins_at), dtype=np.uint8)])[:6019]
        reads.append(piece)
        jobs[i] = (off, 0, len(piece), len(genome), s_ - 40, s_ + Lp + 40, int(0.3 * (90 + 100 * (len(piece) - 1))),
                   M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK)
        off += len(piece)
    blob = np.concatenate(reads)
    log("pieces ready")
    # CPU baseline: the oracle's restatement of the same scheme, one MSA per thread (549 MB matrix each), on a bounded sample
    cpu = None
    if not args.no_cpu_baseline:
        from oracle.oracle import OracleMSA
        cores = min(usable_cores(), 16)
        sample = min(n, 3 * cores)
        done = [0] * cores

        def work(t):
            om = OracleMSA(maxRows=6019, maxColumns=7600, scheme="9pacbio")
            for i in range(t, sample, cores):
                j = jobs[i]
                rd = blob[j["read_off"]: j["read_off"] + j["read_len"]].tobytes()
                r4 = om.fillLimited(rd, genome.tobytes(), int(j["refStartLoc"]), int(j["refEndLoc"]), int(j["minScore"]))
                if r4 is not None:
                    om.traceback(rd, genome.tobytes(), int(j["refStartLoc"]), int(j["refEndLoc"]), r4[0], r4[1], r4[2])
                    done[t] += 1
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.perf_counter() - t0
        cpu = {"value": sample / dt, "unit": "pieces/s", "cores": cores, "kind": "port",
               "sample": "first %d pieces of the same batch, fillLimited + traceback on the CPU oracle (9PacBio constants), %d threads, %.1f s; %d aligned" % (sample, cores, dt, sum(done))}
        log("cpu baseline done")
    torch.cuda.set_device(0)
    ctx = M.MSAContext(maxRows=6019, maxColumns=7600, scheme=M.SCHEME_9PACBIO)
    stride = 6019 + 7600 + 64
    for _ in range(max(1, args.warmup)):
        res, match = ctx.align_batch(jobs, blob, genome, match_stride=stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(args.steps):
        res, match = ctx.align_batch(jobs, blob, genome, match_stride=stride)
        kms += ctx.last_kernel_ms3()[1]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cells = int(res["iterations"].sum())
    swept = float((jobs["read_len"].astype(np.int64) * (jobs["refEndLoc"] - jobs["refStartLoc"] + 1)).sum())
    kms /= args.steps
    algo = int((2 * (jobs["read_len"].astype(np.int64) + (jobs["refEndLoc"] - jobs["refStartLoc"] + 1)) + 59).sum())
    out = {"metric": "pacbio_dp_pieces_per_sec", "value": n * args.steps / elapsed, "unit": "pieces/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32", "data": "synthetic",
           "config": {"workload": "DP stage of configs[4] only (the PacBio index probe is not built): %d pieces of <= 6,000 bases with 13-17 %% PacBio errors, "
                                  "each aligned against its window of 6,080 columns with the MultiStateAligner9PacBio scheme (fillAndScoreLimited + "
                                  "traceback); the timed region includes the host-to-device and device-to-host copies of the batch" % n,
                      "pieces_per_step": n, "aligned": int((res["score_len"] > 0).sum()), "visited_cells_per_step": cells,
                      "gcups_visited": cells / (kms * 1e-3) / 1e9, "gcups_swept": swept / (kms * 1e-3) / 1e9, "strip_kernel_ms": kms,
                      "handed_to_generic_kernel_ms": ctx.last_kernel_ms3()[2]},
           "roofline": {"bound": "hbm", "achieved": algo / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "msa_fill_strip_kernel",
                        "kernel_ms": kms, "algorithmic_bytes_per_launch": algo,
                        "note": "integer-VALU-bound (about 180 VALU instructions per cell): the HBM fraction is reported as the contract asks, the rate that says something is gcups"}}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))


def pacbio_main(args):
    """--workload pacbio: BASELINE.json configs[4] end to end -- mapPacBio.sh's classes (BBIndexPacBio, BBMapThreadPacBio,
    MultiStateAligner9PacBio; BBMapPacBio.setDefaults, current/align2/BBMapPacBio.java:47-69) on the hg38-shaped reference:
    10 kb PacBio-like reads (13-17 % errors) pre-split at fastareadlen = 6000 into pieces of 6,000 and 4,000 bases, keys placed as
    quickMap places them (density floor 2.8: 1,400 / 934 keys per piece).  A step = bbmap_map_batch_device over the resident
    batch of pieces: probe (long-read kernel) -> trimList -> ungapped scores / tip search -> scoreSlow fills + traceback (strip-tiled
    wavefront kernel).  One GPU per rank, pieces shard per rank, index replicated (no collective)."""
    import torch
    from bbmap_amd import dist as D
    from bbmap_amd import keys as K
    from bbmap_amd import workload as W
    from bbmap_amd.index import DeviceIndex, PROFILE_PACBIO
    from bbmap_amd.mapper import Mapper
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    genome = args.pacbio_genome
    lens = WORKLOADS[genome][0]
    n = args.reads if args.reads != 2000000 else 8192                  # pieces per GPU and step (4,096 reads of 10 kb: large enough that the
                                                                       # late scoreSlow rounds, a lone fill's latency each, stay a small share)
    n -= n % 2
    world_local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    chroms, shm_path = shared_reference(genome, lens, 0.0 if genome == "ecoli" else 0.1, local_rank, world_local)
    log("reference ready")
    seed = D.shard_seed(5, rank)
    a, ta = W.make_pacbio_pieces(chroms, n // 2, seed=seed, min_len=6000, max_len=6000)
    b, tb = W.make_pacbio_pieces(chroms, n // 2, seed=seed + 1, min_len=4000, max_len=4000)
    pieces = [x for pair in zip(a, b) for x in pair]                   # piece 2i = the first 6,000 bases of read i, 2i+1 = its last 4,000
    kcfg = K.default_config(K.PROFILE_PACBIO)
    recs, blob, bs, keyinfo = K.make_batch(pieces, None, kcfg)
    log("pieces and keys ready")
    rehearse = os.environ.get("BBMAP_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if rehearse else "nccl", **({} if rehearse else {"device_id": torch.device("cuda", dev_index)}))
    red_dev = None if rehearse else torch.device("cuda", dev_index)
    t_lib = _first_library_call(dev_index)
    t_ix = time.perf_counter()
    di = DeviceIndex.build(chroms, device=dev_index, profile=PROFILE_PACBIO)
    torch.cuda.synchronize()
    t_ix = time.perf_counter() - t_ix
    log("index built")
    k = di.host.k
    oi, cpu = None, None
    want_cpu = rank == 0 and not args.no_cpu_baseline
    if rank == 0 and (args.parity_sample > 0 or want_cpu):
        from oracle.oracle import OracleIndexView, map_reads, map_default_params
        blocks = [di.export_block(bk) for bk in range(di.host.nblocks)]
        oi = OracleIndexView(chroms, k, di.host.chromBits, di.host.params, blocks, profile="pacbio")
        log("index exported")
    if want_cpu:
        cores = min(usable_cores(), 16)                               # one 549 MB matrix per worker thread (MSA(6020, 7600))
        m = min(n, 2 * cores)
        o1 = map_reads(oi, recs[:m], blob, keyinfo, base_scores=bs, cap=64, want_log=False, threads=cores)
        m2 = int(min(n, max(m, m / max(o1["seconds"], 1e-6) * 15.0)))
        m2 -= m2 % 2
        o2 = map_reads(oi, recs[:m2], blob, keyinfo, base_scores=bs, cap=64, want_log=False, threads=cores) if m2 > m else o1
        m2 = max(m2, m) if m2 > m else m
        cpu = {"value": (m2 / 2) / o2["seconds"], "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": "first %d pieces (%d reads) of the same batch through probe + trimList + ungapped scores + scoreSlow DP on the CPU "
                         "oracle compiled with mapPacBio's constants (index arrays shared with the device build), %d threads, %.1f s; %d fills, "
                         "%d visited cells, %d pieces with a site" % (m2, m2 // 2, cores, o2["seconds"], o2["stats"][0], o2["stats"][1], o2["stats"][3])}
        log("cpu baseline done")
    mp = Mapper.from_records(di, recs, blob, bs, keyinfo, paired=False, device=dev_index, max_sites=args.max_sites, profile=PROFILE_PACBIO)
    for _ in range(args.warmup):
        mp.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        mp.step()
        for key, v in mp.stats().items():
            if key.startswith("ms_"):
                acc[key] = acc.get(key, 0.0) + v
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, red_dev)
    log("timed region done")
    st = mp.stats()
    ms = {key: v / max(1, args.steps) for key, v in acc.items()}
    out = mp.fetch(with_match=args.parity_sample > 0 and rank == 0)
    parity = None
    if rank == 0 and args.parity_sample > 0:
        from tests.mapper_check import compare
        cnt = min(n, max(2, args.parity_sample if args.parity_sample < 600 else 32))        # 32 pieces: 3-4 s of the CPU oracle on 16 threads
        orc = map_reads(oi, recs[:cnt], blob, keyinfo, base_scores=bs, cap=1024, threads=min(usable_cores(), 8))
        good = [i for i in range(cnt) if out["nsites"][i] >= 0 or out["nsites"][i] == -3]
        bad = compare(out, orc, cnt, False, reads_range=good)
        parity = {"checked_pieces": cnt, "mismatches": len(bad), "first": bad[:3]}
        if bad:
            raise SystemExit("parity check failed: %s" % parity)
    if rank == 0:
        nsites = out["nsites"]
        lens_p = recs["len"].astype(np.int64)
        maxq = 90 + (lens_p - 1) * 100
        top = out["sites"][:, 0]
        mapped = int(((nsites > 0) & (top["slowScore"] >= (np.float32(0.46) * maxq.astype(np.float32)).astype(np.int64))).sum())
        near = 0
        truth = np.empty((n, 4), np.int64)
        truth[0::2], truth[1::2] = ta, tb
        for i in range(n):
            if nsites[i] > 0:
                near += int(top["chrom"][i] == truth[i][0] and top["strand"][i] == truth[i][1] and abs(int(top["start"][i]) - int(truth[i][2])) < 500)
        cells = int(out["results"]["iterations"].sum() + out["gresults"]["iterations"].sum())
        ps = st["probe_stats"]
        nkeys_total = int(recs["nkeys"].astype(np.int64).sum())
        probe_bytes = 2 * nkeys_total * 16 + 4 * (ps[0] + ps[1]) + ps[3] + 64 * ps[4]
        jobs_all = np.concatenate([out["jobs"], out["gjobs"]]) if len(out["gjobs"]) else out["jobs"]
        dp_bytes = int((2 * (jobs_all["read_len"].astype(np.int64) + (jobs_all["refEndLoc"] - jobs_all["refStartLoc"] + 1)) + 59).sum()) if len(jobs_all) else 0
        dp_ms = ms["ms_dp_wave"] + ms["ms_dp_gapped"]
        kern = {"probe_long_kernel": {"ms": ms["ms_probe"], "algorithmic_bytes": int(probe_bytes)},
                "msa_fill_strip_kernel": {"ms": dp_ms, "algorithmic_bytes": dp_bytes},
                "msa_fill_generic_kernel(hand-overs)": {"ms": ms["ms_dp_generic"]},
                "mapper_glue(begin+score+finish kernels)": {"ms": ms["ms_begin"] + ms["ms_score"] + ms["ms_finish"]}}
        dom = "probe_long_kernel" if ms["ms_probe"] >= dp_ms else "msa_fill_strip_kernel"
        dom_ms, dom_bytes = kern[dom]["ms"], kern[dom]["algorithmic_bytes"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        reads_per_step = n // 2
        traffic, traffic_src = None, None                       # counter traffic of a committed profile of this very workload, if there is one
        try:
            ent = json.load(open(os.path.join(ROOT, "profiles", "traffic_r04.json"))).get("pacbio", {}).get(dom)
            if ent and ent.get("pieces_per_step") == n and args.pacbio_genome == "hg38":
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = "committed profile %s (rocprofv3 --pmc passes on this workload, not this run)" % ent["source"]
        except Exception:
            traffic = None
        out_json = {
            "metric": "aligned_reads_per_sec", "value": reads_per_step * world * args.steps / elapsed, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[4], one GPU's shard: mapPacBio mode (BBIndexPacBio k=%d, BBMapThreadPacBio, MultiStateAligner9PacBio) on the "
                                   "synthetic %s reference (%d bp); %d synthetic 10 kb PacBio-like reads (13-17 %% errors) per GPU and step, pre-split at "
                                   "fastareadlen=6000 into %d pieces of 6,000 / 4,000 bases with %d keys in all; per step: index probe -> trimList -> "
                                   "ungapped scores / tip search -> scoreSlow DP + traceback; a read = two pieces" % (
                                       k, genome, sum(lens), reads_per_step, n, nkeys_total),
                       "pieces_per_gpu_per_step": n, "pieces_per_sec": n * world * args.steps / elapsed, "max_sites": args.max_sites,
                       "fills_per_step": st["fills"] + st["gapped_fills"], "refills_per_step": st["refills"], "scoreslow_rounds": st["rounds"],
                       "pieces_without_site": st["reads_without_site"], "pieces_overflowed": st["reads_overflowed"],
                       "mapped_fraction": mapped / n, "pieces_whose_top_site_is_their_origin": near / n, "dp_cells_per_step": cells,
                       "dp_gcups_visited": (cells / (ms["ms_slow"] * 1e-3) / 1e9) if cells and ms["ms_slow"] > 0 else 0.0,
                       "probe_list_entries_per_step": int(ps[0] + ps[1]), "probe_extend_calls_per_step": int(ps[2]), "probe_stats_raw": [int(x) for x in ps],
                       "stage_ms": {key[3:]: round(v, 3) for key, v in ms.items()}, "index_build_s_gpu": t_ix, "library_first_call_s": t_lib, "parity": parity},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": dom, "kernel_ms": dom_ms,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "kernels": kern,
                         "note": "the long-read probe is latency-bound: 1.5 wavefronts per SIMD (36 KB of LDS each), 12.6 % of the VALU issue rate, waves "
                                 "waiting 53 % of their cycles, L2 hit rate 8 % (profiles/r04_pacbio_pmc_summary.txt); the strip DP is instruction-bound"}}
        if cpu is not None:
            out_json["cpu_baseline"] = cpu
        print(json.dumps(out_json))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if shm_path and os.path.exists(shm_path):
        os.remove(shm_path)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) before anything here touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def _first_library_call(dev_index):
    """The process's first call into libbbmap_amd.so, timed on its own: loading the shared object, its code objects (a few hundred kernel
    instantiations) and rocPRIM's sort kernels happens inside whichever call comes first -- on a fresh box 1.5-2 s, which round 3's
    `index_build_s_gpu` had absorbed (0.9 s here against 2.6 s in the driver's line).  A 4 kb reference is built and dropped."""
    import numpy as np
    import torch
    from bbmap_amd.index import DeviceIndex
    t = time.perf_counter()
    rng = np.random.default_rng(1)
    tiny = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 4096)].copy()
    DeviceIndex.build([tiny], k=13, device=dev_index).close()
    torch.cuda.synchronize()
    return time.perf_counter() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["pacbio", "pacbio_dp"], default="hg38")
    ap.add_argument("--pacbio-genome", choices=sorted(WORKLOADS), default="hg38", help="reference of --workload pacbio")
    ap.add_argument("--reads", type=int, default=2000000, help="reads per GPU per step (pairs x 2 in the paired workloads)")
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--max-sites", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=600)
    ap.add_argument("--stream-steps", type=int, default=10, help="steps of the PCIe-inclusive streaming region (0 = skip it)")
    ap.add_argument("--default-set-steps", type=int, default=3, help="steps timed on the 'default' (unmutated) read set after the main region (0 = skip)")
    args = ap.parse_args()

    if args.workload == "pacbio_dp":
        if args.gpus != 1 or "WORLD_SIZE" in os.environ:
            raise SystemExit("--workload pacbio_dp is a one-GPU DP benchmark")
        return pacbio_dp_main(args)
    if args.workload == "pacbio":
        return pacbio_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))

    import torch
    from bbmap_amd import dist as D
    from bbmap_amd import workload as W
    from bbmap_amd.index import DeviceIndex
    from bbmap_amd.mapper import Mapper

    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    lens, paired, desc = WORKLOADS[args.workload]
    L, k = 150, args.k
    n = args.reads - (args.reads % 2 if paired else 0)
    repeat_frac = 0.0 if args.workload == "ecoli" else 0.1
    world_local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    chroms, shm_path = shared_reference(args.workload, lens, repeat_frac, local_rank, world_local)
    log("reference ready")
    # every rank draws its own shard of reads (same generator, different stream); the index is replicated per GPU
    lead = LEAD_N.get(args.workload)
    reads = make_batch(chroms, n, paired, D.shard_seed(4, rank), lead=lead)
    log("reads ready")
    # Key offsets and scores as quickMap makes them for a read without qualities (AbstractMapThread.java:659-728 through the product's
    # bbkeys_make): all key error probabilities 0, density window floor 1.5 -> 18 keys for 150 bases, every key score
    # BASE_KEY_HIT_SCORE.  (Rounds 1-2 placed 22 keys with KeyRing.makeOffsets at density 1.9, the branch the reference only takes
    # for reads WITH qualities when GENERATE_KEY_SCORES_FROM_QUALITY is off.)
    from bbmap_amd import keys as K
    kcfg = K.default_config(K.PROFILE_BBMAP, k=k)
    offsets, key_scores, _ = K.make_keys(np.frombuffer(b"ACGT" * ((L + 3) // 4), np.uint8)[:L], None, kcfg)

    # BBMAP_BENCH_REHEARSE=1: every rank on GPU 0 and gloo instead of RCCL -- a way to run the N > 1 code path on a one-GPU box
    rehearse = os.environ.get("BBMAP_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    red_dev = None if rehearse else torch.device("cuda", dev_index)
    t_lib = _first_library_call(dev_index)
    t_ix = time.perf_counter()
    di = DeviceIndex.build(chroms, k=k, device=dev_index)          # IndexMaker4 + analyzeIndex on the device (bbidx_build)
    torch.cuda.synchronize()
    t_ix = time.perf_counter() - t_ix
    log("index built")
    oi, cpu = None, None
    # the CPU baseline runs on rank 0 whatever the world size, before the ranks' barrier (the others wait there)
    if rank == 0 and (args.parity_sample > 0 or not args.no_cpu_baseline):
        oi = oracle_index(di, chroms, k)
        if paired:
            oi.s.p.quitAfterTwoPerfects = 0
        log("index exported")
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(oi, reads, L, paired, offsets, key_scores)
        log("cpu baseline done")

    mp = Mapper(di, n, L, offsets, key_scores, paired=paired, device=dev_index, max_sites=args.max_sites)
    mp.load_reads(reads)
    for _ in range(args.warmup):
        mp.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        mp.step()
        st = mp.stats()
        for key, v in st.items():
            if key.startswith("ms_"):
                acc[key] = acc.get(key, 0.0) + v
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, red_dev)
    log("timed region done")

    # ---- outside the timed region: checks and bookkeeping
    st = mp.stats()
    ms = {key: v / max(1, args.steps) for key, v in acc.items()}
    stream_res = None
    if args.stream_steps > 0:
        # the PCIe-inclusive rate: distinct batches, uploads and downloads inside the timed region (reported beside `value`)
        reads_b = make_batch(chroms, n, paired, D.shard_seed(5, rank), lead=lead)
        if dist is not None:
            dist.barrier()
        try:
            s_el, s_up, s_down = streaming_region(mp, [reads, reads_b], args.stream_steps, 1)
        except (RuntimeError, MemoryError) as e:                 # e.g. the host refuses that much pinned memory: the line goes out without it
            log("streaming region failed: %s" % e)
            s_el, s_up, s_down = float("inf"), 0, 0
        if dist is not None:
            s_el = D.max_over_ranks(s_el, dist, red_dev)
        stream_res = (s_el, s_up, s_down) if s_el != float("inf") else None
        mp.step()                                                # the parity sample below looks at batch `reads` again
        log("streaming region done")
    out = mp.fetch(with_match=args.parity_sample > 0 and rank == 0)
    nsites = out["nsites"]
    top = out["sites"][:, 0]
    minScore = int(np.float32(0.56) * np.float32(70 + 149 * 100))
    mapped = int(((nsites > 0) & (top["slowScore"] >= minScore)).sum())
    cells = int(out["results"]["iterations"].sum() + out["gresults"]["iterations"].sum())
    tier = out.get("overflow")
    if tier is not None:                                    # reads the overflow tier mapped (nsites == -3 in the main list)
        mapped += int(((tier["nsites"] > 0) & (tier["sites"][:, 0]["slowScore"] >= minScore)).sum())
        cells += int(tier["results"]["iterations"].sum() + tier["gresults"]["iterations"].sum())
    fin = out.get("final")
    final_info = None
    if fin is not None:                                     # the final records: what BBMap would print
        mapped = int((fin["mapped"] > 0).sum())
        final_info = {"mapped": mapped, "paired": int((fin["paired"] > 0).sum()), "ambiguous": int((fin["ambiguous"] > 0).sum()),
                      "perfect": int((fin["perfect"] > 0).sum()), "rescued": int(((fin["mapped"] > 0) & (fin["rescued"] > 0)).sum()),
                      "match_string_bytes": int(fin["match_len"].sum())}
    parity = None
    if rank == 0 and args.parity_sample > 0:
        parity = parity_sample(mp, out, oi, reads, L, paired, offsets, key_scores, min(n, args.parity_sample))
        if parity["mismatches"]:
            raise SystemExit("parity check failed: %s" % parity)
    # ---- the "default" read set (randomreads.sh defaults: unmutated reads) beside the mutated one: a second, shorter timed region
    default_res = None
    if args.default_set_steps > 0:
        reads_d = make_batch(chroms, n, paired, D.shard_seed(6, rank), read_set="default", lead=lead)
        mp.load_reads(reads_d)
        mp.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        td = time.perf_counter()
        for _ in range(args.default_set_steps):
            mp.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        d_el = time.perf_counter() - td
        if dist is not None:
            d_el = D.max_over_ranks(d_el, dist, red_dev)
        std = mp.stats()
        od = mp.fetch(with_match=False)
        topd = od["sites"][:, 0]
        default_res = {"value": n * world * args.default_set_steps / d_el, "unit": "reads/s", "steps": args.default_set_steps,
                       "ms_per_step": 1e3 * d_el / args.default_set_steps,
                       "mapped_fraction": float((od["final"]["mapped"] > 0).mean()) if "final" in od else float(((od["nsites"] > 0) & (topd["slowScore"] >= minScore)).mean()),
                       "perfect_fraction": float(((od["nsites"] > 0) & (topd["perfect"] != 0)).mean()),
                       "fills_per_step": int(std["fills"] + std["gapped_fills"]), "final_fills_per_step": int(std["final_fills"]), "rescue_scans_per_step": int(std["rescue_scans"]),
                       "stage_ms": {key[3:]: round(float(v), 3) for key, v in std.items() if key.startswith("ms_")},
                       "what": "same reference and batch size, reads as sh/randomreads.sh makes them by default (every mutation rate 0; "
                               "seed 6); stage_ms of the last step"}
        del reads_d, od
        log("default read set done")
    if rank == 0:
        from bbmap_amd import workload as W2
        total_reads = n * world * args.steps
        value = total_reads / elapsed
        nkeys = len(offsets)
        ps = st["probe_stats"]
        # 2 strands x nkeys x (8 + 8) + 2 x 4 x (list entries streamed) + ref bytes compared + 64 x sites out
        probe_bytes = n * 2 * nkeys * 16 + 4 * (ps[0] + ps[1]) + ps[3] + 64 * ps[4]
        dp_bytes = W2.algorithmic_bytes(out["jobs"]) if len(out["jobs"]) else 0
        kern = {"probe_wave_kernel": {"ms": ms["ms_probe"], "algorithmic_bytes": int(probe_bytes)},
                "msa_fill_fast_kernel": {"ms": ms["ms_dp_wave"], "algorithmic_bytes": int(dp_bytes)},
                "msa_fill_narrow_kernel": {"ms": ms["ms_dp_narrow"]},
                "second_dp_context(gapped refs, wide windows)": {"ms": ms["ms_dp_gapped"]},
                "quick_rescue_kernel": {"ms": ms["ms_quick_rescue"]},
                "mapper_glue(begin+score+finish kernels)": {"ms": ms["ms_begin"] + ms["ms_score"] + ms["ms_finish"]}}
        # the dominant kernel: the one with the longest single launch of a step.  The probe is ONE launch per step; the wavefront DP's
        # ms_dp_wave is the sum over a dozen launch sequences (rounds, rescue passes), each timed by events that also see the second
        # context's kernels running beside it; its longest pass is ms_dp_wave_max.
        kern["msa_fill_fast_kernel"]["longest_pass_ms"] = st["ms_dp_wave_max"]
        dom = "probe_wave_kernel" if ms["ms_probe"] >= st["ms_dp_wave_max"] else "msa_fill_fast_kernel"
        dom_ms, dom_bytes = kern[dom]["ms"], kern[dom]["algorithmic_bytes"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_r04.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(args.workload, {}).get(dom)
                if ent and ent.get("reads_per_step") == n:          # only a profile of this very workload counts
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_src = "committed profile %s (rocprofv3 --pmc passes on this workload, not this run)" % ent.get("source", tpath)
            except Exception:
                traffic = None
        out_json = {
            "metric": "aligned_reads_per_sec", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s; k=%d index built on the device, resident in HBM; %d x %d-bp %s per GPU and step (seed 4, "
                                   "mutated mix with deletions of 1..400 bases, 3 %% hard mates); per step: index probe (BBIndex.findAdvanced) -> %sungapped scores -> "
                                   "scoreSlow DP + traceback in rounds%s -> final alignment stage (final pairing, ambiguity policy, genMatchString -> realign_new DP in rounds, "
                                   "clipping, penalties: the printed start / stop / score and match string of every read)" % (
                                       desc, k, n // 2 if paired else n, L, "read pairs (2 x 150, opposite strands, insert 200-400)" if paired
                                       else "single-ended reads", "mate pairing + list trimming -> " if paired else "list trimming -> ",
                                       (" -> rescue (quickRescue scan + slowRescue DP) for unpaired mates" if paired else "") +
                                       ".  Not carried over from a real run: the adaptive state DYNAMIC_INSERT_LENGTH keeps per mapping thread "
                                       "(AVERAGE_PAIR_DIST follows the pairs seen so far, BBMapThread.java:1307-1309; here it stays at its initial "
                                       "100) and the 'mating is not working' switch that turns pairing off after many unpaired reads"),
                       "reads_per_gpu_per_step": n, "read_len": L, "paired": paired, "keys_per_read": nkeys, "max_sites": args.max_sites,
                       "fills_per_step": st["fills"] + st["gapped_fills"], "fills_second_context": st["gapped_fills"],
                       "final_fills_per_step": st["final_fills"], "final_rounds": st["final_rounds"], "final_reads_through_local_alignment": st["final_local"],
                       "final_records": final_info,
                       "refills_per_step": st["refills"], "scoreslow_rounds": st["rounds"], "fills_ahead_dropped_per_step": st["fills_dropped"],
                       "rescue_scans_per_step": st["rescue_scans"], "rescue_fills_per_step": st["rescue_fills"],
                       "reads_remapped_by_overflow_tier": st["reads_reprobed"], "reads_left_unmapped_by_overflow": st["reads_overflowed"], "reads_without_site": st["reads_without_site"],
                       "mapped_fraction": mapped / n, "dp_cells_per_step": cells,
                       # (the two DP contexts run side by side, so their kernel times overlap: the rate is over the two stages' wall time)
                       "dp_gcups_over_scoreslow_rescue_and_final_stages": (cells / ((ms["ms_slow"] + ms["ms_rescue"] + ms.get("ms_final", 0.0)) * 1e-3) / 1e9) if cells else 0.0,
                       "probe_list_entries_per_step": int(ps[0] + ps[1]), "probe_extend_calls_per_step": int(ps[2]),
                       "stage_ms": {key[3:]: round(v, 3) for key, v in ms.items()},
                       "index_build_s_gpu": t_ix, "library_first_call_s": t_lib, "parity": parity},
            "pcie_inclusive": None if stream_res is None else {
                "value": n * world * args.stream_steps / stream_res[0], "unit": "reads/s", "steps": args.stream_steps,
                "ms_per_step": 1e3 * stream_res[0] / args.stream_steps, "host_to_device_bytes_per_step": int(stream_res[1]),
                "device_to_host_bytes_per_step": int(stream_res[2]),
                "what": "two distinct batches alternating; batch i+1 uploaded from pinned host memory and batch i's results (packed site "
                        "lists, per-read counts, both fill logs with traceback strings) downloaded to pinned host memory on a second "
                        "stream while a batch is mapped; per GPU bytes"},
            "default_read_set": default_res,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dom, "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": int(dom_bytes), "kernels": kern,
                         "note": ("latency-bound gather kernel: 34 % of the VALU issue rate, waves waiting 53 % of their cycles at 5 waves per SIMD "
                                  "(profiles/r04_hg38_pmc_summary.txt); HBM is not what limits it" if dom == "probe_wave_kernel" else
                                  "integer DP: instruction-bound (8.6e10 VALU wave-instructions per step = 42 % of the VALU issue rate over the DP stages' "
                                  "wall time, four waves per SIMD of one dependent chain each; profiles/r04_hg38_pmc_summary.txt); its "
                                  "algorithmic bytes are a few hundred per fill, so an HBM fraction says nothing about it")},
        }
        if cpu is not None:
            out_json["cpu_baseline"] = cpu
        print(json.dumps(out_json))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if shm_path and os.path.exists(shm_path):
        os.remove(shm_path)


if __name__ == "__main__":
    main()
