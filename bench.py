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


def cpu_baseline(jobs, reads, ref, max_rows, max_cols, target_seconds=15.0):
    """Times the CPU oracle (a port of the reference's C + Java walkers) on a bounded sample."""
    from oracle import oracle as orc
    L = orc.lib()
    L.orc_bench_align.restype = C.c_double
    L.orc_bench_align.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    cores = usable_cores()
    jobs = np.ascontiguousarray(jobs)
    cells, chk = C.c_int64(), C.c_int64()

    def run(n):
        return L.orc_bench_align(jobs.ctypes.data, n, reads.ctypes.data, ref.ctypes.data, max_rows, max_cols,
                                 cores, 1, C.byref(cells), C.byref(chk))
    probe_n = min(len(jobs), 2000 * cores)
    t = run(probe_n)
    rate = probe_n / max(t, 1e-6)
    n = int(min(len(jobs), max(probe_n, rate * target_seconds)))
    t = run(n)
    return {"value": n / t, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the same job list, fillAndScoreLimited+traceback, %d threads, %.1f s"
                      % (n, cores, t),
            "cells_per_s": cells.value / t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step")
    ap.add_argument("--ref-len", type=int, default=0, help="reference length (default: E. coli K-12)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=2000)
    args = ap.parse_args()

    import torch
    from bbmap_amd import msa as M
    from bbmap_amd import workload as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    read_len = 150
    ref_len = args.ref_len or W.ECOLI_K12_LEN
    ref = W.make_reference(ref_len, seed=1)
    # every rank draws its own shard of reads (same generator, different stream)
    reads, jobs, truth = W.make_reads_and_jobs(ref, args.reads, read_len=read_len, seed=2 + 1000 * rank)
    cols = (jobs["refEndLoc"] - jobs["refStartLoc"] + 1)
    max_rows, max_cols = 160, 256
    assert int(cols.max()) <= max_cols
    match_stride = 352
    n = len(jobs)

    dev = torch.device("cuda", local_rank)
    d_ref = torch.from_numpy(ref).to(dev)
    d_reads = torch.from_numpy(reads).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_res = torch.zeros(n * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_match = torch.zeros(n * match_stride, dtype=torch.uint8, device=dev)
    ctx = M.MSAContext(maxRows=max_rows, maxColumns=max_cols, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.align_batch_device(n, d_jobs.data_ptr(), d_reads.data_ptr(), d_ref.data_ptr(), d_res.data_ptr(),
                               d_match.data_ptr(), match_stride, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, slow_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        a, b = ctx.last_kernel_ms()     # HIP events recorded on the launch stream around the kernels
        kernel_ms.append(a)
        slow_ms.append(b)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- outside the timed region: checks and bookkeeping
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE)
    aligned = int((res["score_len"] > 0).sum())
    cells = int(res["iterations"].sum())
    parity = None
    if rank == 0 and args.parity_sample > 0:
        from oracle.oracle import OracleMSA
        om = OracleMSA(max_rows, max_cols)
        refb = ref.tobytes()
        mt = d_match[: args.parity_sample * match_stride].cpu().numpy().reshape(-1, match_stride)
        bad = 0
        for k in range(min(n, args.parity_sample)):
            j = jobs[k]
            rd = reads[j["read_off"]: j["read_off"] + j["read_len"]].tobytes()
            sv, mx = om.fillAndScoreLimited(rd, refb, int(j["refStartLoc"]), int(j["refEndLoc"]), int(j["minScore"]))
            g = res[k]
            gs = None if g["score_len"] == 0 else g["score"][: g["score_len"]].tolist()
            ok = gs == sv
            if ok and sv is not None:
                tb = om.traceback(rd, refb, int(j["refStartLoc"]), int(j["refEndLoc"]), mx[0], mx[1], mx[2])
                ok = mt[k, : g["match_len"]].tobytes() == tb
            bad += (not ok)
        parity = {"checked": min(n, args.parity_sample), "mismatches": bad}
        if bad:
            raise SystemExit("parity check failed: %d of %d sample alignments differ from the oracle" % (bad, parity["checked"]))

    if rank == 0:
        total_reads = n * world * args.steps
        value = total_reads / elapsed
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = W.algorithmic_bytes(jobs)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("msa_fill_fast_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "aligned_reads_per_sec", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic E. coli K-12 sized reference (%d bp, seed 1), %d x %d-bp SE reads "
                                   "per GPU (seed 2, mutated mix); slow-align DP stage only: one fillAndScoreLimited + "
                                   "traceback per read at its candidate site (index probe not yet on the GPU)" % (ref_len, n, read_len),
                       "reads_per_gpu_per_step": n, "read_len": read_len, "mean_columns": float(cols.mean()),
                       "aligned_fraction": aligned / n, "dp_cells_per_step": cells,
                       "gcups": cells / (k_ms * 1e-3) / 1e9, "parity": parity},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "msa_fill_fast_kernel", "kernel_ms": k_ms, "generic_kernel_ms": float(np.mean(slow_ms)),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(jobs, reads, ref, max_rows, max_cols)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
