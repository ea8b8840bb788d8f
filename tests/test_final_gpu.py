"""GPU tests of the final alignment stage of the device mapper (bbmap_amd/csrc/mapper_final.h: genMatchString -> genMatchStringForSite ->
realign_new, fixXY / clipTipIndels / toLocalAlignment, final pairing, penalties) against its CPU restatement oracle/final_stage.inc:
the per-read final records (what BBMap prints), the match strings byte for byte, the site lists after the stage and every fill.
The whole-flow comparison on ordinary reads is in test_mapper_gpu.py / test_golden_phix.py (both sides run the stage by default);
here the stage runs alone over lists built to reach its rare branches (tests/final_problems.py), and the oracle's branch counters
prove they were reached."""
import numpy as np
import pytest

from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper
from oracle import oracle as O
from tests.final_problems import edge_reads, perturb, plant_edge_sites, tip_reads
from tests.mapper_check import compare

pytestmark = pytest.mark.gpu
L, K = 150, 12


def _recs(n):
    recs = np.zeros(n, O.READ_DTYPE)
    recs["bases_off"] = np.arange(n, dtype=np.int64) * L
    recs["len"] = L
    return recs


def _problem(paired, n_tip, n_edge, seed):
    ref = W.make_reference(200000, seed=15 + seed, pad=0, repeat_frac=0.1)
    if paired:
        reads, _ = W.make_pairs(ref, n_tip // 2, read_len=L, seed=seed, pad=300, hard_frac=0.1)
        reads = reads.reshape(-1, L)
        tips = tip_reads(ref, n_tip, L, seed, 300)
        reads[1::4] = tips[1::4]                              # every second pair gets a damaged second mate
    else:
        reads = tip_reads(ref, n_tip, L, seed, 300)
    er, info = edge_reads(ref, n_edge, L, seed + 1)
    reads = np.concatenate([reads, er])
    oi = O.OracleIndex([ref], k=K)
    if paired:
        oi.s.p.quitAfterTwoPerfects = 0
    offs = O.make_offsets(L, K, 1.9)
    ks = [100 * K] * len(offs)
    n = len(reads)
    recs = _recs(n)
    recs["nkeys"] = len(offs)
    keyinfo = np.concatenate([np.asarray(offs, np.int32), np.asarray(ks, np.int32)])
    pre = O.map_reads(oi, recs, reads.reshape(-1), keyinfo, None, paired, O.map_default_params(finalStage=0), cap=32, threads=8, match_stride=4200)
    s, ns = perturb(pre["sites"], pre["nsites"], seed + 2, len(ref))
    plant_edge_sites(s, ns, n_tip, info)
    s["match_job"] = -1                                       # (the lists' fills belong to another run's log)
    return ref, oi, reads, recs, offs, ks, s, ns


def _run(paired, cols, seed=3, n_tip=2400, n_edge=400):
    ref, oi, reads, recs, offs, ks, s, ns = _problem(paired, n_tip, n_edge, seed)
    n = len(reads)
    params = O.map_default_params(msaMaxColumns=cols, alignColumns=cols)
    O.final_branch_counts(oi)
    orc = O.final_reads(oi, recs, reads.reshape(-1), s, ns, paired=paired, params=params)
    branches = O.final_branch_counts(oi)
    di = DeviceIndex.build([ref], k=K)
    mp = Mapper(di, n, L, offs, ks, paired=paired, max_sites=32, msaMaxColumns=cols, alignColumns=cols)
    mp.load_reads(reads)
    mp.step()                                                 # (writes the reverse complements the stage reads)
    mp.final_only(s, ns)
    out, st = mp.fetch(), mp.stats()
    mp.close()
    di.close()
    bad = compare(out, orc, n, paired)
    assert not bad, "\n".join(bad[:20])
    kinds = np.bincount(orc["log"]["kind"], minlength=7)
    assert st["final_fills"] == len(orc["log"])
    return branches, kinds, orc, st


def test_final_stage_rare_paths_single_ended():
    branches, kinds, orc, st = _run(False, 3000)
    for name in ("clip_tip_indels", "fix_xy", "to_local", "to_local_clipped", "realign_recursion", "second_realign", "later_site_matched",
                 "duplicate_best_removed"):
        assert branches[name] > 0, (name, branches)
    assert kinds[3] > 1000 and kinds[4] > 100                 # first fills and padded refills
    f = orc["final"]
    clipped = sum(1 for i in range(len(f)) if f["match_len"][i] and orc["fmatch"][i][0] == ord("C"))
    assert clipped > 50 and st["final_local"] > 50            # 'C' strings out of toLocalAlignment
    assert st["final_rounds"] >= 5


def test_final_stage_third_fill_and_fill_unlimited_in_a_small_msa():
    branches, kinds, _, _ = _run(False, 250, seed=3)
    assert kinds[5] > 10, kinds                               # realign_new's third fill
    assert kinds[6] >= 1, kinds                               # and its fillUnlimited
    assert branches["resort_loop"] > 0


def test_final_stage_rare_paths_paired():
    branches, kinds, orc, _ = _run(True, 3000, seed=5)
    for name in ("clip_tip_indels", "fix_xy", "to_local_clipped", "realign_recursion", "second_realign"):
        assert branches[name] > 0, (name, branches)
    assert kinds[3] > 500 and kinds[4] > 50
    f = orc["final"]
    assert 0 < int(f["paired"].sum()) < len(f)                # some pairs stay paired, damaged ones do not


def test_final_stage_pool_and_logs_grow(monkeypatch):
    """A match-string pool and fill logs far too small for the batch: the stage repeats the steps that found no room after the
    host has grown them, and nothing changes in the results."""
    monkeypatch.setenv("BBMAP_FINAL_POOL_UNITS", "4096")
    ref, oi, reads, recs, offs, ks, s, ns = _problem(False, 1200, 100, 9)
    n = len(reads)
    orc = O.final_reads(oi, recs, reads.reshape(-1), s, ns)
    di = DeviceIndex.build([ref], k=K)
    mp = Mapper(di, n, L, offs, ks, paired=False, max_sites=32, jobsPerRead=-64)
    mp.load_reads(reads)
    mp.step()
    mp.final_only(s, ns)
    out, st = mp.fetch(), mp.stats()
    mp.close()
    di.close()
    bad = compare(out, orc, n, False)
    assert not bad, "\n".join(bad[:20])
    assert st["log_growths"] >= 1 and len(out["final_match"]) > 4 * 4096
