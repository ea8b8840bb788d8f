"""GPU test of the device-resident pipeline (revcomp -> probe -> site filter -> DP) against an emulation
built from the oracle's pieces."""
import numpy as np
import pytest

from bbmap_amd import msa as M
from bbmap_amd.index import HostIndex
from bbmap_amd.pipeline import MapPipeline
from bbmap_amd import workload as W
from oracle.oracle import OracleIndex, OracleMSA, make_offsets, score_no_indels, score_no_indels_match, set_perfect
from tests.index_problems import revcomp

pytestmark = pytest.mark.gpu


def test_pipeline_matches_emulation():
    k, L, n = 12, 150, 3000
    ref = W.make_reference(300000, seed=5, pad=2000)
    reads, _, truth = W.make_reads_and_jobs(ref, n, read_len=L, seed=9, pad=2000, long_del_frac=0.4)
    hi = HostIndex([ref], k=k)
    offs = make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    pipe = MapPipeline(hi, n, L, offs, ks, max_sites=8, max_columns=256)
    pipe.load_reads(reads)
    njobs = pipe.step()
    out = pipe.fetch(njobs)
    cnt = pipe.last_counters
    assert njobs == cnt[0] and njobs > 0.05 * n          # the mutated mix sends a good share of reads to DP
    assert cnt[1] > 0.5 * n                               # and finishes most reads without DP

    oi = OracleIndex([ref], k=k)
    om = OracleMSA(160, 256)
    refb = ref.tobytes()
    maxSw = 70 + (L - 1) * 100
    maxImp = maxSw - 495
    minMsaLimit = -258 + int(np.float32(0.56) * np.float32(maxSw))
    by_src = {int(s): i for i, s in enumerate(out["src"])}
    assert len(by_src) == njobs
    by_gsrc = {int(s): i for i, s in enumerate(out["gsrc"])}
    assert len(by_gsrc) == cnt[2]
    omg = OracleMSA(160, 3000)
    checked_jobs = checked_gapped = checked_ungapped = 0
    for r in range(600):
        bp = reads[r * L:(r + 1) * L].tobytes()
        bm = revcomp(bp)
        exp_sites = oi.find(bp, bm, [0] * L, ks, offs, cap=8)
        ns = int(out["nsites"][r])
        assert ns == len(exp_sites)
        near, force, sws = 0, False, []
        for s, e in enumerate(exp_sites):
            g = out["sites"][r, s]
            assert (int(g["chrom"]), int(g["strand"]), int(g["hits"])) == (e["chrom"], e["strand"], e["hits"])
            bases = bm if e["strand"] else bp
            # AbstractMapThread.scoreNoIndels (current/align2/AbstractMapThread.java:762-856), restated on the oracle's pieces
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
            assert int(out["no_indel"][r, s]) == sw
            assert (int(g["start"]), int(g["stop"]), int(g["perfect"]), int(g["semiperfect"])) == \
                   (e["start"], e["stop"], int(e["perfect"]), int(e["semiperfect"])), (r, s, g, e)
        num_near = -near if force else near
        # reads finished without DP: state = (best site << 2) | 1 and the ungapped match string of that site
        state = int(out["read_state"][r])
        if not exp_sites:
            assert state == -1
        elif num_near >= 1:
            best = max(range(len(sws)), key=lambda q: (sws[q], -q))
            assert state == (best << 2) | 1, (r, state, sws)
            e = exp_sites[best]
            sc, ms = score_no_indels_match(bm if e["strand"] else bp, refb, e["start"])
            if sc == -99999:
                assert int(out["ungapped_len"][r]) == -1
            else:
                assert int(out["ungapped_len"][r]) == L and out["ungapped_match"][r].tobytes() == ms
            checked_ungapped += 1
        else:
            assert state == 2 and int(out["ungapped_len"][r]) == 0
        for s, e in enumerate(exp_sites):
            src = r * 8 + s
            semip = bool(e["semiperfect"])
            gapped_now = bool(e["gaps"])
            needs_any = num_near < 1 and sws[s] < maxImp and not semip
            needs = needs_any and not gapped_now
            assert (src in by_src) == needs, (r, s, e, sws[s], num_near)
            assert (src in by_gsrc) == (needs_any and gapped_now), (r, s, e, sws[s], num_near)
            if needs_any and gapped_now:
                i = by_gsrc[src]
                j = out["gjobs"][i]
                gg = out["ggaps"][i]
                assert gg["gaps"][: gg["ngaps"]].tolist() == e["gaps"]
                assert (int(j["refStartLoc"]), int(j["refEndLoc"])) == (e["start"] - 4, e["stop"] + 4)
                bases = bm if e["strand"] else bp
                sv, mx = omg.fillAndScoreLimited(bases, refb, int(j["refStartLoc"]), int(j["refEndLoc"]), int(j["minScore"]), e["gaps"])
                res = out["gresults"][i]
                assert int(res["status"]) != 2
                gs = None if res["score_len"] == 0 else res["score"][: res["score_len"]].tolist()
                assert gs == sv
                if sv is not None:
                    tb = omg.traceback(bases, refb, max(0, int(j["refStartLoc"])), min(len(refb) - 1, int(j["refEndLoc"])),
                                       mx[0], mx[1], mx[2], gapped=True)
                    assert out["gmatch"][i, : res["match_len"]].tobytes() == tb
                checked_gapped += 1
            if needs:
                i = by_src[src]
                j = out["jobs"][i]
                assert (int(j["refStartLoc"]), int(j["refEndLoc"])) == (e["start"] - 4, e["stop"] + 4)
                assert int(j["minScore"]) == max(sws[s], minMsaLimit)
                bases = bm if e["strand"] else bp
                sv, mx = om.fillAndScoreLimited(bases, refb, int(j["refStartLoc"]), int(j["refEndLoc"]), int(j["minScore"]))
                res = out["results"][i]
                gs = None if res["score_len"] == 0 else res["score"][: res["score_len"]].tolist()
                assert gs == sv
                if sv is not None:
                    tb = om.traceback(bases, refb, max(0, int(j["refStartLoc"])), int(j["refEndLoc"]), mx[0], mx[1], mx[2])
                    assert out["match"][i, : res["match_len"]].tobytes() == tb
                checked_jobs += 1
    assert checked_jobs > 20 and checked_ungapped > 200
    assert cnt[2] > 0 and checked_gapped > 0


def test_match_no_indels_kernel_agrees_with_the_fused_path():
    """bbpipe_match_no_indels_device (stand-alone) writes what the site filter writes on the way."""
    import ctypes as C
    import torch
    from bbmap_amd import _lib
    k, L, n = 12, 150, 2000
    ref = W.make_reference(200000, seed=6, pad=2000)
    reads, _, _ = W.make_reads_and_jobs(ref, n, read_len=L, seed=10, pad=2000)
    hi = HostIndex([ref], k=k)
    offs = make_offsets(L, k, 1.9)
    pipe = MapPipeline(hi, n, L, offs, [100 * k] * len(offs), max_sites=8, max_columns=256)
    pipe.load_reads(reads)
    nj = pipe.step()
    fused = pipe.fetch(nj)
    m2 = torch.zeros_like(pipe.ungapped_match)
    l2 = torch.full_like(pipe.ungapped_len, 77)
    lib = _lib.load()
    _lib.check(lib.bbpipe_match_no_indels_device(None, n, pipe.reads.data_ptr(), pipe.bases.data_ptr(), pipe.total_bytes,
                                                 pipe.sites.data_ptr(), pipe.max_sites, pipe.read_state.data_ptr(),
                                                 pipe.chrom_off.data_ptr(), pipe.chrom_len.data_ptr(), pipe.refs.data_ptr(),
                                                 m2.data_ptr(), L, l2.data_ptr()), "bbpipe_match_no_indels_device")
    torch.cuda.synchronize()
    l2 = l2.cpu().numpy(); m2 = m2.cpu().numpy().reshape(n, L)
    assert (l2 == fused["ungapped_len"]).all() and (l2 == L).sum() > 500
    rows = np.nonzero(l2 == L)[0]
    assert (m2[rows] == fused["ungapped_match"][rows]).all()
