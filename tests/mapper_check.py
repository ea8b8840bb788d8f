"""Shared by the mapper tests, __graft_entry__.smoke() and bench.py's parity sample: compares what the device mapper (bbmap_*)
returned for a batch with what the CPU oracle's mapper restatement (oracle/mapper_oracle.c) returns for the same reads."""
import numpy as np

SITE_FIELDS = ("chrom", "strand", "start", "stop", "hits", "quickScore", "score", "slowScore", "pairedScore",
               "perfect", "semiperfect", "rescued", "ngaps")
FINAL_FIELDS = ("mapped", "chrom", "strand", "start", "stop", "mapScore", "paired", "ambiguous", "perfect", "rescued", "match_len", "nsites")
GAPPED_BIT = 1 << 30
TRACE_KEEP_GAPS = 1 << 7


def gpu_fills(out):
    """{(read, seq): dict} over both fill logs of a Mapper.fetch()."""
    fills = {}
    for jobs, res, info, match, bit, stride in ((out["jobs"], out["results"], out["jobinfo"], out.get("match"), 0, out["match_stride"]),
                                                (out["gjobs"], out["gresults"], out["gjobinfo"], out.get("gmatch"), GAPPED_BIT, out["gmatch_stride"])):
        for i in range(len(jobs)):
            if int(info["seq"][i]) < 0:                 # issued ahead of time, not part of the reference's sequence
                continue
            key = (int(info["read"][i]), int(info["seq"][i]))
            assert key not in fills, "two fills with the same (read, seq) %r" % (key,)
            n = int(res["score_len"][i])
            ml = int(res["match_len"][i])
            mstr = match[i, :ml].tobytes() if (match is not None and n > 0 and ml > 0) else b""
            if int(jobs["flags"][i]) & TRACE_KEEP_GAPS and ml > 0:      # the final stage's fills keep gap symbols compact in the log
                mstr = mstr.replace(b"-", b"D" * 128)
                ml = len(mstr) if match is not None else ml
            fills[key] = dict(kind=int(info["kind"][i]), refStartLoc=int(jobs["refStartLoc"][i]), refEndLoc=int(jobs["refEndLoc"][i]),
                              minScore=int(jobs["minScore"][i]), score=res["score"][i][:n].tolist(), iterations=int(res["iterations"][i]),
                              match=mstr, index=i | bit, match_len=ml if n > 0 else 0, stride=max(int(stride), ml))
    return fills


def oracle_fills(orc):
    fills = {}
    log, match = orc["log"], orc["match"]
    for i in range(len(log)):
        key = (int(log["read"][i]), int(log["seq"][i]))
        n = int(log["score_len"][i])
        ml = int(log["match_len"][i])
        fills[key] = dict(kind=int(log["kind"][i]), refStartLoc=int(log["refStartLoc"][i]), refEndLoc=int(log["refEndLoc"][i]),
                          minScore=int(log["minScore"][i]), score=log["score"][i][:n].tolist(), iterations=int(log["iterations"][i]),
                          match=(match[i, :ml].tobytes() if (n > 0 and ml <= match.shape[1]) else b""), index=i,
                          match_len=ml if n > 0 else 0, stride=int(match.shape[1]))
    return fills


def match_differs(a, b):
    """Traceback strings of a device fill and the oracle's.  A string longer than the slot its log gives it is reported by
    length only (device: match_len = -1, "did not fit"; the oracle keeps the length)."""
    if a["match_len"] < 0:           # the oracle's 0 on a successful fill: longer than its own traceback buffer (12 k symbols)
        return not (b["match_len"] > a["stride"] or b["match_len"] == 0)
    if a["match_len"] != b["match_len"]:
        return True
    if b["match_len"] > b["stride"]:
        return False                     # the oracle's copy was not kept: lengths agree, contents unchecked
    return a["match"] != b["match"]


def compare(out, orc, n_reads, paired, reads_range=None, check_match=True):
    """Returns a list of human-readable differences (empty = identical).  Read r of the device batch is oracle read r
    (single-ended) or mate r % 2 of pair r // 2 (paired).  Reads the overflow tier mapped (nsites == -3) are looked up in
    out["overflow"]: their site lists and their fills come from the tier's logs, whatever the main pass logged for them."""
    bad = []
    tier = out.get("overflow")
    in_tier = {}
    if tier is not None:
        in_tier = {int(r): i for i, r in enumerate(tier["read_ids"]) if int(out["nsites"][int(r)]) == -3}
    gf = {k: v for k, v in gpu_fills(out).items() if k[0] not in in_tier}
    g_by_index = {v["index"]: k for k, v in gf.items()}
    t_by_index = {}
    if tier is not None:
        ids = tier["read_ids"]
        for (i, seq), v in gpu_fills(tier).items():
            if int(ids[i]) in in_tier:
                gf[(int(ids[i]), seq)] = v
                t_by_index[v["index"]] = (int(ids[i]), seq)
    of = oracle_fills(orc)
    rng = range(n_reads) if reads_range is None else reads_range
    in_range = set(rng)
    gkeys = {k for k in gf if k[0] in in_range}
    okeys = {k for k in of if k[0] in in_range}
    if gkeys != okeys:
        bad.append("fill sets differ: only on device %s, only in oracle %s" % (sorted(gkeys - okeys)[:5], sorted(okeys - gkeys)[:5]))
    for k in sorted(gkeys & okeys):
        a, b = gf[k], of[k]
        for f in ("kind", "refStartLoc", "refEndLoc", "minScore", "score", "iterations"):
            if a[f] != b[f]:
                bad.append("fill %r field %s: device %r, oracle %r" % (k, f, a[f], b[f]))
                break
        else:
            if check_match and match_differs(a, b):
                bad.append("fill %r traceback string: device (%d) %r, oracle (%d) %r" % (k, a["match_len"], a["match"][:200], b["match_len"], b["match"][:200]))
    o_by_index = {v["index"]: k for k, v in of.items()}
    for r in rng:
        if "sites" in orc:                                  # oracle.map_reads: one list per read record, pairs interleaved
            osites, on, oi = orc["sites"], orc["nsites"], r
        elif paired:
            osites, on = (orc["sites1"], orc["nsites1"]) if r % 2 == 0 else (orc["sites2"], orc["nsites2"])
            oi = r // 2
        else:
            osites, on, oi = orc["sites1"], orc["nsites1"], r
        if r in in_tier:
            gsites, gn, by_index = tier["sites"][in_tier[r]], int(tier["nsites"][in_tier[r]]), t_by_index
        else:
            gsites, gn, by_index = out["sites"][r], int(out["nsites"][r]), g_by_index
        if gn != int(on[oi]):
            bad.append("read %d: %d sites on the device, %d in the oracle" % (r, gn, int(on[oi])))
            continue
        for s in range(max(gn, 0)):
            g, o = gsites[s], osites[oi, s]
            dif = [f for f in SITE_FIELDS if int(g[f]) != int(o[f])]
            if not dif and int(g["ngaps"]) and g["gaps"][: int(g["ngaps"])].tolist() != o["gaps"][: int(o["ngaps"])].tolist():
                dif = ["gaps"]
            gj, oj = int(g["match_job"]), int(o["match_job"])
            if not dif and ((gj < 0) != (oj < 0) or (gj >= 0 and by_index.get(gj) != o_by_index.get(oj))):
                dif = ["match_job"]
            if dif:
                bad.append("read %d site %d differs in %s: device %s, oracle %s" % (
                    r, s, dif, {f: int(g[f]) for f in SITE_FIELDS}, {f: int(o[f]) for f in SITE_FIELDS}))
                break
    bad += compare_final(out, orc, rng, paired)
    return bad


def oracle_final(orc, r, paired):
    """(record, match bytes) of read r in an oracle result (map_reads: interleaved; map_batch: per mate)"""
    if "final" in orc:
        f, fm = orc["final"][r], orc["fmatch"][r]
    elif paired:
        w = 1 + (r % 2)
        f, fm = orc["final%d" % w][r // 2], orc["fmatch%d" % w][r // 2]
    else:
        f, fm = orc["final1"][r], orc["fmatch1"][r]
    ml = int(f["match_len"])
    return f, (fm[:ml].tobytes() if ml <= fm.shape[0] else None)


def compare_final(out, orc, rng, paired):
    """The final alignment stage: what BBMap prints per read (mapped / chrom / strand / start / stop / mapScore / flags) and the match
    string, device (bbmap_get_final, tier included) against the oracle.  Skipped when either side ran without the stage."""
    if "final" not in out or not ("final" in orc or "final1" in orc):
        return []
    bad = []
    gfin, blob = out["final"], out["final_match"]
    for r in rng:
        o, om = oracle_final(orc, r, paired)
        g = gfin[r]
        if int(g["nsites"]) < 0 and int(g["nsites"]) != -3:
            continue                                        # flagged (overflow): reported with the site lists
        dif = [f for f in FINAL_FIELDS if int(g[f]) != int(o[f])]
        if dif:
            bad.append("read %d final record differs in %s: device %s, oracle %s" % (r, dif, {f: int(g[f]) for f in FINAL_FIELDS}, {f: int(o[f]) for f in FINAL_FIELDS}))
            continue
        ml = int(g["match_len"])
        if ml and om is not None:
            gm = blob[int(g["match_off"]): int(g["match_off"]) + ml].tobytes()
            if gm != om:
                bad.append("read %d match string: device %r, oracle %r" % (r, gm[:300], om[:300]))
    return bad
