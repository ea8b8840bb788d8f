"""BASELINE.json configs[4] end to end on the device, against the oracle compiled with mapPacBio's classes (-DORC_PACBIO):
BBIndexPacBio probe (long-read kernel) -> trimList -> scoreNoIndels / tip search -> scoreSlow with MultiStateAligner9PacBio fills
(strip-tiled wavefront kernel) -- site lists field by field, every fill's window / minScore / score vector / visited-cell count /
traceback string.  Keys are placed by bbkeys_make_batch as quickMap places them (density floor 2.8: ~230 keys per kilobase)."""
import numpy as np
import pytest

from bbmap_amd import keys as K
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex, PROFILE_PACBIO
from bbmap_amd.mapper import Mapper
from oracle import oracle as O
from tests.mapper_check import compare

pytestmark = pytest.mark.gpu


def _run(chroms, pieces, msa_rows, msa_cols, threads=4, max_sites=32):
    cfg = K.default_config(K.PROFILE_PACBIO)
    recs, blob, bs, keyinfo = K.make_batch(pieces, None, cfg)
    di = DeviceIndex.build(chroms, profile=PROFILE_PACBIO)
    mp = Mapper.from_records(di, recs, blob, bs, keyinfo, paired=False, max_sites=max_sites, profile=PROFILE_PACBIO, msaMaxColumns=msa_cols)
    mp.step()
    out, st = mp.fetch(), mp.stats()
    oi = O.OracleIndex(chroms, profile="pacbio")
    params = O.map_default_params("pacbio", msaMaxRows=msa_rows, msaMaxColumns=msa_cols)
    orc = O.map_reads(oi, recs, blob, keyinfo, base_scores=bs, paired=False, params=params, cap=64, threads=threads)
    mp.close()
    di.close()
    return out, orc, st


def test_pacbio_pieces_match_the_oracle():
    chroms = [W.make_reference(300000, seed=81, pad=3000, repeat_frac=0.1, families=60), W.make_reference(200000, seed=82, pad=3000)]
    pieces, truth = W.make_pacbio_pieces(chroms, 48, seed=3, min_len=300, max_len=2600, pad=3000, junk_frac=0.1)
    # a few nearly clean pieces (perfect / semiperfect handling) and one shorter than k
    clean, _ = W.make_pacbio_pieces(chroms, 6, seed=4, min_len=400, max_len=1500, err=(0.0, 0.01), pad=3000)
    pieces = pieces + clean + [pieces[0][:9].copy()]
    out, orc, st = _run(chroms, pieces, 2700, 3400)
    n = len(pieces)
    bad = compare(out, orc, n, paired=False)
    assert not bad, "\n".join(bad[:20])
    assert st["reads_overflowed"] == 0 and st["fills"] + st["gapped_fills"] >= 30
    # the pieces come back at their origin
    hit = 0
    for i in range(48):
        ns = int(out["nsites"][i])
        if ns > 0:
            s = out["sites"][i][0]
            hit += int(s["chrom"] == truth[i][0] and s["strand"] == truth[i][1] and abs(int(s["start"]) - int(truth[i][2])) < 300)
    assert hit >= 36


def test_full_length_pieces():
    """fastareadlen = 6000: pieces of exactly 6000 bases (1400 keys each) through the probe and the strip kernel."""
    chroms = [W.make_reference(500000, seed=83, pad=8000, repeat_frac=0.05, families=40)]
    pieces, truth = W.make_pacbio_pieces(chroms, 6, seed=8, pad=8000)
    out, orc, st = _run(chroms, pieces, 6020, 7600, threads=2)
    bad = compare(out, orc, len(pieces), paired=False)
    assert not bad, "\n".join(bad[:20])
    assert all(int(x) > 0 for x in out["nsites"])
