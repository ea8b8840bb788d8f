"""GPU tests of the boundary's edge behaviour: empty batches, bad arguments, shapes outside the limits."""
import ctypes as C

import numpy as np
import pytest

from bbmap_amd import _lib
from bbmap_amd import msa as M
from bbmap_amd.index import DeviceIndex
from bbmap_amd.rescue import quick_rescue_batch

pytestmark = pytest.mark.gpu


def test_empty_batches_are_fine():
    al = M.MultiStateAligner11ts(maxRows=64, maxColumns=128)
    assert al.align([]) == []
    assert al.alignGapped([]) == []
    assert quick_rescue_batch([], [b"ACGT" * 100]) == []
    di = DeviceIndex.build([b"N" * 50 + b"ACGTTGCA" * 200 + b"N" * 50], k=10)
    assert di.find_batch([]) == []
    di.close()


def test_bad_shapes_are_reported_not_computed():
    al = M.MultiStateAligner11ts(maxRows=64, maxColumns=128)
    ref = bytes(np.random.default_rng(1).choice(list(b"ACGT"), 1000).astype(np.uint8))
    too_long_read = (ref[100:180], ref, 96, 190, 1000)                 # 80 rows > maxRows
    too_wide = (ref[100:150], ref, 96, 400, 1000)                      # 305 columns > maxColumns, no clamp flag
    with pytest.raises(ValueError):
        al.align([too_long_read], M.FILL_LIMITED | M.DO_SCORE)
    with pytest.raises(ValueError):
        al.align([too_wide], M.FILL_LIMITED | M.DO_SCORE)
    # gapped jobs: odd gap arrays, gaps outside the reference, gaps shorter than the reference's minimum
    read = ref[100:150]
    bad = [(read, ref, 96, 760, 500, [100, 120, 700]),                 # odd count
           (read, ref, 96, 760, 500, [100, 120, 700, 2000]),           # beyond the reference array
           (read, ref, 96, 400, 500, [100, 120, 150, 400])]            # a 29-base "gap"
    got = al.alignGapped(bad)
    assert [g["status"] for g in got] == [M.ST_BAD_SHAPE] * 3
    assert all(g["score"] is None for g in got)


def test_index_build_rejects_bad_geometry_and_handles_tiny_chromosomes():
    with pytest.raises(_lib.BBMapAmdError):
        DeviceIndex.build([b"ACGT" * 100], k=7)
    with pytest.raises(_lib.BBMapAmdError):
        DeviceIndex.build([b"ACGT" * 100], k=16)
    # a chromosome shorter than k contributes nothing; an all-N chromosome neither
    di = DeviceIndex.build([b"ACGTAC", b"N" * 500, b"N" * 20 + b"ACGTTGCATGCATTGACCAGT" * 40 + b"N" * 20], k=11, chromBits=2)
    starts, sites, counts, hist = di.export_block(0)
    assert starts[-1] == len(sites) and len(sites) > 0
    assert ((sites >> (31 - 2)) == 3).all()                             # every entry belongs to chromosome 3
    # a read without a single defined k-mer has no site; a key offset that does not fit its read is an argument error
    assert di.find_batch([(b"N" * 60, [0] * 60, [1100, 1100], [0, 49])]) == [[]]
    with pytest.raises(_lib.BBMapAmdError):
        di.find_batch([(b"ACGTAC", [0] * 6, [1100], [0])])
    # the read-length announcement is a sizing hint: nonsense is refused, and a read longer than announced is still answered
    with pytest.raises(_lib.BBMapAmdError):
        di.set_max_read_len(0)
    rd = (b"ACGTTGCATGCATTGACCAGT" * 40)[5:205]
    offs = list(range(0, 190, 9))
    di.set_max_read_len(600)
    want = di.find_batch([(rd, [0] * len(rd), [1100] * len(offs), offs)])
    di.set_max_read_len(100)
    assert di.find_batch([(rd, [0] * len(rd), [1100] * len(offs), offs)]) == want
    di.close()


def test_rescue_edges():
    ref = bytes(np.random.default_rng(2).choice(list(b"ACGT"), 3000).astype(np.uint8))
    probs = [(ref[1000:1009], 1, 900, 300, True, 1000, 2),              # shorter than 10 bases: never rescued
             (ref[1000:1700], 1, 900, 300, True, 1000, 2),              # longer than the kernel's 600: reported, not computed
             (ref[1000:1100], 1, 2950, 300, True, 1000, 2),             # search range empty after clipping to the chromosome
             (ref[1000:1100], 1, 990, 0, True, 1000, 2)]                # searchDist 0, start not at loc
    got = quick_rescue_batch(probs, [ref])
    assert got == [None, None, None, None]
    assert quick_rescue_batch([(ref[1000:1100], 1, 1000, 0, True, 1000, 2)], [ref])[0]["start"] == 1000


def test_revcomp_kernel():
    import torch
    from bbmap_amd.index import READ_DTYPE
    L = _lib.load()
    rng = np.random.default_rng(5)
    lens = [150, 1, 77, 600, 64, 65]
    reads = np.concatenate([rng.choice(list(b"ACGTNacgtnRYKM-"), n).astype(np.uint8) for n in lens])
    recs = np.zeros(len(lens), READ_DTYPE)
    recs["bases_off"] = np.cumsum([0] + lens[:-1]); recs["len"] = lens
    dev = torch.device("cuda", 0)
    t_in = torch.from_numpy(reads).to(dev); t_out = torch.zeros_like(t_in)
    t_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
    _lib.check(L.bbpipe_revcomp_device(None, len(lens), t_recs.data_ptr(), t_in.data_ptr(), t_out.data_ptr()), "bbpipe_revcomp_device")
    comp = {ord(a): ord(b) for a, b in zip("ACGTNacgtnRYKM-", "TGCANtgcanYRMK-")}
    out = t_out.cpu().numpy()
    for off, n in zip(recs["bases_off"], lens):
        assert out[off:off + n].tolist() == [comp[b] for b in reads[off:off + n][::-1]]
