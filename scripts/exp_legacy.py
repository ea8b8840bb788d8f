"""debug: where the MAT planes differ from the oracle's packed"""
import random, sys
import numpy as np
sys.path.insert(0, ".")
from bbmap_amd import msa as M
from oracle.oracle import OracleMSA
from tests.test_msa_gpu import _legacy_cases
rng = random.Random(91)
maxRows, maxCols = 160, 300
band = (0, 0.0) if len(sys.argv) < 2 else (40, 0.18)
ctx = M.MSAContext(maxRows=maxRows, maxColumns=maxCols, bandwidth=band[0], bandwidthRatio=band[1], legacy=True)
ref = bytes(rng.choice(b"ACGT") for _ in range(3000))
MARK = 0x5a5a5a5a
shown = 0
for n, (rd, a, b, limited, ms) in enumerate(_legacy_cases(rng, ref, 48)):
    om = OracleMSA(maxRows, maxCols, bandwidth=band[0], bandwidthRatio=band[1])
    view = np.ctypeslib.as_array(om.s.packed, shape=(3, maxRows + 1, maxCols + 1))
    pristine = view.copy()
    view[:, 1:, 1:] = MARK
    exp = om.fill_limited_raw(rd, ref, a, b, ms) if limited else om.fill_unlimited_raw(rd, ref, a, b)
    packed = pristine.copy().reshape(-1)
    got = ctx.fill_packed(rd, ref, a, b, ms, limited, packed, limits=True)
    rows, cols = len(rd), b - a + 1
    ours = packed.reshape(3, maxRows + 1, maxCols + 1)
    o = view[:, 1:rows + 1, 1:cols + 1]; g = ours[:, 1:rows + 1, 1:cols + 1]
    wrote = o != MARK
    dif = wrote & (o != g)
    print("case", n, "limited", limited, "rows", rows, "cols", cols, "res", got[0], exp[0], "wrote", int(wrote.sum()), "diff", [int(dif[s].sum()) for s in range(3)])
    if dif.any() and shown < 3:
        shown += 1
        idx = np.argwhere(dif)
        for s, r, c in idx[:12]:
            print("   plane", s, "row", r + 1, "col", c + 1, "ours", g[s, r, c], (g[s, r, c] >> 11, g[s, r, c] & 2047), "oracle", o[s, r, c], (o[s, r, c] >> 11, o[s, r, c] & 2047))
print(ctx.legacy_stats())
