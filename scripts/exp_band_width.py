"""How wide a band of diagonals do the fills of the bench's read mix need?  CPU only: the oracle's mapper gives the fill log (window,
minScore per fillAndScoreLimited call), the oracle's fill on a marked matrix gives the cells the native code visits; a fill's band is
max(col - row) - min(col - row) + 1 over its visited cells in rows >= 2 (row 1 visits every column by construction)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bbmap_amd import workload as W
from oracle import oracle as O
L, k = 150, 13
ref = W.make_reference(3000000, seed=38, repeat_frac=0.1, families=200)
reads, _ = W.make_pairs(ref, 6000, read_len=L, seed=4, del_model="randomreads")
oi = O.OracleIndex([ref], k=k)
oi.s.p.quitAfterTwoPerfects = 0
from bbmap_amd import keys as K
offs, ks_, _ = K.make_keys(np.frombuffer(b"ACGT" * 38, np.uint8)[:L], None, K.default_config(K.PROFILE_BBMAP, k=k))
r = reads.reshape(-1, L)
out = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), L, offs, ks_, threads=8)
log = out["log"]
print("fills", len(log), "kinds", np.bincount(log["kind"]))
MARK = 0x5a5a5a5a
maxRows, maxCols = 160, 640
om = O.OracleMSA(maxRows, maxCols)
view = np.ctypeslib.as_array(om.s.packed, shape=(3, maxRows + 1, maxCols + 1))
refb = ref.tobytes()
widths, cols_all, slack = [], [], []
t0 = time.time()
rng = np.random.default_rng(1)
pick = rng.permutation(len(log))[:3000]
for i in pick:
    e = log[i]
    rd_index, a, b, ms = int(e["read"]), int(e["refStartLoc"]), int(e["refEndLoc"]), int(e["minScore"])
    a = max(0, a); b = min(len(ref) - 1, b)
    cols = b - a + 1
    if cols > maxCols or int(e["ngaps"]):
        continue
    strand = int(e["strand"])
    rd = r[rd_index]
    if strand:
        rd = W.revcomp_rows(rd[None, :])[0]
    view[:, 1:, 1:] = MARK
    # fillAndScoreLimited's gate: limited unless ...; use the raw limited fill with minScore - 120 as the Java wrapper does
    res, it = om.fill_limited_raw(rd.tobytes(), refb, a, b, max(ms - 120, 1))
    wrote = (view[:, 2:L, 1:cols + 1] != MARK).any(axis=0)          # (not the last row: the native code spreads BADoff over all of it first)
    rows_i, cols_i = np.nonzero(wrote)
    if len(rows_i) == 0:
        widths.append(0); cols_all.append(cols); continue
    d = (cols_i + 1) - (rows_i + 2)
    widths.append(int(d.max() - d.min() + 1)); cols_all.append(cols)
    slack.append(70 + 100 * (L - 1) - ms)
w = np.array(widths); c = np.array(cols_all)
print("sampled", len(w), "in %.0f s" % (time.time() - t0))
for lim in (8, 12, 16, 24, 32, 48, 64, 96, 128):
    print("band <= %3d: %5.1f %%   (windows <= 256 columns only: %5.1f %%)" % (lim, 100 * (w <= lim).mean(), 100 * (w[c <= 256] <= lim).mean()))
print("windows <= 256 columns: %.1f %%" % (100 * (c <= 256).mean()))
