// k-mer index probe for LONG reads with MANY keys -- align2.BBIndexPacBio.findAdvanced (mapPacBio), gfx950.
//
// mapPacBio cuts reads into pieces of up to 6000 bases and places keys at a density of at least 2.8 (BBMapPacBio.setDefaults,
// current/align2/BBMapPacBio.java:55-58; the density window of AbstractMapThread.java:663-665 floors at minKeyDensity whatever
// maxDesiredKeys says), so a piece carries up to 6000 * 2.8 / 12 = 1400 keys (BBIndexPacBio sizes its heaps for 2047,
// current/align2/BBIndexPacBio.java:2394-2396).  index_probe_wave.hip gives every key list a lane (<= 64 keys, 400 bases); here:
//   * ONE READ PER WAVEFRONT, one wavefront per workgroup; list l = 64 j + lane belongs to lane `lane` (up to 32 lists per lane);
//   * per wave in LDS: the read (both strands), its base scores, extendScore's per-base location array (6016 ints), every list's
//     current value, a two-entry look-ahead per list; the cold per-list data (cursor, end, offset, key score) sits in a per-wave
//     workspace in HBM, laid out [j][lane] so that every access is one coalesced line per j;
//   * the reference's heap (QuadHeap ordered by (site, column), Quad.java:19-22) only ever exposes its minimum: every lane scans
//     its own lists' heads, a DPP reduction gives the site, a second one the lowest column among the lanes that hold it;
//   * a list is consumed one entry at a time; ALL lists' look-ahead buffers are refilled together when a popped list finds its
//     own empty (64 x 2 gathers in flight per j instead of one exposed round trip per pop);
//   * quickScore's chain walk, scoreZ2's coverage and extendScore's per-key extension run over the columns 64 at a time, in
//     the reference's column order, with ballots and readlanes; calcAffineScore / setPerfect over the bases 64 at a time.
// The kernel is a template over the class family (bbidx_params.profile): ProfPacBio = BBIndexPacBio's constants and
// MultiStateAligner9PacBio.calcAffineScore (current/align2/MultiStateAligner9PacBio.java:1681-1870), ProfBBMap = BBIndex's with
// MultiStateAligner11tsJNI's -- the second instantiation exists so that this kernel is cross-checked against the two older
// kernels and the oracle on the same problems (tests/test_index_gpu.py), and takes BBMap reads with more than 128 keys.
//
// Functions follow current/align2/BBIndexPacBio.java (= BBIndex.java minus camelWalk3): find :394-615, trimExcessHitListsByGreedy
// :267-349 (+ Solver.java:46-151), prescanAllBlocks :618-711, findMaxQscore2 (BBIndex.java:2294-2450), slowWalk3 :1200-1680,
// quickScore / scoreLeft / scoreRight / scoreZ2 / maxQuickScore, extendScore, makeGapArray, calcApproxHitsCutoff :2562-2585.
#include <hip/hip_runtime.h>

#include <mutex>

#include <climits>
#include <cstdio>

#include "bbmap_amd.h"
#include "index_common.h"
#include "wave_prims.h"

void bbmap_set_error(const char *msg);

namespace bbidxl {
using namespace bbidx;
using namespace wavep;

constexpr int NB = 2;                       // look-ahead entries per list (1 would make room for a fifth wavefront per CU: measured, 5.5 -> 7.1 s per 8,192 pieces)
constexpr int KMAX = 2048;                  // lists per read the LDS arrays are sized for (BBIDX_PACBIO_MAX_KEYS + 1)
constexpr int LMAX = BBIDX_PACBIO_MAX_READ_LEN;
constexpr int WS_ARRAYS = 12;               // per-wave workspace: WS_ARRAYS x KMAX ints
constexpr int DEADV = -(1 << 30);           // value of an unused column: outside every window
typedef const int __attribute__((address_space(1))) *GlobalIntsT;   // global_load instead of flat_load

struct ProfBBMap {      // BBIndex.java:3168-3305 ; MultiStateAligner11tsJNI.java:871-1027, jni/MultiStateAligner11tsJNI.c:18-98
    static constexpr int Z_MULT = 20, SMALL_LIST = 20, MIN_LISTS_RETAIN = 6, INDEL_MULT = 20, PERFECT_RED = 0;
    static constexpr float HIT_FRACTION = 0.85f, MIN_SCORE_MULT = 0.15f, MIN_QSCORE_MULT = 0.025f, MIN_QSCORE_MULT2 = 0.1f, DYN_SCORE = 0.84f;
    static constexpr int RELAX1 = 4, RELAX2 = 3, RELAX3 = 3, RELAX4 = 2;
    __device__ static inline int indelPenalty(int bkhs) { return bkhs / 2 - 1; }
    static constexpr int MATCH = 70, MATCH2 = 100, SUB = -127, SUB2 = -51, SUB3 = -25;
    static constexpr int INS = -395, INS2 = -39, DEL = -472, DEL2 = -33, DEL3 = -9, DEL4 = -1, DEL5 = -1, GAP = -2;
    static constexpr int INS_DIF_PLUS = 0;              // POINTS_INS_ARRAY_C[min(loc - lastLoc, 5)]
};
struct ProfPacBio {     // BBIndexPacBio.java:2461-2596 ; MultiStateAligner9PacBio.java:2375-2407, :1681-1870
    static constexpr int Z_MULT = 25, SMALL_LIST = 80, MIN_LISTS_RETAIN = 12, INDEL_MULT = 25, PERFECT_RED = 2;
    static constexpr float HIT_FRACTION = 0.97f, MIN_SCORE_MULT = 0.02f, MIN_QSCORE_MULT = 0.005f, MIN_QSCORE_MULT2 = 0.005f, DYN_SCORE = 0.64f;
    static constexpr int RELAX1 = 20, RELAX2 = 18, RELAX3 = 16, RELAX4 = 14;
    __device__ static inline int indelPenalty(int bkhs) { return bkhs / 8 - 1; }
    static constexpr int MATCH = 90, MATCH2 = 100, SUB = -137, SUB2 = -49, SUB3 = -25;
    static constexpr int INS = -205, INS2 = -42, DEL = -292, DEL2 = -37, DEL3 = -17, DEL4 = -2, DEL5 = -1, GAP = -2;
    static constexpr int INS_DIF_PLUS = 1;              // dif = min(loc - lastLoc + 1, 5), :1729
};

// debug build (-DBBIDXL_TIMERS): the five work counters become cycle counts (in units of 1024 cycles) of
// {refillAll, popSite without its refills, quick scores, minHead + countWindow, everything else of the walk loops}
#ifdef BBIDXL_TIMERS
#define LT_BEGIN(u) const unsigned long long lt0_ = __builtin_readcyclecounter()
#define LT_END(u, slot) (u).tm[slot] += __builtin_readcyclecounter() - lt0_
#else
#define LT_BEGIN(u) do { } while (0)
#define LT_END(u, slot) do { } while (0)
#endif

struct LongParams {
    Params P;
    int *ws;                 // per-wave workspace, WS_ARRAYS * KMAX ints each
    int maxKeys, maxLen;     // what the launch's LDS / workspace hold (<= KMAX, <= LMAX)
};

constexpr int JM = KMAX / 64;               // lists per lane
typedef int vJM __attribute__((ext_vector_type(32)));   // JM ints in registers; a uniform variable index is one indexed register move
static_assert(JM == 32, "vJM holds JM values");

// LDS of one wave (77 KB: two waves per CU)
// A wave's LDS, laid out at launch for the longest read and the most keys of THE BATCH (Q.maxLen / Q.maxKeys, found by a pre-pass):
// the block size in LDS decides how many wavefronts a CU holds (one per block), and this kernel is latency-bound -- 6,000-base
// pieces with 1,400 keys take 48 KB instead of the 77 KB of the largest shape (6,016 / 2,047), three blocks per CU instead of two.
// The read's bases (both strands) and base scores are NOT mirrored in LDS: they are read from global memory where they lie
// (plus strand and scores: the caller's buffers; minus strand: the rc output, or the wave's workspace), in coalesced chunks by
// the few functions that need them (extendScore, setPerfect, the key extraction, calcAffineScore).
struct Lds {
    unsigned short *loc;     // [LM] location array, as 16-bit offsets from U::locBase (ld_loc / st_loc); 0xFFFF = -1, 0xFFFE = -2
    int *val;                // [KM] every list's current value (kept after it ran out): what the scoring functions index by column
    int *nb[NB];             // [KM] look-ahead entries, already adjusted by the key's offset
    uint8_t *st;             // [KM] bits 0-1 look-ahead entries consumed, bits 2-3 valid look-ahead entries, bit 7 they reach the list's end
    short *ksc;              // [KM] the list's key score (<= 100 k) and offset (< 6016)
    short *off;
    const uint8_t *base[2];  // global memory: the read, plus and minus strand
    const int8_t *bsc;       // global memory: base scores
    int *gaps;               // [BBIDX_MAX_GAPS]
    int *ngapsP;
};
__host__ __device__ inline int lds_km(int maxKeys) { int k = (maxKeys + 1 + 63) & ~63; return k > KMAX ? KMAX : k; }
__host__ __device__ inline int lds_lm(int maxLen) { int l = (maxLen + 15) & ~15; return l > LMAX ? LMAX : l; }
__host__ __device__ inline int lds_bytes(int maxKeys, int maxLen) {
    const int KM = lds_km(maxKeys), LM = lds_lm(maxLen);
    return 2 * LM + 4 * KM * (1 + NB) + 2 * KM * 2 + KM + 4 * (BBIDX_MAX_GAPS + 4);
}

struct U {
    const DevIndex *ix;
    Codec c;
    int k, baseKeyHitScore, indelPenalty, maxPenalty, scoreZ1Key;
    int lane, blen;
    int locBase;             // what the 16-bit entries of the location array are offsets from (set by extendScoreL per site)
    unsigned cPrescan, cWalk, cExtend, cRefBytes;
#ifdef BBIDXL_TIMERS
    mutable unsigned long long tm[5];
#endif
};

// The location array holds reference positions of key diagonals, all within [center - MAX_INDEL, center + MAX_INDEL2] of the site being
// extended (and, in makeGapArray, those plus a base index): 16 bits as offsets from that lower bound halve its LDS, which decides how
// many wavefronts a CU holds.
__device__ __forceinline__ int ld_loc(const Lds &S, const U &u, int i) { const int r = S.loc[i]; return r >= 0xFFFE ? r - 0x10000 : r + u.locBase; }
__device__ __forceinline__ void st_loc(const Lds &S, const U &u, int i, int v) { S.loc[i] = (unsigned short)(v < 0 ? v + 0x10000 : v - u.locBase); }

// The lists of one (block, strand) cycle.  The heap stand-in works on registers: lane `lane` keeps the value, cursor and end of its
// lists 64 j + lane in v[j] (every loop over j is fully unrolled, so the array never leaves the register file).
struct Lists {
    int n, nlive;            // lists, lists still in the heap
    int J;                   // ceil(n / 64)
    unsigned live;           // bit j: list 64 j + lane is alive (per lane)
    int dmax;                // per lane: the largest value among its lists that ran out (INT_MIN: none)
    vJM v;
    int *rowW, *stopW;       // every list's cursor and end, in the per-wave workspace (read at refills only)
    GlobalIntsT sites;
};

template <class PF> __device__ __forceinline__ int calcApproxHitsCutoffP(const bbidx_params &p, int keys, int hits, int currentCutoff, bool perfect) {
    const int reduction = min(max(hits / p.hitReductionDiv, p.maxHitsReduction2), max(p.maximumMaxHitsReduction, keys / 8));
    int r = max(p.minApproxHitsToKeep, max(currentCutoff, hits - reduction));
    if (perfect) r = max(r, keys - PF::PERFECT_RED);
    return r;
}

// cross-lane exchange through the workspace (HBM / L2): writes of one phase are visible to every lane's reads of the next
__device__ __forceinline__ void wsfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ int adjustSite(const U &u, int a, int offset, int baseChrom) {
    const int below = u.c.toNumber(0, u.c.chromOf(a, baseChrom));
    return (a & u.c.siteMask) >= offset ? a - offset : below;
}

// ---------------------------------------------------------------------------------------------- heap stand-ins
// heap.peek(): the smallest (site, column) among the live lists.  Per lane also its smallest and second smallest live value, for
// the window test below.  L.v[j] holds a live list's value, INT_MAX once the list ran out (its last value stays in S.val) and for
// the columns past the last list, so the scan needs no liveness test; JT = the lists per lane rounded up to a multiple of 8.
template <int JT> __device__ __forceinline__ void minHeadT(const U &u, const Lists &L, int &site, int &center, int &best, int &second) {
    best = INT_MAX; second = INT_MAX; int bj = 0;
#pragma unroll
    for (int j = 0; j < JT; j++) {
        const int x = L.v[j];
        second = min(second, max(best, x));
        bj = x < best ? j : bj;
        best = min(best, x);
    }
    site = wmin(best);
    center = wmin(best == site ? bj * 64 + u.lane : INT_MAX);
}
__device__ __forceinline__ void minHead(const U &u, const Lists &L, int &site, int &center, int &best, int &second) {
    if (L.J <= 8) minHeadT<8>(u, L, site, center, best, second);
    else if (L.J <= 16) minHeadT<16>(u, L, site, center, best, second);
    else if (L.J <= 24) minHeadT<24>(u, L, site, center, best, second);
    else minHeadT<32>(u, L, site, center, best, second);
}
// how many columns hold a value in [lo, hi] (the reference's `chances` early exit only cuts the count short when it stays below the
// cutoff anyway), and the largest such value
__device__ __forceinline__ int countWindowFull(const U &u, const Lds &S, const Lists &L, int lo, int hi, int site, int &maxNearby) {
    int cnt = 0, mx = site;
    const unsigned span = (unsigned)(hi - lo);
    for (int j = 0; j < L.J; j++) {
        const int l = j * 64 + u.lane;
        const int x = l < L.n ? S.val[l] : DEADV;
        const bool in = (unsigned)(x - lo) <= span;
        cnt += in ? 1 : 0;
        mx = in ? max(mx, x) : mx;
    }
    maxNearby = wmax(mx);
    return wsum(cnt);
}
// The same, after a test that settles nearly every pop on a large genome: live lists sit at or above `site`, lists that ran out
// below it, so a second column can only be in the window if some lane's smallest live value (the centre's lane: its second
// smallest) is <= hi, or some lane's largest run-out value is >= lo.  Otherwise the window holds the centre alone.
__device__ __forceinline__ int countWindow(const U &u, const Lds &S, const Lists &L, int lo, int hi, int site, int center, int best, int second, int &maxNearby) {
    const int mine = (u.lane == (center & 63)) ? second : best;
    if (!__ballot(mine <= hi || L.dmax >= lo)) { maxNearby = site; return 1; }
    return countWindowFull(u, S, L, lo, hi, site, maxNearby);
}
// every live list's look-ahead, from its cursor: 64 x NB gathers in flight per j
__device__ __forceinline__ void refillAll(const U &u, Lds &S, Lists &L, int baseChrom) {
    LT_BEGIN(u);
#pragma unroll
    for (int j = 0; j < JM; j++) {
        if (j < L.J) {
            const int l = j * 64 + u.lane;
            if ((L.live >> j) & 1u) {
                const int row = L.rowW[l] + (int)(S.st[l] & 3);
                const int stop = L.stopW[l], off = S.off[l];
                L.rowW[l] = row;
                const int avail = stop - row - 1;
#pragma unroll
                for (int t = 0; t < NB; t++) S.nb[t][l] = adjustSite(u, L.sites[min(row + 1 + t, stop - 1)], off, baseChrom);
                S.st[l] = (uint8_t)((min(avail, NB) << 2) | (avail <= NB ? 0x80 : 0));
            }
        }
    }
    wsync();
    LT_END(u, 0);
}
// Pops the head of list `center`, the smallest (site, column) of the heap (QuadHeap.poll / add of the reference's inner loops,
// BBIndexPacBio.java:1618-1668, BBIndex.java:2420-2444: they pop while the heap's minimum sits on the site just looked at; the
// callers here look at a site once and keep popping while the minimum stays on it).  Returns true when the caller's loop ends: the
// list ran out and fewer than `cutoff` stay alive (or it ran out at all when `anyDeath`).
__device__ __forceinline__ bool popOne(const U &u, Lds &S, Lists &L, int site, int center, int cutoff, bool anyDeath, int baseChrom, unsigned &counter) {
    LT_BEGIN(u);
    int st = S.st[center];                                    // (one address for the whole wave: an LDS broadcast)
    if ((st & 3) >= ((st >> 2) & 3) && !(st & 0x80)) { refillAll(u, S, L, baseChrom); st = S.st[center]; }
    st = uni(st);
    const int t = st & 3;
    const bool dies = t >= ((st >> 2) & 3);
    const bool owner = u.lane == (center & 63);
    const int cj = center >> 6;
    counter += 1u;
    bool fin = false;
    if (dies) {
        L.nlive -= 1;
        if (owner) { L.live &= ~(1u << cj); L.dmax = max(L.dmax, site); L.v[cj] = INT_MAX; }
        fin = anyDeath || L.nlive < cutoff;
        if (fin) L.nlive = 0;
    } else {
        const int nv = S.nb[t][center];
        if (owner) { S.val[center] = nv; S.st[center] = (uint8_t)(st + 1); }
        if (owner) L.v[cj] = nv;                              // (cj is uniform: an indexed register move under the owner's EXEC bit)
        wsync();
    }
    LT_END(u, 1);
    return fin || L.nlive == 0;
}

// ---------------------------------------------------------------------------------------------- key-level scores
// BBIndex.maxQuickScore :2473-2487 (+ maxScoreZ :2948-2964) over ascending offsets off[0..n)
template <class PF, class T> __device__ __forceinline__ int maxQuickScoreL(const U &u, const T *off, const T *ksc, int n) {
    int sum = 0, cover = 0;
    for (int j = 0; j * 64 < n; j++) {
        const int l = j * 64 + u.lane;
        if (l < n) {
            const int o = off[l];
            sum += ksc[l];
            cover += (l < n - 1) ? min(u.k, off[l + 1] - o) : u.k;
        }
    }
    return wsum(sum) + PF::Z_MULT * wsum(cover) + Y_MULT * (off[n - 1] - off[0]);
}

// BBIndex.scoreZ2 :2882-2914: covered bases of the columns whose value lies in [center - MAX_INDEL, center + MAX_INDEL2]
template <class PF> __device__ __forceinline__ int scoreZ2L(const U &u, const Lds &S, const Lists &L, int centerVal, int numApproxHits) {
    if (numApproxHits == 1) return u.scoreZ1Key;
    const int maxLoc = centerVal + u.ix->p.maxIndel2, minLoc = max(0, centerVal - u.ix->p.maxIndel);
    int total = 0, nextOff = INT_MAX;                      // offset of the nearest in-range column above the chunk (uniform)
    for (int j = L.J - 1; j >= 0; j--) {
        const int l = j * 64 + u.lane;
        const int v = l < L.n ? S.val[l] : DEADV;
        const bool inr = v >= minLoc && v <= maxLoc;
        const u64 R = __ballot(inr);
        if (!R) continue;
        const int offs = inr ? (int)S.off[l] : 0;
        const u64 above = R & gt_mask(u.lane);
        const int src = above ? __builtin_ctzll(above) : u.lane;
        const int offAbove = __shfl(offs, src);
        const int nxt = above ? offAbove : nextOff;
        const int contrib = inr ? (nxt == INT_MAX ? u.k : min(u.k, nxt - offs)) : 0;
        total += wsum(contrib);
        nextOff = rl(offs, __builtin_ctzll(R));
    }
    return total * PF::Z_MULT;
}

// BBIndex.quickScore :2490-2511 with scoreLeft / scoreRight :2967-3035 and AbstractIndex.scoreY :52-78.  The chain "take the next
// column whose value lies within MAX_INDEL of the last column taken" is sequential in the columns it takes, not in the ones it
// passes over: per 64-column chunk a ballot finds the next one.
template <class PF> __device__ __forceinline__ int quickScoreL(const U &u, const Lds &S, const Lists &L, int centerIndex, int centerVal, int numApproxHits) {
    const int ksC = S.ksc[centerIndex];
    if (numApproxHits == 1) return ksC;
    const int maxIndel = u.ix->p.maxIndel;
    int x = ksC;
    // right side, ascending columns
    {
        int loc = centerVal;
        for (int j = centerIndex >> 6; j < L.J; j++) {
            const int l = j * 64 + u.lane;
            const bool valid = l > centerIndex && l < L.n;
            const int v = valid ? S.val[l] : DEADV;
            const int ks = valid ? (int)S.ksc[l] : 0;
            u64 pending = __ballot(valid && v >= 0);
            while (pending) {
                const u64 cand = pending & __ballot(absdif(v, loc) <= maxIndel);
                if (!cand) break;
                const int f = __builtin_ctzll(cand);
                const int vf = rl(v, f), offset = absdif(vf, loc);
                x += rl(ks, f);
                if (offset != 0) x -= min(u.indelPenalty + PF::INDEL_MULT * offset, u.maxPenalty);
                loc = vf;
                pending &= gt_mask(f);
            }
        }
    }
    // left side, descending columns
    {
        int loc = centerVal;
        for (int j = centerIndex >> 6; j >= 0; j--) {
            const int l = j * 64 + u.lane;
            const bool valid = l < centerIndex;
            const int v = valid ? S.val[l] : DEADV;
            const int ks = valid ? (int)S.ksc[l] : 0;
            u64 pending = __ballot(valid && v >= 0);
            while (pending) {
                const u64 cand = pending & __ballot(absdif(v, loc) <= maxIndel);
                if (!cand) break;
                const int f = hibit(cand);
                const int vf = rl(v, f), offset = absdif(vf, loc);
                x += rl(ks, f);
                if (offset != 0) x -= min(u.indelPenalty + PF::INDEL_MULT * offset, u.maxPenalty);
                loc = vf;
                pending &= lt_mask(f);
            }
        }
    }
    x -= centerIndex;
    int right = -1;                                         // last column that sits exactly on the centre's site
    for (int j = 0; j < L.J; j++) {
        const int l = j * 64 + u.lane;
        if (l < L.n && S.val[l] == centerVal) right = l;
    }
    right = wmax(right);
    return x + Y_MULT * ((int)S.off[right] - (int)S.off[centerIndex]);
}

// ---------------------------------------------------------------------------------------------- location-array scores
template <class PF> __device__ __forceinline__ int delApprox(int len) {       // calcDelScore(len, approximateGaps = true)
    if (len <= 0) return 0;
    int score = PF::DEL;
    if (len > MINGAP) { const int rem = len % 128, div = (len - 128) / 128; score += div * PF::GAP; len = rem + 128; }
    if (len > 80) { score += ((len - 80 + 3) / 4) * PF::DEL5; len = 80; }
    if (len > 20) { score += (len - 20) * PF::DEL4; len = 20; }
    if (len > 5) { score += (len - 5) * PF::DEL3; len = 5; }
    if (len > 1) score += (len - 1) * PF::DEL2;
    return score;
}
template <class PF> __device__ __forceinline__ int insShort(int n) { return PF::INS + (n > 1 ? (n - 1) * PF::INS2 : 0); }     // n in 1..5
template <class PF> __device__ __forceinline__ int subRun(int t) { return t > 5 ? PF::SUB3 : (t > 1 ? PF::SUB2 : PF::SUB); }

// calcAffineScore(locArray, baseScores, bases[, minContig]) over the LDS location array, 64 bases per step (the sequential state is
// recovered from ballot masks exactly as in index_probe_wave.hip's calcAffineScoreW)
template <class PF> __device__ __forceinline__ int calcAffineScoreL(const U &u, const Lds &S, int strand, int minContig) {
    const int blen = u.blen, lane = u.lane;
    int score = 0, carryLastLoc = -3, carryRun = 0, carryContig = 0, maxContig = 0;
    for (int base = 0; base < blen; base += 64) {
        const int p = base + lane;
        const bool valid = p < blen;
        const int loc = valid ? ld_loc(S, u, p) : 0;
        const int prev = (valid && p > 0) ? ld_loc(S, u, p - 1) : -1;
        const bool pos = valid && loc > 0, neg1 = valid && loc == -1;
        const u64 posM = __ballot(pos), n1M = __ballot(neg1);
        const u64 lt = lt_mask(lane);
        const u64 mlo = posM & lt;
        const int lastLoc = mlo ? ld_loc(S, u, base + hibit(mlo)) : carryLastLoc;
        int c = 0, ev = 0;                                   // ev: 1 equal, 2 restart, 3 indel
        if (pos) {
            const int bs = S.bsc[strand ? blen - 1 - p : p];
            if (loc == prev) { c = PF::MATCH2 + bs; ev = 1; }
            else if (loc == lastLoc || lastLoc < 0) { c = PF::MATCH + bs; ev = 2; }
            else if (loc < lastLoc) { c = PF::MATCH + bs + delApprox<PF>(lastLoc - loc + 1); ev = 3; }
            else { c = PF::MATCH + bs + insShort<PF>(min(loc - lastLoc + PF::INS_DIF_PLUS, 5)); ev = 3; }
        } else if (neg1) {
            const u64 nb = ~n1M & lt;
            const int t = nb ? lane - hibit(nb) : lane + 1 + carryRun;
            c = subRun<PF>(t);
        }
        score += wsum(c);
        if (minContig > 1) {
            const u64 EM = __ballot(ev == 1), SM = __ballot(ev == 2), IM = __ballot(ev == 3), RM = SM | IM;
            int cval = 0;
            if (ev == 1) {
                const u64 rlo = RM & lt;
                if (rlo) { const int r = hibit(rlo); cval = popc(EM & lt & gt_mask(r)) + 1 + (int)((SM >> r) & 1); }
                else cval = popc(EM & lt) + 1 + carryContig;
            } else if (ev == 2) cval = 1;
            maxContig = max(maxContig, wmax(cval));
            const u64 all = EM | RM;
            if (all) carryContig = rl(cval, hibit(all));
        }
        if (posM) carryLastLoc = rl(loc, hibit(posM));
        const int last = min(63, blen - 1 - base);
        if ((n1M >> last) & 1) {
            const u64 nbAll = ~n1M & (lt_mask(last) | (1ull << last));
            carryRun = nbAll ? last - hibit(nbAll) : last + 1 + carryRun;
        } else carryRun = 0;
    }
    if (minContig > 1 && maxContig < minContig) score = min(score, -50 * blen);
    return score;
}

// BBIndex.extendScore :2558-2833 (BBIndexPacBio.java:1907-2100): the columns in range are walked in column order, 64 at a time
template <class PF> __device__ __forceinline__ int extendScoreL(U &u, Lds &S, const Lists &L, int strand, int chrom, int centerVal) {
    const bbidx_params &p = u.ix->p;
    const int blen = u.blen, lane = u.lane, k = u.k;
    const int centerLoc = u.c.siteOf(centerVal);
    const int minVal = centerVal - p.maxIndel, maxVal = centerVal + p.maxIndel2;
    const uint8_t *ref = u.ix->chromArr[chrom];
    const int reflen = u.ix->chromArrLen[chrom];
    const uint8_t *rb = S.base[strand];
    u.cExtend += 1u;
    u.locBase = centerLoc - p.maxIndel;                          // every diagonal in range lies at or above it
    for (int i = lane; i < blen; i += 64) st_loc(S, u, i, -1);
    wsync();
    // backward from each key's last base; the first key in range runs through mismatches, the others stop at the first
    int keynum = 0;
    for (int j = 0; j < L.J; j++) {
        const int l = j * 64 + lane;
        const int value = l < L.n ? S.val[l] : DEADV;
        const u64 R = __ballot(value >= minVal && value <= maxVal);
        if (!R) continue;
        const int offs = ((R >> lane) & 1) ? (int)S.off[l] : 0;
        for (u64 m = R; m; m &= m - 1) {
            const int i = __builtin_ctzll(m);
            const int refbase = u.c.siteOf(rl(value, i)), c0 = rl(offs, i) + k - 1;
            keynum++;
            if (c0 < 0 || refbase + c0 >= reflen) continue;
            if (keynum == 1) {
                for (int base = 0; base <= c0; base += 64) {
                    const int q = base + lane;
                    if (q <= c0 && rb[q] == ref[refbase + q]) st_loc(S, u, q, refbase);
                }
                u.cRefBytes += (unsigned)(c0 + 1);
            } else {
                for (int top = c0; top >= 0; top -= 64) {
                    const int q = top - lane;
                    const bool valid = q >= 0;
                    const int old = valid ? ld_loc(S, u, q) : 0;
                    const u64 Em = __ballot(valid && old == refbase);
                    if (Em & 1) break;                                              // already holds this site: nothing compared
                    const bool mm = valid && rb[q] != ref[refbase + q];
                    const u64 stopM = Em | __ballot(mm);
                    const int s = stopM ? __builtin_ctzll(stopM) : 64;
                    if (valid && lane < s && (old < 0 || refbase == centerLoc)) st_loc(S, u, q, refbase);
                    u.cRefBytes += (unsigned)(min(s, min(64, top + 1)) + ((s < 64 && !((Em >> s) & 1)) ? 1 : 0));
                    if (s < 64) break;
                }
            }
            wsync();
        }
    }
    // forward from the base after each key: runs through mismatches over unassigned bases, stops on an assigned base once a
    // mismatch has been seen
    for (int j = 0; j < L.J; j++) {
        const int l = j * 64 + lane;
        const int value = l < L.n ? S.val[l] : DEADV;
        const u64 R = __ballot(value >= minVal && value <= maxVal);
        if (!R) continue;
        const int offs = ((R >> lane) & 1) ? (int)S.off[l] : 0;
        for (u64 m = R; m; m &= m - 1) {
            const int i = __builtin_ctzll(m);
            const int refbase = u.c.siteOf(rl(value, i));
            bool mmprev = false;
            for (int c = rl(offs, i) + k; c < blen; c += 64) {
                const int q = c + lane;
                const bool valid = q < blen && refbase + q < reflen;
                const int old = valid ? ld_loc(S, u, q) : -1;
                const bool A = valid && old >= 0;
                const u64 Em = __ballot(valid && old == refbase);
                if (Em & 1) break;
                const bool mm = valid && rb[q] != ref[refbase + q];
                const u64 mmM = __ballot(mm), validM = __ballot(valid), Am = __ballot(A);
                const bool mmBefore = mmprev || (mmM & lt_mask(lane)) != 0;
                const bool stop = valid && (old == refbase || (A && (mmBefore || mm)));
                const u64 stopM = __ballot(stop) | ~validM;
                const int s = stopM ? __builtin_ctzll(stopM) : 64;
                if (valid && lane < s && !mm && (old < 0 || refbase == centerLoc)) st_loc(S, u, q, refbase);
                unsigned cnt = (unsigned)s;
                if (s < 64 && ((validM >> s) & 1)) {
                    const bool Es = (Em >> s) & 1, As = (Am >> s) & 1;
                    const bool mmBeforeS = mmprev || (mmM & lt_mask(s)) != 0;
                    if (!Es && !(mmBeforeS && As)) cnt++;
                }
                u.cRefBytes += cnt;
                if (s < 64) break;
                mmprev = mmprev || mmM != 0;
            }
            wsync();
        }
    }
    for (int i = lane; i < blen; i += 64) if (rb[i] == 'N') st_loc(S, u, i, -2);
    wsync();
    return uni(calcAffineScoreL<PF>(u, S, strand, p.kfilter));
}

// BBIndex.makeGapArray :2837-2878 -- rare (a site spanning more than MINGAP + read length); one lane walks LDS
__device__ __forceinline__ int makeGapArrayL(const U &u, Lds &S, int minLoc, int minGap) {
    if (u.lane == 0) {
        // (the array is rewritten in place as the reference does; positions plus base indices still fit the 16-bit offsets)
        auto LA = [&](int i) -> int { return ld_loc(S, u, i); };
        auto SET = [&](int i, int v) { st_loc(S, u, i, v); };
        const int n = u.blen;
        int gaps = 0; bool doSort = false;
        if (LA(0) < 0) SET(0, minLoc);
        for (int i = 1; i < n; i++) {
            if (LA(i) < 0) SET(i, LA(i - 1) + 1); else SET(i, LA(i) + i);
            if (LA(i) < LA(i - 1)) doSort = true;
        }
        if (doSort) {
            for (int i = 1; i < n; i++) { const int v = LA(i); int j = i - 1; while (j >= 0 && LA(j) > v) { SET(j + 1, LA(j)); j--; } SET(j + 1, v); }
        }
        for (int i = 1; i < n; i++) if (LA(i) - LA(i - 1) > minGap) gaps++;
        int len = 0;
        if (gaps >= 1) {
            len = 2 + gaps * 2;
            if (len > BBIDX_MAX_GAPS) len = -1;
            else {
                S.gaps[0] = LA(0); S.gaps[len - 1] = LA(n - 1);
                for (int i = 1, j = 1; i < n; i++) if (LA(i) - LA(i - 1) > minGap) { S.gaps[j] = LA(i - 1); S.gaps[j + 1] = LA(i); j += 2; }
            }
        }
        *S.ngapsP = len;
    }
    wsync();
    return __builtin_amdgcn_readfirstlane(*S.ngapsP);
}

// SiteScore.setPerfect (current/stream/SiteScore.java:239-292), order-independent form as in index_probe_wave.hip
__device__ __forceinline__ void setPerfectL(const U &u, const Lds &S, int chrom, int strand, int start, int stop, int &perfectOut, int &semiOut) {
    const int blen = u.blen;
    perfectOut = 0; semiOut = 0;
    if (blen != stop - start + 1) return;
    const uint8_t *ref = u.ix->chromArr[chrom];
    const int reflen = u.ix->chromArrLen[chrom];
    const uint8_t *rb = S.base[strand];
    bool perfect = true;
    int refloc = start, readloc = 0, N = 0;
    const int mx = min(stop, reflen - 1), nlimit = blen / 2;
    if (start < 0) { N -= start; readloc -= start; refloc -= start; perfect = false; }
    if (stop >= reflen) { N += (stop - reflen + 1); perfect = false; }
    if (N > nlimit) return;
    bool anyHard = false, anyCN = false, anyBad = false;
    const int total = uni(mx - refloc + 1);
    for (int j0 = 0; j0 < total; j0 += 64) {
        const bool in = j0 + u.lane < total;
        const int j = in ? j0 + u.lane : total - 1;
        const int c = rb[readloc + j], r = ref[refloc + j];
        const bool bad = in && (c != r || c == 'N'), hard = bad && r != 'N', cn = bad && c == 'N';
        const u64 badM = __ballot(bad);
        if (badM) {
            anyBad = true;
            if (__ballot(hard)) { anyHard = true; break; }
            if (__ballot(cn)) anyCN = true;
            N += popc(badM);
            if (N > nlimit) break;
        }
    }
    if (anyHard || N > nlimit) return;
    const bool semi = !anyCN;
    semiOut = semi ? 1 : 0;
    perfectOut = (perfect && !anyBad && semi && N == 0) ? 1 : 0;
}
__device__ __forceinline__ bool overlap(int a1, int b1, int a2, int b2) { return a2 <= b1 && b2 >= a1; }

struct SiteOut { bbidx_site *v; int n, cap; bool overflow; };
struct PrevSite { int idx, chrom, strand, start, stop, score, perfect, semiperfect, ngaps; };

// ---------------------------------------------------------------------------------------------- findMaxQscore2
// BBIndex.findMaxQscore2 :2294-2450 (earlyExit = true, as prescanAllBlocks calls it)
template <class PF> __device__ __forceinline__ void findMaxQscore2L(U &u, Lds &S, Lists &L, int baseChrom, int prevMaxHits, bool perfectOnly,
                                                                    int &outQ, int &outHits) {
    const bbidx_params &p = u.ix->p;
    const int numHits = L.n;
    const int mqs = uni(maxQuickScoreL<PF>(u, S.off, S.ksc, numHits));      // of THIS cycle's lists (:2307)
    int topQscore = -999999999, maxHits = 0, approxHitsCutoff, indelCutoff;
    if (perfectOnly) { approxHitsCutoff = numHits; indelCutoff = 0; }
    else { approxHitsCutoff = max(prevMaxHits, min(p.minApproxHitsToKeep, numHits - 1)); indelCutoff = p.maxIndel2; }
    int lastSite = INT_MIN;
    while (L.nlive > 0) {
        approxHitsCutoff = uni(approxHitsCutoff); topQscore = uni(topQscore); maxHits = uni(maxHits); L.nlive = uni(L.nlive); u.cPrescan = uni(u.cPrescan);
        lastSite = uni(lastSite);
        int site, centerIndex, best, second;
        { LT_BEGIN(u); minHead(u, L, site, centerIndex, best, second); LT_END(u, 3); }
        if (site != lastSite) {                                 // a site is looked at once, with every list that sits on it still there
            lastSite = site;
            int unusedMax, approxHits;
            { LT_BEGIN(u); approxHits = countWindow(u, S, L, site - min(p.maxIndel, indelCutoff), site + p.maxIndel2, site, centerIndex, best, second, unusedMax); LT_END(u, 3); }
            if (approxHits >= approxHitsCutoff) {
                int qscore;
                { LT_BEGIN(u); qscore = quickScoreL<PF>(u, S, L, centerIndex, site, approxHits);
                  qscore += scoreZ2L<PF>(u, S, L, site, approxHits); LT_END(u, 2); }
                if (qscore > topQscore) {
                    maxHits = max(approxHits, maxHits);
                    approxHitsCutoff = max(approxHitsCutoff, approxHits - 1);
                    topQscore = qscore;
                    if (qscore >= mqs) break;
                }
            }
        }
        if (popOne(u, S, L, site, centerIndex, approxHitsCutoff, perfectOnly, baseChrom, u.cPrescan)) break;
    }
    outQ = topQscore; outHits = maxHits;
}

// ---------------------------------------------------------------------------------------------- slowWalk3
template <class PF> __device__ __forceinline__ void slowWalk3L(U &u, Lds &S, Lists &L, int strand, int numKeys, int mqs, int baseChrom_, SiteOut &ssl,
                                                               int *bestScores, bool allBasesCovered, int maxScore, bool fullyDefined) {
    const bbidx_params &p = u.ix->p;
    const int blen = u.blen, lane = u.lane;
    const int baseChrom = u.c.baseChrom(baseChrom_);
    const int numHits = L.n;
    const bool filter_by_qscore = numKeys >= 5;
    const int minScore = (int)(PF::MIN_SCORE_MULT * maxScore);
    const int minQuickScore = (int)(PF::MIN_QSCORE_MULT * mqs);
    int currentTopScore = bestScores[0];
    int cutoff = max(minScore, (int)(currentTopScore * PF::DYN_SCORE));
    int qcutoff = max(bestScores[2], minQuickScore);
    int bestqscore = bestScores[3], maxHits = bestScores[1], perfectsFound = bestScores[5];
    int approxHitsCutoff = calcApproxHitsCutoffP<PF>(p, numKeys, maxHits, p.minApproxHitsToKeep, currentTopScore >= maxScore);
    if (approxHitsCutoff > numHits) return;
    const bool shortCircuit = allBasesCovered && numKeys == numHits && filter_by_qscore;
    if (currentTopScore >= maxScore) qcutoff = max(qcutoff, (int)(mqs * DYN_QSCORE_PERFECT));

    PrevSite pv; pv.idx = -1; pv.chrom = pv.strand = pv.start = pv.stop = pv.score = pv.perfect = pv.semiperfect = pv.ngaps = 0;
    bool finished = false;
    int lastSite = INT_MIN;
    while (L.nlive > 0 && !finished) {
        approxHitsCutoff = uni(approxHitsCutoff); cutoff = uni(cutoff); qcutoff = uni(qcutoff); currentTopScore = uni(currentTopScore);
        maxHits = uni(maxHits); perfectsFound = uni(perfectsFound); bestqscore = uni(bestqscore); L.nlive = uni(L.nlive);
        pv.idx = uni(pv.idx); pv.chrom = uni(pv.chrom); pv.strand = uni(pv.strand); pv.start = uni(pv.start); pv.stop = uni(pv.stop);
        pv.score = uni(pv.score); pv.perfect = uni(pv.perfect); pv.semiperfect = uni(pv.semiperfect); pv.ngaps = uni(pv.ngaps);
        ssl.n = uni(ssl.n); ssl.overflow = uni(ssl.overflow); u.cWalk = uni(u.cWalk); u.cExtend = uni(u.cExtend); u.cRefBytes = uni(u.cRefBytes);
        lastSite = uni(lastSite);
        int site, centerIndex, maxNearbySite = 0, best, second;
        int approxHits = 0;
        { LT_BEGIN(u); minHead(u, L, site, centerIndex, best, second); LT_END(u, 3); }
        const bool fresh = site != lastSite;                    // a site is looked at once, with every list that sits on it still there
        lastSite = site;
        if (fresh) { LT_BEGIN(u); approxHits = countWindow(u, S, L, site - p.maxIndel, site + p.maxIndel2, site, centerIndex, best, second, maxNearbySite); LT_END(u, 3); }
        if (fresh && approxHits >= approxHitsCutoff) {
            int score;
            int qscore = filter_by_qscore ? quickScoreL<PF>(u, S, L, centerIndex, site, approxHits) : qcutoff;
            qscore += scoreZ2L<PF>(u, S, L, site, approxHits);
            int mapStart = site, mapStop = maxNearbySite;
            bool locArrayValid = false;
            if (qscore < qcutoff) score = -1;
            else {
                const int chrom = u.c.chromOf(site, baseChrom);
                if (shortCircuit && qscore == mqs) score = maxScore;
                else {
                    score = extendScoreL<PF>(u, S, L, strand, chrom, site);
                    locArrayValid = true;
                    int mn = INT_MAX, mx = INT_MIN;
                    for (int i = lane; i < blen; i += 64) { const int x = ld_loc(S, u, i); if (x > -1) { mn = min(mn, x); mx = max(mx, x); } }
                    mn = wmin(mn); mx = wmax(mx);
                    if (mn < 0 || mx < 0) score = -99999;
                    mapStart = u.c.toNumber(mn, chrom);
                    mapStop = u.c.toNumber(mx, chrom);
                }
                if (score == maxScore) {
                    qcutoff = max(qcutoff, (int)(mqs * DYN_QSCORE_PERFECT));
                    approxHitsCutoff = calcApproxHitsCutoffP<PF>(p, numKeys, maxHits, p.minApproxHitsToKeep, true);
                }
                if (score >= cutoff) { qcutoff = max(qcutoff, (int)(qscore * DYN_QSCORE)); bestqscore = max(qscore, bestqscore); }
            }
            if (score >= cutoff) {
                if (score > currentTopScore) {
                    maxHits = max(approxHits, maxHits);
                    approxHitsCutoff = calcApproxHitsCutoffP<PF>(p, numKeys, maxHits, approxHitsCutoff, currentTopScore >= maxScore);
                    cutoff = max(cutoff, (int)(score * PF::DYN_SCORE));
                    if (score >= maxScore) cutoff = max(cutoff, (int)(score * 0.95f));
                    currentTopScore = score;
                }
                const int chrom = u.c.chromOf(mapStart, baseChrom);
                const int site2 = u.c.siteOf(mapStart);
                const int site3 = u.c.siteOf(mapStop) + blen - 1;
                int ngaps = 0;
                if (site3 - site2 >= MINGAP + blen && locArrayValid) {
                    ngaps = makeGapArrayL(u, S, site2, MINGAP);
                    if (ngaps < 0) ngaps = 0;
                    if (ngaps > 0) {
                        if (lane == 0) { S.gaps[0] = min(S.gaps[0], site2); S.gaps[ngaps - 1] = max(S.gaps[ngaps - 1], site3); }
                        wsync();
                    }
                }
                ngaps = uni(ngaps);
                const bool perfect1 = (score == maxScore && fullyDefined);
                const bool inbounds = (site2 >= 0 && site3 < u.ix->chromLengths[chrom]);
                const bool havePrev = pv.idx >= 0;
                bool makeNew = false, withGaps = false;
                int wb = 0;
                if (inbounds && ngaps == 0 && havePrev && pv.chrom == chrom && pv.strand == strand && overlap(pv.start, pv.stop, site2, site3)) {
                    const int betterScore = max(score, pv.score);
                    const int minStart = min(pv.start, site2), maxStop = max(pv.stop, site3);
                    const bool perfect2 = (pv.score == maxScore && fullyDefined);
                    const bool shortEnough = (maxStop - minStart < 2 * blen);
                    bbidx_site *pd = &ssl.v[pv.idx];
                    if (pv.start == site2 && pv.stop == site3) {
                        pv.score = betterScore;
                        pv.perfect = (pv.perfect || perfect1 || perfect2) ? 1 : 0;
                        if (pv.perfect) pv.semiperfect = 1;
                        wb = 1;
                    } else if (shortEnough && pv.start == site2 && !pv.semiperfect) {
                        if (perfect2) { }
                        else if (perfect1) {
                            pv.stop = site3;
                            if (!pv.perfect) perfectsFound++;
                            pv.perfect = pv.semiperfect = 1;
                        } else {
                            pv.stop = maxStop;
                            setPerfectL(u, S, pv.chrom, pv.strand, pv.start, pv.stop, pv.perfect, pv.semiperfect);
                        }
                        pv.score = betterScore;
                        wb = 2;
                    } else if (shortEnough && pv.stop == site3 && !pv.semiperfect) {
                        if (perfect2) { }
                        else if (perfect1) {
                            pv.start = site2;
                            if (!pv.perfect) perfectsFound++;
                            pv.perfect = pv.semiperfect = 1;
                        } else {
                            pv.start = minStart;
                            setPerfectL(u, S, pv.chrom, pv.strand, pv.start, pv.stop, pv.perfect, pv.semiperfect);
                        }
                        pv.score = betterScore;
                        wb = 3;
                    } else makeNew = true;
                    wb = uni(wb);
                    if (wb && lane == 0) {
                        if (wb == 2) { pd->stop = pv.stop; if (pv.ngaps) pd->gaps[pv.ngaps - 1] = pv.stop; }
                        if (wb == 3) { pd->start = pv.start; if (pv.ngaps) pd->gaps[0] = pv.start; }
                        pd->perfect = pv.perfect; pd->semiperfect = pv.semiperfect; pd->score = pv.score;
                    }
                } else if (inbounds) { makeNew = true; withGaps = true; }
                pv.chrom = uni(pv.chrom); pv.strand = uni(pv.strand); pv.start = uni(pv.start); pv.stop = uni(pv.stop);
                pv.score = uni(pv.score); pv.perfect = uni(pv.perfect); pv.semiperfect = uni(pv.semiperfect); perfectsFound = uni(perfectsFound);
                if (uni(makeNew)) {
                    int sp = perfect1 ? 1 : 0, ssemi = sp;
                    if (!perfect1) setPerfectL(u, S, chrom, strand, site2, site3, sp, ssemi);
                    sp = uni(sp); ssemi = uni(ssemi);
                    const int sg = withGaps ? ngaps : 0;
                    if (ssl.n >= ssl.cap) { ssl.overflow = true; finished = true; }
                    else {
                        int wv = 0;
                        switch (lane) {
                            case 0: wv = chrom; break; case 1: wv = strand; break; case 2: wv = site2; break; case 3: wv = site3; break;
                            case 4: wv = approxHits; break; case 5: wv = score; break; case 6: wv = sp; break; case 7: wv = ssemi; break;
                            case 8: wv = sg; break;
                            default: wv = (lane < 9 + sg) ? S.gaps[lane - 9] : 0; break;
                        }
                        if (lane < 25) ((int *)&ssl.v[ssl.n])[lane] = wv;
                        const int idx = ssl.n++;
                        bool stopNow = false;
                        if (sp) {
                            if (!havePrev || !pv.perfect || !(pv.chrom == chrom && pv.strand == strand && overlap(site2, site3, pv.start, pv.stop))) {
                                perfectsFound++;
                                if (p.quitAfterTwoPerfects && perfectsFound >= 2) stopNow = true;
                            }
                        }
                        pv.idx = idx; pv.chrom = chrom; pv.strand = strand; pv.start = site2; pv.stop = site3; pv.score = score;
                        pv.perfect = sp; pv.semiperfect = ssemi; pv.ngaps = sg;
                        if (stopNow) finished = true;
                    }
                }
            }
        }
        if (uni(finished)) break;
        if (popOne(u, S, L, site, centerIndex, approxHitsCutoff, false, baseChrom, u.cWalk)) break;
    }
    bestScores[0] = max(bestScores[0], currentTopScore);
    bestScores[1] = max(bestScores[1], maxHits);
    bestScores[2] = max(bestScores[2], qcutoff);
    bestScores[3] = max(bestScores[3], bestqscore);
    bestScores[4] = mqs;
    bestScores[5] = perfectsFound;
}

// ---------------------------------------------------------------------------------------------- greedy trim
// Solver.valueOfElement (current/align2/Solver.java:97-151)
__device__ __forceinline__ long long valueOfElement(const int *offsets, int noffsets, const int *lengths, float keyWeight, int chunk,
                                                    const int *lists, int numlists, int index, long long pointsPerSite) {
    const long long PPL = 30000, PPB1 = 6000, BONUS_END = 40000, WIDTH = 5500, SPACING = -30;
    if (numlists < 1) return 0;
    const int prospect = lists[index];
    if (lengths[prospect] == 0) return -999999;
    long long valuep = PPL + (PPL * 2 / numlists) + ((PPL * 10) / lengths[prospect]);
    const long long valuem = pointsPerSite * lengths[prospect];
    if (prospect == 0 || prospect == noffsets - 1) valuep += BONUS_END;
    if (numlists == 1) { valuep += (WIDTH + PPB1) * chunk; return ((long long)__fmul_rn((float)valuep, keyWeight)) + valuem; }
    const int first = lists[0], last = lists[numlists - 1];
    const int offL = (prospect == first ? -1 : offsets[lists[index - 1]]);
    const int offP = offsets[prospect];
    const int offR = (prospect == last ? offsets[noffsets - 1] + 1 : offsets[lists[index + 1]]);
    const int oldL = offP - offL, oldR = offR - offP, newS = offR - offL;
    valuep += (long long)((oldL * oldL + oldR * oldR) - (newS * newS)) * SPACING;
    int uniquelyCovered;
    if (prospect == first) uniquelyCovered = offR - offP;
    else if (prospect == last) uniquelyCovered = offP - offL;
    else { const int b = offR - (offL + chunk); uniquelyCovered = b > 0 ? b : 0; }
    if (prospect == first || prospect == last) valuep += (PPB1 + WIDTH) * uniquelyCovered;
    else valuep += PPB1 * uniquelyCovered;
    return ((long long)__fmul_rn((float)valuep, keyWeight)) + valuem;
}
__device__ __forceinline__ long long rl64(long long v, int l) {
    return (long long)(((u64)(unsigned)rl((int)(v >> 32), l) << 32) | (unsigned)rl((int)v, l));
}

// BBIndex.trimExcessHitListsByGreedy :266-350 (+ Solver.findWorstGreedy :46-95).  keyW / lenW (= COUNTS of the key, 0 once dropped)
// are updated in place; listsW is scratch.
template <class PF> __device__ __forceinline__ int trimByGreedyL(const U &u, int *keyW, const int *offW, const int *kscW, int *lenW, int *listsW, int n, int maxHitLists) {
    const DevIndex &ix = *u.ix;
    const bbidx_params &p = ix.p;
    const int lane = u.lane;
    const float inv = __fdiv_rn(1.0f, (float)u.baseKeyHitScore);
    const int limit = max(PF::SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch]) * n;
    const int limit2 = max(PF::SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch2]);
    const int limit3 = max(PF::SMALL_LIST, ix.lengthHistogram[p.maxShortestListToSearch]);
    int sum = 0, initialHitCount = 0, shortest = INT_MAX - 1, longest = 0;
    for (int j = 0; j * 64 < n; j++) {
        const int l = j * 64 + lane;
        const int x = l < n ? lenW[l] : 0;
        sum += wsum(x);
        initialHitCount += popc(__ballot(x != 0));
        shortest = min(shortest, wmin(x > 0 ? x : INT_MAX - 1));
        longest = max(longest, wmax(x));
    }
    if (initialHitCount < p.minApproxHitsToKeep) return initialHitCount;
    if (shortest > limit3 && !p.slow) {
        for (int j = 0; j * 64 < n; j++) { const int l = j * 64 + lane; if (l < n) keyW[l] = -1; }
        wsfence();
        return 0;
    }
    if (longest < PF::SMALL_LIST) return initialHitCount;      // whichever list the first round picks, the loop returns there
    int hitsCount = initialHitCount;
    const long long EARLY = -50LL * 2000;
    while (hitsCount >= p.minApproxHitsToKeep && (sum > limit || sum / initialHitCount > limit2 || hitsCount > maxHitLists)) {
        sum = uni(sum); hitsCount = uni(hitsCount);
        // lists[]: positions of the remaining keys, ascending
        int m = 0;
        for (int j = 0; j * 64 < n; j++) {
            const int l = j * 64 + lane;
            const bool keep = l < n && lenW[l] > 0;
            const u64 M = __ballot(keep);
            if (keep) listsW[m + popc(M & lt_mask(lane))] = l;
            m += popc(M);
        }
        wsfence();
        // the first strict prefix minimum that follows a prefix minimum below EARLY ends the scan; otherwise the global minimum
        long long runMin = LLONG_MAX, worstValue64 = 0; int worstIndex = -1; bool early = false;
        for (int j = 0; j * 64 < hitsCount && !early; j++) {
            const int i = j * 64 + lane;
            long long v = LLONG_MAX;
            if (i < hitsCount) v = valueOfElement(offW, n, lenW, __fmul_rn((float)kscW[i], inv), p.k, listsW, hitsCount, i, p.pointsPerSite);
            long long pm = v;                                   // inclusive prefix minimum within the chunk
            for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(pm, d); if (lane >= d) pm = min(pm, t); }
            long long ex = __shfl_up(pm, 1);
            if (lane == 0) ex = LLONG_MAX;
            ex = min(ex, runMin);                               // exclusive prefix minimum over everything before i
            const bool upd = i < hitsCount && v < ex;
            const u64 earlyM = __ballot(upd && i != 0 && ex < EARLY);
            if (earlyM) {
                const int f = __builtin_ctzll(earlyM);
                worstIndex = j * 64 + f; worstValue64 = rl64(v, f); early = true;
            } else {
                const u64 updM = __ballot(upd);
                if (updM) { const int f = hibit(updM); worstIndex = j * 64 + f; worstValue64 = rl64(v, f); }
                runMin = min(runMin, rl64(pm, 63));
            }
        }
        const int worstValue = worstValue64 < INT_MIN ? INT_MIN : (worstValue64 > INT_MAX ? INT_MAX : (int)worstValue64);
        const int worst = listsW[worstIndex];
        const int lenWorst = lenW[worst];
        sum -= lenWorst;
        if (worstValue > 0 || lenWorst < PF::SMALL_LIST) return hitsCount;
        hitsCount--;
        if (lane == 0) { lenW[worst] = 0; keyW[worst] = -1; }
        wsfence();
    }
    return hitsCount;
}

// stable compaction of the key arrays: keeps the entries whose key is >= 0
__device__ __forceinline__ int compactKeys(const U &u, int *keyW, int *offW, int *kscW, int *lenW, int n) {
    int m = 0;
    for (int j = 0; j * 64 < n; j++) {
        const int l = j * 64 + u.lane;
        const bool act = l < n;
        const int key = act ? keyW[l] : -1, off = act ? offW[l] : 0, ksc = act ? kscW[l] : 0, len = act ? lenW[l] : 0;
        const bool keep = key >= 0;
        const u64 M = __ballot(keep);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the chunk is in registers before any lane overwrites part of it
        if (keep) { const int d = m + popc(M & lt_mask(u.lane)); keyW[d] = key; offW[d] = off; kscW[d] = ksc; lenW[d] = len; }
        m += popc(M);
    }
    wsfence();
    return m;
}

__device__ inline int base_num_fast(int b) { return base_num(b); }

template <class PF> __global__ __launch_bounds__(64) void probe_long_kernel(const LongParams Q) {
    extern __shared__ __align__(16) unsigned char ldsRaw[];
    Lds S;
    {
        const int KM = lds_km(Q.maxKeys), LM = lds_lm(Q.maxLen);
        int *w = reinterpret_cast<int *>(ldsRaw);
        S.loc = reinterpret_cast<unsigned short *>(w); w += LM / 2;
        S.val = w; w += KM;
        for (int j = 0; j < NB; j++) { S.nb[j] = w; w += KM; }
        S.gaps = w; w += BBIDX_MAX_GAPS;
        S.ngapsP = w; w += 4;
        S.ksc = reinterpret_cast<short *>(w); S.off = S.ksc + KM;
        S.st = reinterpret_cast<uint8_t *>(S.off + KM);
        S.base[0] = S.base[1] = nullptr; S.bsc = nullptr;
    }
    const Params &P = Q.P;
    const int lane = threadIdx.x & 63;
    const DevIndex &ix = P.ix;
    const bbidx_params &p = ix.p;
    int *ws = Q.ws + (long long)blockIdx.x * (WS_ARRAYS * KMAX);
    int *keyW = ws, *offW = ws + KMAX, *kscW = ws + 2 * KMAX, *lenW = ws + 3 * KMAX, *origW = ws + 4 * KMAX, *listsW = ws + 5 * KMAX;
    Lists L;
    L.rowW = ws + 6 * KMAX; L.stopW = ws + 7 * KMAX;
    U u;
    u.ix = &ix;
    u.c.shift = 31 - p.chromBits; u.c.siteMask = (int)(0xFFFFFFFFu >> (p.chromBits + 1));
    u.c.cpb = 1 << p.chromBits; u.c.lowMask = u.c.cpb - 1; u.c.highMask = ~u.c.lowMask;
    u.k = p.k; u.baseKeyHitScore = BASE_HIT_SCORE * p.k;
    u.indelPenalty = PF::indelPenalty(u.baseKeyHitScore);
    u.maxPenalty = u.baseKeyHitScore - (1 + u.baseKeyHitScore / 8);
    u.scoreZ1Key = PF::Z_MULT * p.k;
    u.lane = lane; u.blen = 0;
    u.cPrescan = u.cWalk = u.cExtend = u.cRefBytes = 0;
#ifdef BBIDXL_TIMERS
    for (int q = 0; q < 5; q++) u.tm[q] = 0;
#endif
    unsigned cSites = 0;

    for (;;) {
        long long r = 0;
        if (lane == 0) r = (long long)atomicAdd(&P.queue[3], 1u);
        r = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)r);
        if (r >= P.nreads) break;
        if (P.onlyPending && P.nsites[r] != NSITES_PENDING) continue;
        int result = 0;
        const bbidx_read rr = P.reads[r];
        const int blen = rr.len;
        int n = rr.nkeys;
        u.blen = blen;
        bool done = false;
        if (n < 1 || blen < p.k) { result = 0; done = true; }
        else if (n > Q.maxKeys || blen > Q.maxLen) { result = -2; done = true; }
        if (!done) do {
            const uint8_t *bP = P.bases + rr.bases_off;
            const int8_t *qP = P.baseScores + rr.bases_off;
            const int *koff = P.keyinfo + rr.keys_off, *kscore = koff + n;
            int sumBS = 0; bool undefinedBase = false;
            // minus strand: into the caller's rc buffer when there is one, else into the wave's workspace (slots 8-9); this wave reads
            // it back through other lanes, hence the release / acquire pair below
            uint8_t *rcG = P.rcOut ? P.rcOut + rr.bases_off : reinterpret_cast<uint8_t *>(ws + 8 * KMAX);
            uint8_t *stage = reinterpret_cast<uint8_t *>(S.loc);     // the plus strand for the key extraction below (loc is not in use yet)
            for (int i = lane; i < blen; i += 64) {
                const int b = bP[i], q = qP[i];
                rcG[blen - 1 - i] = (uint8_t)complement_extended(b);
                stage[i] = (uint8_t)b;
                sumBS += q;
                if (base_num(b) < 0 || b >= 128) undefinedBase = true;
            }
            sumBS = wsum(sumBS);
            const bool fullyDefined = __ballot(undefinedBase) == 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            wsync();
            S.base[0] = bP; S.base[1] = rcG; S.bsc = qP;
            // KeyRing.makeKeys; COUNTS of every key
            bool badOrder = false;
            for (int j = 0; j * 64 < n; j++) {
                const int l = j * 64 + lane;
                if (l < n) {
                    const int off = koff[l];
                    int key = 0;
                    for (int q = off; q < off + p.k; q++) { const int x = base_num(stage[q]); if (x < 0) { key = -1; break; } key = (key << 2) | x; }
                    keyW[l] = key; origW[l] = key; offW[l] = off; kscW[l] = kscore[l];
                    lenW[l] = key >= 0 ? ix.counts[key] : 0;
                    if (l > 0 && off < koff[l - 1]) badOrder = true;
                }
            }
            if (__ballot(badOrder)) { result = -2; break; }      // the coverage arithmetic needs ascending offsets (KeyRing.makeOffsets3 gives them)
            wsfence();
            auto countHits = [&](int maxLen) __attribute__((always_inline)) -> int {
                int cnt = 0;
                for (int j = 0; j * 64 < n; j++) {
                    const int l = j * 64 + lane;
                    const bool act = l < n;
                    const int len = act ? lenW[l] : 0;
                    const bool v = act && origW[l] >= 0 && len > 0 && len < maxLen;
                    if (act) keyW[l] = v ? origW[l] : -1;
                    cnt += popc(__ballot(v));
                }
                return cnt;
            };
            const int maxLen = p.maxUsableLength;
            int numHits = countHits(maxLen);
            if (numHits > 0) {
                const int trigger = (3 * n) / 4;
                if (numHits < PF::RELAX1 && numHits < trigger) numHits = countHits((maxLen * 3) / 2);
                if (numHits < PF::RELAX2 && numHits < trigger) numHits = countHits(maxLen * 2);
                if (numHits < PF::RELAX3 && numHits < trigger) numHits = countHits(maxLen * 3);
                if (numHits < PF::RELAX4 && numHits < trigger) numHits = countHits(maxLen * 5);
            }
            wsfence();
            const int nOriginal = n;
            if (numHits < n) n = compactKeys(u, keyW, offW, kscW, lenW, n);
            if (p.trimByGreedy && n > 0) {
                const int maxLists = max((int)(PF::HIT_FRACTION * n), PF::MIN_LISTS_RETAIN);
                numHits = trimByGreedyL<PF>(u, keyW, offW, kscW, lenW, listsW, n, maxLists);
            }
            numHits = uni(numHits);
            if (numHits < p.minApproxHitsToKeep || n < 1) { result = 0; break; }
            if (numHits < n) n = compactKeys(u, keyW, offW, kscW, lenW, n);
            n = uni(n);
            const int mqs = uni(maxQuickScoreL<PF>(u, offW, kscW, n));     // the minus strand's arrays are these mirrored: same sum, coverage, span
            int bestScores[6] = {0, 0, 0, 0, 0, 0};
            const bool prescan = p.prescanQscore && numHits >= 5;
            int hitsCutoff = 0, qscoreCutoff = (int)(PF::MIN_QSCORE_MULT * mqs);
            bool allBasesCovered = true, pretend;
            {
                const int off0 = offW[0], offLast = offW[n - 1];
                if (off0 != 0 || offLast != blen - p.k) allBasesCovered = false;
                else {
                    bool hole = false;
                    for (int j = 0; j * 64 < n; j++) { const int l = j * 64 + lane; if (l > 0 && l < n && offW[l] > offW[l - 1] + p.k) hole = true; }
                    if (__ballot(hole)) allBasesCovered = false;
                }
                pretend = allBasesCovered || n >= nOriginal - 4 || (n >= 9 && (offLast - off0 + p.k) > max(40, (int)(blen * .75f)));
            }
            const int cpb = u.c.cpb;
            int ncycles = 0;
            for (int chrom = p.minChrom; chrom <= p.maxChrom; chrom = ((chrom & u.c.highMask) + cpb)) ncycles += 2;
            if (ncycles > 64) { result = -2; break; }
            int precount = n, prescore = mqs;                 // lane c holds the prescan result of cycle c

            // BBIndex.getHits :354-391 + the heap fill of slowWalk3 / findMaxQscore2: the lists of one (block, strand) cycle
            auto makeLists = [&](int block, int strand, int baseChrom, int minHits) __attribute__((always_inline)) -> int {
                int nh = 0;
                for (int j = 0; j * 64 < n; j++) {
                    const int l = j * 64 + lane;
                    bool hit = false; int start = 0, len = 0, first = 0, off = 0, ksc = 0;
                    if (l < n) {
                        const int src = strand ? n - 1 - l : l;          // KeyRing.reverseComplementKeys / reverseOffsets
                        const int key = keyW[src];
                        const KeyEntry e = ix.fused[block][key];
                        off = strand ? blen - (offW[src] + p.k) : offW[src];
                        ksc = kscW[src];
                        const int cnt = strand ? e.cntRC : e.cnt;
                        start = strand ? e.startR : e.startF; len = strand ? e.lenR : e.lenF; first = strand ? e.firstR : e.firstF;
                        hit = cnt > 0 && len > 0 && first != -1;
                    }
                    const u64 M = __ballot(hit);
                    if (hit) {
                        const int d = nh + popc(M & lt_mask(lane));
                        L.rowW[d] = start; L.stopW[d] = start + len;
                        S.off[d] = (short)off; S.ksc[d] = (short)ksc;
                        S.val[d] = adjustSite(u, first, off, baseChrom);
                        S.st[d] = 0;
                    }
                    nh += popc(M);
                }
                L.n = L.nlive = nh; L.J = (nh + 63) >> 6; L.sites = (GlobalIntsT)ix.sites[block];
                wsync();
                wsfence();
                L.live = 0; L.dmax = INT_MIN;
#pragma unroll
                for (int j = 0; j < JM; j++) {
                    L.v[j] = INT_MAX;
                    if (j < L.J) {
                        const int l = j * 64 + lane;
                        if (l < nh) { L.v[j] = S.val[l]; L.live |= 1u << j; }
                    }
                }
                if (nh >= minHits && nh > 0) refillAll(u, S, L, baseChrom);
                return nh;
            };

            bool dead = false;
            if (prescan) {                                       // prescanAllBlocks :642-741
                int bestqscore = 0, maxHits = 0, minHitsToScore = p.minApproxHitsToKeep, cycle = 0; bool earlyOut = false;
                for (int chrom = p.minChrom; chrom <= p.maxChrom && !earlyOut; chrom = ((chrom & u.c.highMask) + cpb)) {
                    const int baseChrom = u.c.baseChrom(chrom);
                    const int block = baseChrom >> p.chromBits;
                    for (int pmi = 0; pmi < 2 && !earlyOut; pmi++, cycle++) {
                        const int nh = makeLists(block, pmi, baseChrom, minHitsToScore);
                        if (nh < minHitsToScore) { if (lane == cycle) { prescore = -9999; precount = 0; } }
                        else {
                            int tq, th;
                            { LT_BEGIN(u); findMaxQscore2L<PF>(u, S, L, baseChrom, minHitsToScore, bestqscore >= mqs && pretend, tq, th); LT_END(u, 4); }
                            tq = uni(tq); th = uni(th);
                            if (lane == cycle) { prescore = tq; precount = th; }
                            bestqscore = max(tq, bestqscore); maxHits = max(maxHits, th);
                            if (bestqscore >= mqs && pretend) { minHitsToScore = max(minHitsToScore, maxHits); earlyOut = true; }
                        }
                    }
                }
                bestScores[1] = max(bestScores[1], maxHits);
                bestScores[3] = max(bestScores[3], bestqscore);
                if (bestScores[1] < p.minApproxHitsToKeep) dead = true;
                else if ((float)bestScores[3] < __fmul_rn((float)mqs, PF::MIN_QSCORE_MULT2)) dead = true;
                else if (bestScores[3] >= mqs && pretend) {
                    hitsCutoff = calcApproxHitsCutoffP<PF>(p, n, bestScores[1], p.minApproxHitsToKeep, true);
                    qscoreCutoff = max(qscoreCutoff, (int)(bestScores[3] * DYN_QSCORE_PERFECT));
                } else {
                    hitsCutoff = calcApproxHitsCutoffP<PF>(p, n, bestScores[1], p.minApproxHitsToKeep, false);
                    qscoreCutoff = max(qscoreCutoff, (int)(bestScores[3] * PRESCAN_QSCORE_THRESH));
                }
            }
            if (uni(dead)) { result = 0; break; }
            hitsCutoff = uni(hitsCutoff); qscoreCutoff = uni(qscoreCutoff);

            const int maxScore = PF::MATCH + (blen - 1) * PF::MATCH2 + sumBS;               // msa.maxQuality(baseScores)
            SiteOut ssl; ssl.v = P.sites + r * (long long)P.maxSites; ssl.n = 0; ssl.cap = P.maxSites; ssl.overflow = false;
            int cycle = 0; bool quit = false;
            for (int chrom = p.minChrom; chrom <= p.maxChrom && !quit; chrom = ((chrom & u.c.highMask) + cpb)) {
                const int baseChrom = u.c.baseChrom(chrom);
                const int block = baseChrom >> p.chromBits;
                for (int strand = 0; strand < 2 && !quit; strand++, cycle++) {
                    for (int q = 0; q < 6; q++) bestScores[q] = uni(bestScores[q]);
                    ssl.n = uni(ssl.n); ssl.overflow = uni(ssl.overflow); cycle = uni(cycle); quit = uni(quit);
                    if (!prescan || rl(precount, cycle) >= hitsCutoff || rl(prescore, cycle) >= qscoreCutoff) {
                        const int nh = makeLists(block, strand, baseChrom, p.minApproxHitsToKeep);
                        if (nh >= p.minApproxHitsToKeep)
                            { LT_BEGIN(u); slowWalk3L<PF>(u, S, L, strand, n, mqs, chrom, ssl, bestScores, allBasesCovered, maxScore, fullyDefined); LT_END(u, 4); }
                    }
                    if (p.quitAfterTwoPerfects && bestScores[5] >= 2) quit = true;
                }
            }
            result = ssl.overflow ? -1 : ssl.n;
            cSites += (unsigned)ssl.n;
        } while (0);
        if (lane == 0) P.nsites[r] = result;
        wsync();
    }
    if (P.stats && lane == 0) {
        unsigned long long *st = P.stats + 8 * (blockIdx.x % STAT_SHARDS);
#ifdef BBIDXL_TIMERS
        for (int q = 0; q < 5; q++) atomicAdd(&st[q], u.tm[q] >> 10);
#else
        atomicAdd(&st[0], (unsigned long long)u.cPrescan); atomicAdd(&st[1], (unsigned long long)u.cWalk);
        atomicAdd(&st[2], (unsigned long long)u.cExtend); atomicAdd(&st[3], (unsigned long long)u.cRefBytes);
        atomicAdd(&st[4], (unsigned long long)cSites);
#endif
    }
}

}  // namespace bbidxl

// Workspace and launch.  `ws` = blocks * WS_ARRAYS * KMAX ints (bbidx_long_workspace_ints per block).
namespace bbidxl {
// the batch's longest read and largest key count, into queue[8..9] (zeroed by the caller with the rest of the queue words)
__global__ void long_maxima_kernel(const bbidx_read *reads, long long n, unsigned int *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0, nk = 0;
    if (i < n) { len = reads[i].len; nk = reads[i].nkeys; }
    for (int d = 32; d >= 1; d >>= 1) { len = max(len, __shfl_xor(len, d, 64)); nk = max(nk, __shfl_xor(nk, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], (unsigned)max(len, 0)); atomicMax(&out[1], (unsigned)max(nk, 0)); }
}
}  // namespace bbidxl

long long bbidx_long_workspace_ints_per_block() { return (long long)bbidxl::WS_ARRAYS * bbidxl::KMAX; }
int bbidx_long_lds_bytes() { return bbidxl::lds_bytes(bbidxl::KMAX - 1, bbidxl::LMAX); }      // the largest shape

static const void *long_kernel_fn(int profile) {
    return profile ? (const void *)bbidxl::probe_long_kernel<bbidxl::ProfPacBio> : (const void *)bbidxl::probe_long_kernel<bbidxl::ProfBBMap>;
}
constexpr int LONG_MAX_BLOCKS_PER_CU = 8, LONG_MAX_DEVICES = 64;
// per device (a process may hold index contexts on several GPUs, and the overflow tier's helper thread probes too): the CU count, and
// whether the kernels' dynamic LDS limit has been raised there
static std::mutex g_longMutex;
static int g_longCUs[LONG_MAX_DEVICES];
static bool g_longReady[LONG_MAX_DEVICES][2];
static int long_cus(int profile) {                     // of the CURRENT device (the callers have set it to the context's)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LONG_MAX_DEVICES) return 0;
    std::lock_guard<std::mutex> lock(g_longMutex);
    if (!g_longReady[dev][profile ? 1 : 0]) {
        if (hipFuncSetAttribute(long_kernel_fn(profile), hipFuncAttributeMaxDynamicSharedMemorySize, bbidx_long_lds_bytes()) != hipSuccess) return 0;
        hipDeviceProp_t prop;
        g_longCUs[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        g_longReady[dev][profile ? 1 : 0] = true;
    }
    return g_longCUs[dev];
}

// blocks the workspace has to be sized for: the most any launch uses
int bbidx_long_blocks(int profile) { return long_cus(profile) * LONG_MAX_BLOCKS_PER_CU; }

int bbidx_launch_long(const bbidx::Params &P, hipStream_t stream, int profile, int *ws, int blocks) {
    static thread_local char msg[256];
    // the batch's maxima size the LDS layout, and with it the number of resident wavefronts (one host round trip; the kernel runs
    // for milliseconds per read)
    unsigned int mx[2] = {0, 0};
    {
        const long long n = P.nreads;
        hipLaunchKernelGGL(bbidxl::long_maxima_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P.reads, n, P.queue + 8);
        if (hipMemcpyAsync(mx, P.queue + 8, sizeof mx, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            snprintf(msg, sizeof msg, "probe_long_kernel: reading the batch's maxima failed: %s", hipGetErrorString(hipGetLastError()));
            bbmap_set_error(msg);
            return BBMAP_E_HIP;
        }
    }
    bbidxl::LongParams Q;
    Q.P = P; Q.ws = ws;
    Q.maxKeys = (int)mx[1] < bbidxl::KMAX - 1 ? (int)mx[1] : bbidxl::KMAX - 1;          // reads beyond the largest shape are declined (-2)
    Q.maxLen = (int)mx[0] < bbidxl::LMAX ? (int)mx[0] : bbidxl::LMAX;
    if (Q.maxKeys < 1) Q.maxKeys = 1;
    if (Q.maxLen < 16) Q.maxLen = 16;
    const int lds = bbidxl::lds_bytes(Q.maxKeys, Q.maxLen);
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, long_kernel_fn(profile), 64, (size_t)lds) != hipSuccess || per < 1) per = 1;
    if (per > LONG_MAX_BLOCKS_PER_CU) per = LONG_MAX_BLOCKS_PER_CU;
    const int cus = long_cus(profile);
    long long nb = (long long)(cus > 0 ? cus : 256) * per;
    if (nb > blocks) nb = blocks;
    if (nb > P.nreads) nb = P.nreads;
    if (nb < 1) nb = 1;
    if (profile) hipLaunchKernelGGL(bbidxl::probe_long_kernel<bbidxl::ProfPacBio>, dim3((unsigned)nb), dim3(64), lds, stream, Q);
    else hipLaunchKernelGGL(bbidxl::probe_long_kernel<bbidxl::ProfBBMap>, dim3((unsigned)nb), dim3(64), lds, stream, Q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(msg, sizeof msg, "probe_long_kernel launch failed: %s", hipGetErrorString(e));
        bbmap_set_error(msg);
        return BBMAP_E_HIP;
    }
    return BBMAP_OK;
}
