// k-mer index probe (align2.BBIndex.findAdvanced) on gfx950 -- ONE READ PER WAVEFRONT.
//
// BBIndex.find is an order-dependent state machine per read, so a read cannot be split across workgroups; but every
// step inside it is a small data-parallel operation over the read's K keys (K <= 64) or its L bases:
//   * lane i owns key/list i: the list cursor, its current site, offset and key score live in that lane's registers;
//   * the reference's binary heap (QuadHeap) only ever exposes its minimum under (site, column): a DPP min-reduction
//     plus a ballot gives the same element; the "how many lists are near this site" scans are ballots + popcounts;
//   * extendScore / setPerfect / calcAffineScore walk the read 64 bases per step against coalesced reference bytes,
//     the per-base location array lives in LDS; sequential carry (first stop, streak lengths, last defined
//     location) is recovered from ballot masks with count-leading/trailing-zero arithmetic;
//   * control flow is wave-uniform by construction (every decision is taken on a ballot/readlane value), so EXEC
//     stays full and the cross-lane operations are always legal.
//     Uniform state is also DECLARED uniform (wavep::uni = v_readfirstlane at the top of each loop): LLVM's uniformity
//     analysis cannot see through a ds_bpermute or a loop with a lane-dependent exit, and one misjudged value turns every
//     later branch into EXEC-mask bookkeeping;
//   * the heap walk pops in batches (batchPop: every head below the smallest second entry of any list) and jumps over
//     entries that cannot reach the hit cutoff (bulkSkip); both reproduce the reference's pop sequence exactly.
// Nothing per-read is kept in scratch memory: the per-lane kernel (index_probe.hip) spilled ~10 KB per read to HBM.
// Reads with more than 64 keys or longer than the instantiation's LDS arrays are marked NSITES_PENDING and taken by the
// per-lane kernel afterwards.  Four instantiations: <long lists?, LDS read capacity 160 | 400> (see bbidx_launch_wave).
//
// Functions follow current/align2/BBIndex.java exactly as index_probe.hip does (same citations); the two kernels and
// the CPU oracle are compared SiteScore by SiteScore in tests/test_index_gpu.py.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>

#include "bbmap_amd.h"
#include "index_common.h"
#include "wave_prims.h"

void bbmap_set_error(const char *msg);

namespace bbidxw {
using namespace bbidx;
using namespace wavep;

#ifdef BBIDX_BATCH_STATS
#define BST(u, i, v) ((u).ph[i] += (unsigned)(v))
#else
#define BST(u, i, v) do { } while (0)
#endif
#ifdef BBIDX_NO_WORK_COUNTERS       // experiment: what the four work counters (list entries, extensions, reference bytes) cost
#define WORK(c, v) do { } while (0)
#else
#define WORK(c, v) ((c) += (v))
#endif
#ifdef BBIDX_CYC_STATS              // debug build: {cycles offered to the whole-cycle walk, declined, entries, candidates, candidates visited}
#define CST(u, i, v) ((u).ph[i] += (unsigned)(v))
#else
#define CST(u, i, v) do { } while (0)
#endif
// debug builds with cycle counters: -DBBIDX_PHASE_TIMERS=1 the read's phases {keys + lookup, trim / setup, prescan, walk, extend};
// =2 inside the walk, extendScore left out {lists + cycle load, candidate stepping, quick scores + site bookkeeping,
// single-key batch of a walk at cutoff 1, sequential heap walk}; =3 inside the prescan
// {lists, cycle gather, candidate filter, candidate loop, sequential fallback}
#define PH_ADD_(u, i) do { const unsigned long long now_ = __builtin_readcyclecounter(); (u).ph[i] += (unsigned)(now_ - (u).phT); (u).phT = now_; } while (0)
#define PH_SKIP_(u) do { (u).phT = __builtin_readcyclecounter(); } while (0)
#if defined(BBIDX_PHASE_TIMERS) && BBIDX_PHASE_TIMERS == 2
#define PH_MARK(u, i) PH_SKIP_(u)
#define PH_WALK(u, i) PH_ADD_(u, i)
#define PH_SKIPW(u) PH_SKIP_(u)
#define PH_PRE(u, i) do { } while (0)
#elif defined(BBIDX_PHASE_TIMERS) && BBIDX_PHASE_TIMERS == 3
#define PH_MARK(u, i) PH_SKIP_(u)
#define PH_WALK(u, i) do { } while (0)
#define PH_SKIPW(u) do { } while (0)
#define PH_PRE(u, i) PH_ADD_(u, i)
#elif defined(BBIDX_PHASE_TIMERS)
#define PH_MARK(u, i) PH_ADD_(u, i)
#define PH_WALK(u, i) do { } while (0)
#define PH_SKIPW(u) do { } while (0)
#define PH_PRE(u, i) do { } while (0)
#else
#define PH_MARK(u, i) do { } while (0)
#define PH_WALK(u, i) do { } while (0)
#define PH_SKIPW(u) do { } while (0)
#define PH_PRE(u, i) do { } while (0)
#endif

constexpr int WAVES_PER_BLOCK = 4;
constexpr int WMAXLEN = 400;      // longest read the wave kernel takes (longer ones go to the per-lane kernel)
constexpr int WSHORTLEN = 160;    // its second instantiation: batches whose reads are at most this long (bbidx_set_max_read_len)
                                  // need 4.2 KB of LDS per wave instead of 6.1 KB and run 8 waves per SIMD instead of 6

template <int WLEN>
struct WaveLds {
    int loc[WLEN];                // the per-base location array of extendScore
    int xch[3][64];               // lane <-> lane exchange (compaction), greedy-trim tables
    int gaps[BBIDX_MAX_GAPS];
    int ngaps;
    uint8_t base[2][WLEN + 8];    // [0] the read as given, [1] its reverse complement
    int8_t bsc[WLEN + 8];         // base scores of the plus strand
    int8_t code[WLEN + 8];        // AminoAcid.baseToNumber of the plus strand (-1 = undefined)
};

// wave-uniform state of one read
struct U {
    const DevIndex *ix;
    Codec c;
    int k, baseKeyHitScore, indelPenalty, indelPenaltyMult, maxPenalty, scoreZ1Key;
    int lane, blen;
    unsigned cPrescan, cWalk, cExtend, cRefBytes;
#if defined(BBIDX_PHASE_TIMERS) || defined(BBIDX_BATCH_STATS) || defined(BBIDX_CYC_STATS)
    unsigned ph[5]; unsigned long long phT;     // debug build: cycles per phase instead of the work counters
#endif
};

// one list per lane (the reference's Quad heap entries), compacted: lanes 0..n-1
#ifndef BBIDX_LIST_BUF
#define BBIDX_LIST_BUF 2      // look-ahead entries per list in registers (4 until round 4: since the whole-cycle walk takes most cycles, the
#endif                        // sequential path's buffers are worth less than their registers: 63.2 -> 62.3 ms per 2 M reads; 1 entry: 67.9)
constexpr int NB = BBIDX_LIST_BUF;
typedef const int __attribute__((address_space(1))) *GlobalInts;   // global_load instead of flat_load (no LDS counter traffic)
#ifndef BBIDX_BULK_MIN
#define BBIDX_BULK_MIN 256
#endif
constexpr int BULK_MIN_ENTRIES = BBIDX_BULK_MIN;
#ifndef BBIDX_BATCH_MIN
#define BBIDX_BATCH_MIN 64
#endif
#ifndef BBIDX_BWAIT
#define BBIDX_BWAIT 0
#endif
constexpr int BATCH_RETRY = BBIDX_BWAIT;                  // ordinary rounds after a batch that found nothing to do
constexpr int BATCH_MIN_ENTRIES = BBIDX_BATCH_MIN;        // batchPop likewise (the tests also run a build with both at 0)      // bulkSkip is tried only when a strand's lists hold at least this many entries
constexpr int LANE_UNUSED = -(1 << 30);   // `value` of the lanes past the last list: outside every [minsite, maxsite] window
struct WL {
    int row, stop, value, offs, ksc;
    int hv;                    // the list's head for the "smallest head" search: value while the list lives, INT_MAX once it ran
                               // out (the reference keeps counting an exhausted list's last value as a nearby hit)
    int nb[NB], nbuf;          // nb[j] = sites[row + 1 + j] for j < nbuf: the entries after the cursor.  A list is popped one
                               // entry at a time by one lane, and a load issued for that lane alone costs the wave a full
                               // memory round trip (s_waitcnt counts in order), so the buffers of ALL lists are refilled
                               // together, 64 x NB gathers in flight, whenever a popped list finds its buffer empty
    int n, nlive;              // uniform
    int bulk;                  // uniform: > 0 = rounds until bulkSkip may be tried again, < 0 = lists too short to bother
    int bwait;                 // uniform: rounds until batchPop is tried again, < 0 = lists too short to bother
    GlobalInts sites;          // uniform
};

__device__ __forceinline__ int adjustSite(const U &u, int a, int offset, int baseChrom) {
    // a site in the first `offset` bases of its chromosome maps to position 0 of that chromosome (branch-free: both forms
    // are a handful of ALU ops, and a per-lane branch here would sit in the innermost loop of the probe)
    const int below = u.c.toNumber(0, u.c.chromOf(a, baseChrom));
    return (a & u.c.siteMask) >= offset ? a - offset : below;
}

// BBIndex.maxQuickScore :2473-2487 over lanes 0..n-1 (offsets ascending: the coverage of maxScoreZ :2948-2964 is
// sum(min(k, next - this)) + k)
__device__ __forceinline__ int maxQuickScoreW(const U &u, int off, int ksc, int n) {
    const int nxt = __shfl_down(off, 1);
    int contrib = 0;
    if (u.lane < n) contrib = ksc + Z_MULT * ((u.lane < n - 1) ? min(u.k, nxt - off) : u.k);
    return wsum(contrib) + Y_MULT * (rl(off, n - 1) - rl(off, 0));
}

// BBIndex.scoreZ2 :2882-2914
__device__ __forceinline__ int scoreZ2W(const U &u, int value, int offs, int centerVal, int numApproxHits, int numHits) {
    if (numApproxHits == 1) return u.scoreZ1Key;
    const int maxLoc = centerVal + u.ix->p.maxIndel2, minLoc = max(0, centerVal - u.ix->p.maxIndel);
    const bool inr = u.lane < numHits && value >= minLoc && value <= maxLoc;
    const u64 R = __ballot(inr);
    const u64 above = R & gt_mask(u.lane);
    const int j = above ? __builtin_ctzll(above) : u.lane;
    const int offj = __shfl(offs, j);
    const int contrib = inr ? (above ? min(u.k, offj - offs) : u.k) : 0;
    return wsum(contrib) * Z_MULT;
}

// BBIndex.quickScore :2490-2511 with scoreLeft/scoreRight :2967-3035.  The chain "accept a key if it lies within
// MAX_INDEL of the last accepted one" is sequential; when every key within MAX_INDEL of the centre sits exactly on the
// centre (no indel between keys) the chain degenerates to a sum, otherwise it is walked with scalar readlanes.
__device__ __forceinline__ int quickScoreW(const U &u, int value, int ksc, int offs, int centerIndex, int centerVal, int numApproxHits, int numHits) {
    const int ksC = rl(ksc, centerIndex);
    if (numApproxHits == 1) return ksC;
    const int maxIndel = u.ix->p.maxIndel;
    const bool act = u.lane < numHits;
    const u64 eqM = __ballot(act && value == centerVal);
    const u64 inM = __ballot(act && absdif(value, centerVal) <= maxIndel);
    int x;
    if (eqM == inM) {
        x = wsum((act && value == centerVal) ? ksc : 0);
    } else {
        x = ksC;
        for (int dir = -1; dir <= 1; dir += 2) {
            int loc = centerVal;
            for (int i = centerIndex + dir; i >= 0 && i < numHits; i += dir) {
                const int li = rl(value, i);
                const int offset = absdif(li, loc);
                if (offset <= maxIndel) {
                    x += rl(ksc, i);
                    if (offset != 0) x -= min(u.indelPenalty + u.indelPenaltyMult * offset, u.maxPenalty);
                    loc = li;
                }
            }
        }
    }
    x -= centerIndex;
    const int rightIndex = hibit(eqM);
    return x + Y_MULT * (rl(offs, rightIndex) - rl(offs, centerIndex));
}

// reloads every live list's look-ahead buffer from its cursor
__device__ __forceinline__ void refillLists(WL &L) {
    const bool live = L.hv != INT_MAX;
    const int avail = live ? L.stop - L.row - 1 : 0;
    const int last = live ? L.stop - 1 : 0;                // slots past the list's end re-read its last entry (never used)
#pragma unroll
    for (int j = 0; j < NB; j++) L.nb[j] = L.sites[min(L.row + 1 + j, last)];
    L.nbuf = min(avail, NB);
}

// Pops every list whose head equals `site`, in (site, column) order (QuadHeap.poll/add of the reference's inner
// loop, BBIndex.java:1637-1666 and :2420-2444).  When the caller's loop must end (a list ran out and fewer than
// `cutoff` lists remain, or perfectOnly) L.nlive is set to 0.
__device__ __forceinline__ void popSite(const U &u, WL &L, int site, int cutoff, bool perfectOnly, int baseChrom, unsigned &counter) {
    site = uni(site); cutoff = uni(cutoff); perfectOnly = uni(perfectOnly);
    for (;;) {
        L.nlive = uni(L.nlive); counter = uni(counter);
        const u64 Pm = mask_eq(L.hv, site);
        if (!Pm) break;
        const bool hit = L.hv == site;
        const int row = L.row + 1;
        const u64 D = Pm & __builtin_amdgcn_sicmp(row, L.stop, 39);      // lists whose last entry this was
        if (D) {
            const int nd = popc(D);
            const int jexit = perfectOnly ? 1 : max(1, L.nlive - cutoff + 1);
            if (jexit <= nd) {
                u64 m = D;
                for (int j = 1; j < jexit; j++) m &= m - 1;
                const int d = __builtin_ctzll(m);
                WORK(counter, (unsigned)popc(Pm & (lt_mask(d) | (1ull << d))));
                L.nlive = 0;
                return;
            }
            L.nlive -= nd;
        }
        WORK(counter, (unsigned)popc(Pm));
        const bool dies = hit && row >= L.stop;
        if (__ballot(hit && !dies && L.nbuf == 0)) refillLists(L);
        // per-lane cursor update as selects: no EXEC juggling in the innermost loop
        const bool adv = hit && !dies;
        const int nv = adjustSite(u, L.nb[0], L.offs, baseChrom);
        L.row = adv ? row : L.row; L.value = adv ? nv : L.value;
        L.hv = hit ? (dies ? INT_MAX : nv) : L.hv;
#pragma unroll
        for (int j = 0; j + 1 < NB; j++) L.nb[j] = adv ? L.nb[j + 1] : L.nb[j];
        L.nbuf -= adv ? 1 : 0;
        if (L.nlive == 0) break;
    }
}

// Skips, in one step, list entries the reference's loop would pop one by one without ever scoring them.
//
// A site s is scored only if at least `cutoff` lists hold a value in [s - lo, s + maxIndel2].  Let e be the number of
// exhausted lists whose last value can still fall into such a window (value >= site - lo), m = cutoff - 1 - e, and hc the
// (m+1)-th smallest head.  Heads only grow, so for every later site s < T = hc - maxIndel2 the lists ranked m+1 and up lie
// beyond s + maxIndel2: at most m live lists and e exhausted ones are in its window, fewer than `cutoff`.  Every list whose
// head is below T therefore moves straight to its first entry >= T (found by galloping, all lists of the wave probing
// together), except that a list never gives up its last entry here: running out has consequences (the loop's exit rule)
// that stay with popSite.  The pops are counted as the reference would count them.
__device__ __forceinline__ void bulkSkip(const U &u, WL &L, int site, int lo, int cutoff, int baseChrom, unsigned &counter) {
    const bbidx_params &p = u.ix->p;
    const bool live = L.hv != INT_MAX;
    const int e = popc(__ballot(!live && u.lane < L.n && L.value >= site - lo));
    const int kth = cutoff - e;                              // 1-based rank of the head that bounds the skip
    if (kth < 2) { L.bulk = 8; return; }
    int T = INT_MAX;
    if (kth <= L.nlive) {
        int rank = 0;
        for (u64 m = __ballot(live); m; m &= m - 1) {
            const int i = __builtin_ctzll(m);
            const int v = rl(L.hv, i);
            rank += (v < L.hv || (v == L.hv && i < u.lane)) ? 1 : 0;
        }
        const u64 K = __ballot(live && rank == kth - 1);
        const int hc = rl(L.hv, __builtin_ctzll(K));
        T = hc - p.maxIndel2;
    }
    const int last = L.stop - 1;
    bool act = live && L.hv < T && L.row < last;
    if (!__ballot(act)) { L.bulk = 8; return; }
    // first index in (row, last] whose adjusted value is >= T, or `last`: exponential probe, then bisection
    int a = L.row, b = last, step = 1;
    bool bracketed = false;
    while (__ballot(act)) {
        const int probe = bracketed ? a + ((b - a) >> 1) : ((b - a > step) ? a + step : b);   // (row indices reach 2^31: no a + b)
        const int idx = act ? probe : 0;
        const int v = adjustSite(u, L.sites[idx], L.offs, baseChrom);
        if (act) {
            if (v < T) {
                a = probe;
                if (!bracketed) { step <<= 1; if (probe == b) act = false; }       // every entry is below T: stop at the last
            } else { b = probe; bracketed = true; }
            if (bracketed && b - a <= 1) act = false;
        }
    }
    const bool moved = live && L.hv < T && L.row < last;
    const int np = moved ? b : L.row;
    const int total = wsum(np - L.row);
    WORK(counter, (unsigned)total);
    if (moved) {
        L.row = np;
        L.value = adjustSite(u, L.sites[np], L.offs, baseChrom);
        L.hv = L.value;
    }
    refillLists(L);
    L.bulk = total < 8 ? 16 : 0;
}

// Pops several list heads in one step.  Every head below W = the smallest *second* entry of any list (a list at its last
// entry contributes its head: its pop is a death and stays with popSite) comes before every other entry in the global
// order, so the reference pops exactly these heads next, in ascending order.  For each of them the window count "as of
// that pop" is evaluated against the other lists' positions at that moment: lists with a smaller head have moved to
// their next entry, all others are where they are.  The batch ends in front of the first site that reaches the cutoff
// (that site gets the full treatment in the ordinary path, from the same state the reference has there); the heads in
// front of it are advanced together.  In the prescan a site seen by one list only has a fixed quick score
// (keyScore + scoreZ1Key, quickScore/scoreZ2 with one approximate hit) and feeds nothing but the running maximum, so at a
// cutoff of 1 such sites are folded into `topQscore`/`maxHits` here; anything else at that cutoff ends the batch.
// Returns the number of entries consumed (0: nothing done).
__device__ __forceinline__ int batchPop(const U &u, WL &L, int lo, int hi, int cutoff, bool prescan, int &topQscore, int &maxHits, int mqs,
                               int baseChrom, unsigned &counter) {
    const bool live = L.hv != INT_MAX;
    const bool hasNext = live && L.row + 1 < L.stop;
    if (__ballot(hasNext && L.nbuf == 0)) refillLists(L);
    const int nx = adjustSite(u, L.nb[0], L.offs, baseChrom);            // the list's second entry (where hasNext)
    const int W = wmin(hasNext ? nx : L.hv);
    const u64 B = __builtin_amdgcn_sicmp(L.hv, W, 40);                    // heads < W
    if (popc(B) < 2) return 0;
    const int ceff = prescan ? max(cutoff, 2) : cutoff;
    const unsigned span = (unsigned)(lo + hi);
    int firstQual = INT_MAX;
    for (u64 m = B; m; m &= m - 1) {
        const int s = rl(L.hv, __builtin_ctzll(m));
        const int cur = L.hv < s ? nx : L.value;                          // where each list stands when s is popped
        const int cnt = popc(mask_ule((unsigned)cur - (unsigned)(s - lo), span));
        if (cnt >= ceff) firstQual = min(firstQual, s);
    }
    const int bound = min(W, firstQual);
    const bool pop = L.hv < bound;
    const int np = popc(__builtin_amdgcn_sicmp(L.hv, bound, 40));
    if (np == 0) return 0;
    if (prescan && cutoff <= 1) {
        const int q = wmax(pop ? L.ksc + u.scoreZ1Key : INT_MIN);
        if (q >= mqs) return 0;                                           // the reference's loop would end there
        if (q > topQscore) { maxHits = max(maxHits, 1); topQscore = q; }
    }
    WORK(counter, (unsigned)np);
    L.row += pop ? 1 : 0;
    L.value = pop ? nx : L.value; L.hv = pop ? nx : L.hv;
#pragma unroll
    for (int j = 0; j + 1 < NB; j++) L.nb[j] = pop ? L.nb[j + 1] : L.nb[j];
    L.nbuf -= pop ? 1 : 0;
    return np;
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-cycle form of the heap walk (long-list variant).  The reference merges the K lists of a (block, strand) cycle one
// entry at a time; on a large genome nearly every entry it pops is ISOLATED: no entry of any list within
// [site - MAX_INDEL, site + MAX_INDEL2], so exactly one list "hits" (approxHits = 1) and the site either cannot reach the hit
// cutoff (cutoff >= 2: the pop changes nothing) or contributes the fixed quick score keyScore + scoreZ1Key to a running
// maximum (prescan at cutoff <= 1).  Only a few sites per cycle are anything else.  So instead of K-way merging one pop per
// wave step:
//   1. all entries of the cycle are gathered into LDS at once (64 gathers in flight per instruction);
//   2. presence maps over 32 kbp buckets (two hash functions, 2048 bits each) mark the entries that have company in their
//      own or a neighbouring bucket: the CANDIDATES (a hash collision only makes a harmless extra candidate);
//   3. the candidates are visited in ascending site order with exactly the state the reference has there: every list's
//      head is its first entry >= site (a cursor per lane = per list), an exhausted list keeps its last value;
//   4. popSite's exit rule -- the loop ends at the first pop at which a list runs out and fewer than `cutoff` stay alive --
//      needs no visit: the lists' last entries are ranked once, the pop that ends the loop after site s at cutoff c is
//      max(the (n-c+1)-th smallest last entry, the smallest last entry >= s), and isolated pops at cutoff >= 2 do nothing else;
//   5. (prescan only) before a candidate is looked at, the isolated entries below it are folded into the running maximum
//      while the cutoff is still <= 1 (at that cutoff the loop can only end with the very last entry).
// The sequential functions above remain for the cycles this form declines (more than CYC_EMAX entries or CYC_CMAX candidates,
// a single list, perfect-only prescans, walks that start at a hit cutoff below 2) and for the plain variant.
#ifndef BBIDX_CYCLE
#define BBIDX_CYCLE 1               // 0: always the sequential heap walk (for A/B measurements)
#endif
constexpr int CYC_EMAX = 384, CYC_CMAX = 64;
#ifndef BBIDX_CYC_MAPW
#define BBIDX_CYC_MAPW 192
#endif
constexpr int CYC_MAPW = BBIDX_CYC_MAPW;        // words per presence map (6,144 slots)
struct CycleLds {
    int ent[CYC_EMAX];               // adjusted sites, list after list
    unsigned short isoq[CYC_EMAX];   // 0 for a candidate, else the entry's prescan quick score keyScore + scoreZ1Key
    // Presence maps over hashed 64 kbp buckets of two grids, the second shifted by half a bucket: two entries no farther apart than
    // 32,768 share a bucket in at least one of them.  once = a slot that was hit, twice = hit more than once.
    unsigned once[2][CYC_MAPW], twice[2][CYC_MAPW];
};
struct CycleLanes { int lo, len, last, p, rank; };   // per lane = per list: its slice of ent[], last value, cursor, rank of `last`

__device__ __forceinline__ unsigned cyc_slot(unsigned v, int grid) {
    const unsigned b = (v + (grid ? 32768u : 0u)) >> 16;
    return __umulhi(b * (grid ? 0x85EBCA6Bu : 0x9E3779B1u), (unsigned)(32 * CYC_MAPW));
}
__device__ __forceinline__ bool cyc_bit(const unsigned *m, unsigned sl) { return (m[sl >> 5] >> (sl & 31)) & 1; }

// Gathers the cycle's entries.  Returns the number of candidates (their sites in cs[]), or -1 when the cycle does not fit.
// xrow/xoff/xlo (S.xch) and xksc (S.loc) are scratch of the load phase; cs aliases xrow..xlo afterwards.
// cycleGather: the entries into C.ent / C.isoq (and, with MAPS, into the presence maps); false when the cycle does not fit.
template <bool MAPS, int WLEN> __device__ __forceinline__ bool cycleGather(U &u, CycleLds &C, WaveLds<WLEN> &S, const WL &L, int baseChrom, CycleLanes &cl, int &E) {
    const int lane = u.lane, n = L.n;
    int *xrow = S.xch[0], *xoff = S.xch[1], *xlo = S.xch[2], *xksc = S.loc;
    unsigned *startBits = (unsigned *)(S.loc + 64);           // one bit per entry index: a list starts here (12 words)
    int *startPre = S.loc + 80;                                //   and the number of starts in the words before each of them
    const int len = lane < n ? L.stop - L.row : 0;
    int inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    E = rl(inc, 63);
    if (E > CYC_EMAX || n < 2) return false;
    cl.lo = inc - len; cl.len = len; cl.p = 0;
    wsync();
    xlo[lane] = cl.lo; xksc[lane] = L.ksc; xrow[lane] = L.row; xoff[lane] = L.offs;      // (lanes past the last list: lo = E)
    if (MAPS) for (int i = lane; i < CYC_MAPW; i += 64) { C.once[0][i] = 0; C.once[1][i] = 0; C.twice[0][i] = 0; C.twice[1][i] = 0; }
    if (lane < 12) startBits[lane] = 0;
    wsync();
    if (lane < n) atomicOr(&startBits[cl.lo >> 5], 1u << (cl.lo & 31));
    wsync();
    {
        const int c = lane < 12 ? __builtin_popcount(startBits[lane]) : 0;
        int ps = c;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const int t = __shfl_up(ps, d); if (lane >= d) ps += t; }
        if (lane < 12) startPre[lane] = ps - c;
    }
    wsync();
    // Every lane takes four CONSECUTIVE entries: the list of the first one from the start bits (a rank), the others follow by
    // stepping over at most one list border each.
    for (int base = 0; base < E; base += 256) {
        const int e0 = base + 4 * lane;
        int a = 0;
        if (e0 < E) { const int w = e0 >> 5; a = startPre[w] + __builtin_popcount(startBits[w] & (0xFFFFFFFFu >> (31 - (e0 & 31)))) - 1; }
        int raw[4], qs[4], oj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = e0 + q;
            const bool in = e < E;
            if (q > 0 && in && a + 1 < n && e >= xlo[a + 1]) a++;
            oj[q] = xoff[a]; qs[q] = xksc[a] + u.scoreZ1Key;
            raw[q] = in ? L.sites[xrow[a] + (e - xlo[a])] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = e0 + q;
            if (e < E) {
                const int v = adjustSite(u, raw[q], oj[q], baseChrom);
                C.ent[e] = v; C.isoq[e] = (unsigned short)min(qs[q], 65535);
#pragma unroll
                for (int g = 0; MAPS && g < 2; g++) {
                    const unsigned sl = cyc_slot((unsigned)v, g), bit = 1u << (sl & 31);
                    const unsigned old = atomicOr(&C.once[g][sl >> 5], bit);
                    atomicOr(&C.twice[g][sl >> 5], old & bit);
                }
            }
        }
    }
    wsync();
    PH_PRE(u, 1);
    cl.last = len > 0 ? C.ent[cl.lo + len - 1] : INT_MAX;
    cl.rank = -2;                                              // ranked on first use (cycleExitSite)
    wsync();                                                   // the load phase's scratch is dead: cs[] may be written
    return true;
}
template <int WLEN> __device__ __forceinline__ int cycleLoad(U &u, CycleLds &C, WaveLds<WLEN> &S, const WL &L, int baseChrom, CycleLanes &cl, int &E) {
    if (!cycleGather<true>(u, C, S, L, baseChrom, cl, E)) return -1;
    const int lane = u.lane;
    int *xoff = S.xch[1], *xlo = S.xch[2], *cs = S.xch[0];
    // pass 1: entries that share a bucket with another one in either grid (a superset of the entries with company: a shared
    // bucket can be wider than the window, a slot can be shared by two buckets); pass 2: the exact window among those
    const int lo = min(u.ix->p.maxIndel, u.ix->p.maxIndel2), hi = u.ix->p.maxIndel2;
    int nflag = 0;
    for (int base = 0; base < E; base += 64) {
        const int e = base + lane;
        bool flag = false; int v = 0;
        if (e < E) {
            v = C.ent[e];
            flag = cyc_bit(C.twice[0], cyc_slot((unsigned)v, 0)) || cyc_bit(C.twice[1], cyc_slot((unsigned)v, 1));
        }
        const u64 M = __ballot(flag);
        if (nflag + popc(M) > CYC_CMAX) return -1;
        if (flag) { const int slot = nflag + popc(M & lt_mask(lane)); cs[slot] = v; xlo[slot] = e; }   // xlo (S.xch[2]) holds the entry index
        nflag += popc(M);
    }
    wsync();
    // an entry needs a visit only if some OTHER entry lies in its hit window [v - lo, v + hi]: only then can a second list hit
    int ncand = 0;
    for (int base = 0; base < nflag; base += 64) {
        const int c = base + lane;
        const bool in = c < nflag;
        const int v = in ? cs[c] : 0;
        bool company = false;
        for (int i = 0; i < nflag; i++) { const unsigned d = (unsigned)(cs[i] - v) + (unsigned)lo; company |= (i != c) && d <= (unsigned)(lo + hi); }
        company &= in;
        if (company) C.isoq[xlo[c]] = 0;
        const u64 M = __ballot(company);
        wsync();                                               // every lane has read cs[base..] before it is overwritten below
        if (company) xoff[ncand + popc(M & lt_mask(lane))] = v;    // xoff (S.xch[1]): the surviving candidates
        ncand += popc(M);
    }
    wsync();
    for (int i = lane; i < ncand; i += 64) { const int v = xoff[i]; cs[i] = v; }
    wsync();
    PH_PRE(u, 2);
    return ncand;
}
// smallest candidate site above `prev` (INT_MAX: none left)
__device__ __forceinline__ int cycleNext(const U &u, const int *cs, int ncand, int prev) {
    int m = INT_MAX;
    for (int i = u.lane; i < ncand; i += 64) { const int v = cs[i]; if (v > prev) m = min(m, v); }
    return wmin(m);
}
// every list's state when `site` is the heap minimum: head = first entry >= site, value = head or, once exhausted, the last entry
__device__ __forceinline__ void cycleSeek(const U &u, const CycleLds &C, CycleLanes &cl, WL &L, int site) {
    const bool mine = u.lane < L.n;
    bool adv = mine && cl.p < cl.len && C.ent[cl.lo + cl.p] < site;
    while (__ballot(adv)) { if (adv) { cl.p++; adv = cl.p < cl.len && C.ent[cl.lo + cl.p] < site; } }
    const bool live = mine && cl.p < cl.len;
    const int hv = live ? C.ent[cl.lo + cl.p] : INT_MAX;
    L.hv = hv; L.value = mine ? (live ? hv : cl.last) : LANE_UNUSED;
}
// The pop at which the reference's loop ends after the site `s` has been looked at with hit cutoff c >= 1 (popSite's rule: a
// list runs out and fewer than c stay alive): INT_MAX if none (cannot happen: the very last entry always ends it).
__device__ __forceinline__ int cycleExitSite(const U &u, CycleLanes &cl, int n, int s, int c) {
    const int firstDeath = wmin((u.lane < n && cl.last >= s) ? cl.last : INT_MAX);
    if (n - c < 0) return firstDeath;
    if (n - c == 0) return max(firstDeath, wmin(u.lane < n ? cl.last : INT_MAX));    // every list has to be alive: the smallest last entry
    if (rl(cl.rank, 0) == -2) {                                // rank of every list's last entry (ascending, ties by lane)
        int rk = 0;
        for (int i = 0; i < n; i++) { const int li = rl(cl.last, i); rk += (li < cl.last || (li == cl.last && i < u.lane)) ? 1 : 0; }
        cl.rank = u.lane < n ? rk : -1;
    }
    int dstar = INT_MIN;
    { const u64 K = __ballot(cl.rank == n - c); if (K) dstar = rl(cl.last, __builtin_ctzll(K)); }
    return max(firstDeath, dstar);
}

// BBIndex.findMaxQscore2 in the whole-cycle form.  Returns false when the cycle was declined (nothing touched).
template <int WLEN> __device__ __forceinline__ bool findMaxQscore2Cycle(U &u, CycleLds &C, WaveLds<WLEN> &S, WL &L, int baseChrom, int prevMaxHits, int numKeys, int mqsAllKeys, int &outQ, int &outHits) {
    const bbidx_params &p = u.ix->p;
    const int numHits = L.n, lane = u.lane;
    if (wmin(lane < numHits ? L.ksc : INT_MAX) <= 0) return false;       // an isolated site must not be able to reach maxQuickScore
    if (max(p.maxIndel, p.maxIndel2) > 32768) return false;               // the presence map's buckets must span the hit window
    CycleLanes cl; int E;
    const int ncand = cycleLoad(u, C, S, L, baseChrom, cl, E);
    CST(u, 0, 1);
    if (ncand < 0) { CST(u, 1, 1); return false; }
    CST(u, 2, E); CST(u, 3, ncand);
    const int *cs = S.xch[0];
    WORK(u.cPrescan, (unsigned)E);
    const int mqs = numHits == numKeys ? mqsAllKeys : maxQuickScoreW(u, L.offs, L.ksc, numHits);
    int topQscore = -999999999, maxHits = 0;
    int approxHitsCutoff = max(prevMaxHits, min(p.minApproxHitsToKeep, numHits - 1));
    const int lo = min(p.maxIndel, p.maxIndel2), hi = p.maxIndel2;
    int prev = INT_MIN;
    // at a cutoff >= 2 the pops in front of the first candidate are isolated and the loop may already end among them
    bool ended = approxHitsCutoff >= 2 && cycleExitSite(u, cl, numHits, INT_MIN, approxHitsCutoff) < cycleNext(u, cs, ncand, INT_MIN);
    // the best quick score any isolated entry of the cycle has: once the running maximum is there, folding more of them is a no-op
    int isoBest = 0;
    if (!ended && approxHitsCutoff <= 1) {
        for (int base = 0; base < E; base += 64) { const int e = base + lane; if (e < E) isoBest = max(isoBest, (int)C.isoq[e]); }
        isoBest = wmax(isoBest);
    }
    while (!ended) {
        approxHitsCutoff = uni(approxHitsCutoff); topQscore = uni(topQscore); maxHits = uni(maxHits); prev = uni(prev);
        const int site = cycleNext(u, cs, ncand, prev);
        if (approxHitsCutoff <= 1 && isoBest > topQscore) {
            // isolated entries below this candidate (all remaining ones after the last candidate): each is a site with one hit
            int m = 0;
            for (int base = 0; base < E; base += 64) {
                const int e = base + lane;
                if (e < E) { const int v = C.ent[e]; if (v < site && v > prev) m = max(m, (int)C.isoq[e]); }
            }
            m = wmax(m);
            if (m > 0 && m > topQscore) { maxHits = max(maxHits, 1); topQscore = m; }
        }
        if (site == INT_MAX) break;
        CST(u, 4, 1);
        cycleSeek(u, C, cl, L, site);
        const int approxHits = popc(mask_ule((unsigned)L.value - (unsigned)(site - lo), (unsigned)(lo + hi)));
        if (approxHits >= approxHitsCutoff) {
            const int centerIndex = __builtin_ctzll(mask_eq(L.hv, site));
            const int qscore = quickScoreW(u, L.value, L.ksc, L.offs, centerIndex, site, approxHits, numHits)
                             + scoreZ2W(u, L.value, L.offs, site, approxHits, numHits);
            if (qscore > topQscore) {
                maxHits = max(approxHits, maxHits);
                approxHitsCutoff = max(approxHitsCutoff, approxHits - 1);
                topQscore = qscore;
                if (qscore >= mqs) break;
            }
        }
        if (approxHitsCutoff >= 2) {
            // the loop ends at pop X; isolated pops before it change nothing at this cutoff
            const int X = cycleExitSite(u, cl, numHits, site, approxHitsCutoff);
            if (X < cycleNext(u, cs, ncand, site) || X == site) break;
        }
        prev = site;
    }
    outQ = topQscore; outHits = maxHits;
    return true;
}

// BBIndex.findMaxQscore2 :2294-2450
template <bool LONG> __device__ __forceinline__ void findMaxQscore2W(U &u, WL &L, int baseChrom, int prevMaxHits, bool perfectOnly, int numKeys, int mqsAllKeys,
                                int &outQ, int &outHits) {
    const bbidx_params &p = u.ix->p;
    const int numHits = L.n;
    // maxQuickScore over the lists that have hits; when every key has one that is the read's own value
    const int mqs = numHits == numKeys ? mqsAllKeys : maxQuickScoreW(u, L.offs, L.ksc, numHits);
    int topQscore = -999999999, maxHits = 0, approxHitsCutoff, indelCutoff;
    if (perfectOnly) { approxHitsCutoff = numHits; indelCutoff = 0; }
    else { approxHitsCutoff = max(prevMaxHits, min(p.minApproxHitsToKeep, numHits - 1)); indelCutoff = p.maxIndel2; }
    while (L.nlive > 0) {
        approxHitsCutoff = uni(approxHitsCutoff); topQscore = uni(topQscore); maxHits = uni(maxHits); L.nlive = uni(L.nlive);
        if (LONG && numHits >= 2 && L.bwait == 0) {
            { const int np_ = batchPop(u, L, min(p.maxIndel, indelCutoff), p.maxIndel2, approxHitsCutoff, true, topQscore, maxHits, mqs, baseChrom, u.cPrescan);
              BST(u, 0, 1); BST(u, 1, np_ > 0); BST(u, 2, np_); if (np_ > 0) continue; }
            L.bwait = BATCH_RETRY;
        } else if (LONG && L.bwait > 0) L.bwait--;
        const int site = wmin(L.hv);
        const int minsite = site - min(p.maxIndel, indelCutoff), maxsite = site + p.maxIndel2;
        const int approxHits = popc(mask_ule((unsigned)L.value - (unsigned)minsite, (unsigned)(maxsite - minsite)));
        if (approxHits >= approxHitsCutoff) {
            const int centerIndex = __builtin_ctzll(mask_eq(L.hv, site));
            const int qscore = quickScoreW(u, L.value, L.ksc, L.offs, centerIndex, site, approxHits, numHits)
                             + scoreZ2W(u, L.value, L.offs, site, approxHits, numHits);
            if (qscore > topQscore) {
                maxHits = max(approxHits, maxHits);
                approxHitsCutoff = max(approxHitsCutoff, approxHits - 1);
                topQscore = qscore;
                if (qscore >= mqs) break;
            }
        }
        else if (LONG && approxHitsCutoff >= 2 && L.bulk == 0) { bulkSkip(u, L, site, min(p.maxIndel, indelCutoff), approxHitsCutoff, baseChrom, u.cPrescan); continue; }
        if (LONG && L.bulk > 0) L.bulk--;
        BST(u, 3, 1);
        popSite(u, L, site, approxHitsCutoff, perfectOnly, baseChrom, u.cPrescan);
    }
    outQ = topQscore; outHits = maxHits;
}

// MultiStateAligner11tsJNI.calcAffineScore(locArray, baseScores, bases, minContig) :871-1027 over the LDS location
// array, 64 bases per step.  Sequential state of the reference and how it is recovered:
//   lastValue  = the previous element                       -> loc[p-1]
//   lastLoc    = the last positive element before p         -> highest set bit of the "positive" ballot below p
//   timeInMode = length of the run of -1 ending at p        -> distance to the highest "not -1" bit below p
//   contig     = equal-to-previous streak                   -> popcount of "equal" events since the last reset event
template <int WLEN> __device__ __forceinline__ int calcAffineScoreW(const U &u, const WaveLds<WLEN> &S, int strand, int minContig) {
    const int blen = u.blen, lane = u.lane;
    int score = 0, carryLastLoc = -3, carryRun = 0, carryContig = 0, maxContig = 0;
    for (int base = 0; base < blen; base += 64) {
        const int p = base + lane;
        const bool valid = p < blen;
        const int loc = valid ? S.loc[p] : 0;
        const int prev = (valid && p > 0) ? S.loc[p - 1] : -1;
        const bool pos = valid && loc > 0, neg1 = valid && loc == -1;
        const u64 posM = __ballot(pos), n1M = __ballot(neg1);
        const u64 lt = lt_mask(lane);
        const u64 mlo = posM & lt;
        const int lastLoc = mlo ? S.loc[base + hibit(mlo)] : carryLastLoc;
        int c = 0, ev = 0;                                   // ev: 1 equal, 2 restart, 3 indel
        if (pos) {
            const int bs = S.bsc[strand ? blen - 1 - p : p];
            if (loc == prev) { c = 100 + bs; ev = 1; }
            else if (loc == lastLoc || lastLoc < 0) { c = 70 + bs; ev = 2; }
            else if (loc < lastLoc) { c = 70 + bs + calcDelScoreApprox(lastLoc - loc + 1); ev = 3; }
            else { c = 70 + bs + insCum(min(loc - lastLoc, 5)); ev = 3; }
        } else if (neg1) {
            const u64 nb = ~n1M & lt;
            const int t = nb ? lane - hibit(nb) : lane + 1 + carryRun;
            c = subArr(t);
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

// BBIndex.extendScore :2558-2833
template <int WLEN> __device__ __forceinline__ int extendScoreW(U &u, WaveLds<WLEN> &S, int strand, int value, int offs, int numHits, int chrom, int centerIndex) {
    const bbidx_params &p = u.ix->p;
    const int blen = u.blen, lane = u.lane, k = u.k;
    const int centerVal = rl(value, centerIndex), centerLoc = u.c.siteOf(centerVal);
    const int minVal = centerVal - p.maxIndel, maxVal = centerVal + p.maxIndel2;
    const uint8_t *ref = u.ix->chromArr[chrom];
    const int reflen = u.ix->chromArrLen[chrom];
    const uint8_t *rb = S.base[strand];
    WORK(u.cExtend, 1u);
    for (int i = lane; i < blen; i += 64) S.loc[i] = -1;
    wsync();
    const u64 R = __ballot(lane < numHits && value >= minVal && value <= maxVal);
    // backward from each key's last base; the first key in range runs through mismatches, the others stop at the first
    int keynum = 0;
    for (u64 m = R; m; m &= m - 1) {
        const int i = __builtin_ctzll(m);
        const int refbase = u.c.siteOf(rl(value, i)), c0 = rl(offs, i) + k - 1;
        keynum++;
        if (c0 < 0 || refbase + c0 >= reflen) continue;
        if (keynum == 1) {
            for (int base = 0; base <= c0; base += 64) {
                const int q = base + lane;
                if (q <= c0 && rb[q] == ref[refbase + q]) S.loc[q] = refbase;
            }
            WORK(u.cRefBytes, (unsigned)(c0 + 1));
        } else {
            for (int top = c0; top >= 0; top -= 64) {
                const int q = top - lane;
                const bool valid = q >= 0;
                const int old = valid ? S.loc[q] : 0;
                const u64 Em = __ballot(valid && old == refbase);
                if (Em & 1) break;                                              // already holds this site: nothing compared
                const bool mm = valid && rb[q] != ref[refbase + q];
                const u64 stopM = Em | __ballot(mm);
                const int s = stopM ? __builtin_ctzll(stopM) : 64;
                if (valid && lane < s && (old < 0 || refbase == centerLoc)) S.loc[q] = refbase;
                WORK(u.cRefBytes, (unsigned)(min(s, min(64, top + 1)) + ((s < 64 && !((Em >> s) & 1)) ? 1 : 0)));
                if (s < 64) break;
            }
        }
        wsync();
    }
    // forward from the base after each key: runs through mismatches over unassigned bases, stops on an assigned base
    // once a mismatch has been seen
    for (u64 m = R; m; m &= m - 1) {
        const int i = __builtin_ctzll(m);
        const int refbase = u.c.siteOf(rl(value, i));
        bool mmprev = false;
        for (int c = rl(offs, i) + k; c < blen; c += 64) {
            const int q = c + lane;
            const bool valid = q < blen && refbase + q < reflen;
            const int old = valid ? S.loc[q] : -1;
            const bool A = valid && old >= 0;
            const u64 Em = __ballot(valid && old == refbase);
            if (Em & 1) break;
            const bool mm = valid && rb[q] != ref[refbase + q];
            const u64 mmM = __ballot(mm), validM = __ballot(valid), Am = __ballot(A);
            const bool mmBefore = mmprev || (mmM & lt_mask(lane)) != 0;
            const bool stop = valid && (old == refbase || (A && (mmBefore || mm)));
            const u64 stopM = __ballot(stop) | ~validM;
            const int s = stopM ? __builtin_ctzll(stopM) : 64;
            if (valid && lane < s && !mm && (old < 0 || refbase == centerLoc)) S.loc[q] = refbase;
            unsigned cnt = (unsigned)s;
            if (s < 64 && ((validM >> s) & 1)) {
                const bool Es = (Em >> s) & 1, As = (Am >> s) & 1;
                const bool mmBeforeS = mmprev || (mmM & lt_mask(s)) != 0;
                if (!Es && !(mmBeforeS && As)) cnt++;
            }
            WORK(u.cRefBytes, cnt);
            if (s < 64) break;
            mmprev = mmprev || mmM != 0;
        }
        wsync();
    }
    for (int i = lane; i < blen; i += 64) if (rb[i] == 'N') S.loc[i] = -2;
    wsync();
    return uni(calcAffineScoreW(u, S, strand, p.kfilter));
}

// An upper bound of extendScore for a site with at most three hit keys in range (up to three diagonals), from one pass over the
// read.  extendScore only ever assigns a base to a diagonal on which it matches the reference, so with m(q) = "base q matches
// on one of the diagonals": calcAffineScore gives an assigned base at most 100 + baseScore, an unassigned one at most -25
// (POINTS_SUB_ARRAY's mildest entry) and an N exactly 0.  slowWalk3 uses a site's score only through `score >= cutoff` and
// `score == maxScore`: when the bound is below both, the site cannot do anything and its extension is not computed.  The reads
// that need this are the ones with few hit keys (many substitutions): their hit cutoff is 1 or 2, so every chance hit of a
// single key is a site -- hundreds per read -- and nearly all of them fail here.  Returns INT_MAX when it does not apply.
#ifndef BBIDX_EXTEND_BOUND
#define BBIDX_EXTEND_BOUND 1
#endif
#ifndef BBIDX_ALL_LISTS
#define BBIDX_ALL_LISTS 1           // 0: no shortcut for walks in which every list has to hit
#endif
#ifndef BBIDX_CYCLE_LOW
#define BBIDX_CYCLE_LOW 1           // 0: walks that start at a hit cutoff of 1 take the sequential heap walk
#endif
template <int WLEN> __device__ __forceinline__ int extendBoundW(U &u, const WaveLds<WLEN> &S, int strand, int value, int numHits, int chrom, int centerIndex) {
    const bbidx_params &p = u.ix->p;
    const int blen = u.blen, lane = u.lane;
    const int centerVal = rl(value, centerIndex);
    const int minVal = centerVal - p.maxIndel, maxVal = centerVal + p.maxIndel2;
    const u64 R = __ballot(lane < numHits && value >= minVal && value <= maxVal);
    if (popc(R) > 3) return INT_MAX;
    int d0 = -1, d1 = -1, d2 = -1, nd = 0;
    for (u64 m = R; m; m &= m - 1) {
        const int rbse = u.c.siteOf(rl(value, __builtin_ctzll(m)));
        if (nd == 0) { d0 = rbse; nd = 1; }
        else if (rbse != d0 && nd == 1) { d1 = rbse; nd = 2; }
        else if (rbse != d0 && rbse != d1 && nd == 2) { d2 = rbse; nd = 3; }
    }
    const uint8_t *ref = u.ix->chromArr[chrom];
    const int reflen = u.ix->chromArrLen[chrom];
    const uint8_t *rb = S.base[strand];
    int total = 0;
    for (int base = 0; base < blen; base += 64) {
        const int q = base + lane;
        int c = 0;
        if (q < blen) {
            const int b = rb[q];
            bool m = d0 + q < reflen && b == ref[d0 + q];
            if (nd > 1) m |= d1 + q < reflen && b == ref[d1 + q];
            if (nd > 2) m |= d2 + q < reflen && b == ref[d2 + q];
            c = b == 'N' ? 0 : (m ? max(100 + (int)S.bsc[strand ? blen - 1 - q : q], -25) : -25);
        }
        total += c;
    }
    WORK(u.cRefBytes, (unsigned)(blen * nd));
    return wsum(total);
}

// extendScore of a site hit by ONE key, one site per lane.  With a single key in range the first-key rule applies to it in both
// directions -- backward and forward it runs through mismatches and assigns every base that matches on its diagonal (:2598-2700)
// -- so the location array is the match mask of that diagonal, and calcAffineScore over it is a run-length sum: an assigned base
// scores 100 + baseScore after an assigned one and 70 + baseScore otherwise, the t-th base of a run of unassigned ones
// POINTS_SUB_ARRAY[t], an N nothing (and it ends either run).  The caller keeps away the sites this does not hold for: refbase 0
// (calcAffineScore tests loc > 0), a window that reaches the end of the chromosome array, kfilter > 1.
template <int WLEN> __device__ __forceinline__ int singleKeyScoreLane(const U &u, const WaveLds<WLEN> &S, int strand, const uint8_t *ref, int refbase, bool active) {
    const int blen = u.blen;
    const uint8_t *rb = S.base[strand];
    const int sh = (int)((unsigned long long)(ref + refbase) & 3ull);
    const unsigned *rw = (const unsigned *)(ref + refbase - sh);
    const int nw = (sh + blen + 3) >> 2, nwMax = (blen + 6) >> 2;
    int score = 0, run = 0; bool prevA = false;
    unsigned w = active ? rw[0] : 0u;
    for (int j = 0; j < nwMax; j++) {
        const unsigned nxt = (active && j + 1 < nw) ? rw[j + 1] : 0u;    // the next word is on its way while this one is used
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int q = 4 * j + t - sh;
            if (active && q >= 0 && q < blen) {
                const int b = rb[q], r = (int)((w >> (8 * t)) & 255u);
                if (b == 'N') { run = 0; prevA = false; }
                else if (b == r) { score += (prevA ? 100 : 70) + (int)S.bsc[strand ? blen - 1 - q : q]; prevA = true; run = 0; }
                else { run++; score += subArr(run); prevA = false; }
            }
        }
        w = nxt;
    }
    return score;
}

// BBIndex.makeGapArray :2837-2878 -- rare (a site spanning more than MINGAP + read length); one lane walks LDS
template <int WLEN> __device__ __forceinline__ int makeGapArrayW(const U &u, WaveLds<WLEN> &S, int minLoc, int minGap) {
    if (u.lane == 0) {
        int *locArray = S.loc;
        const int n = u.blen;
        int gaps = 0; bool doSort = false;
        if (locArray[0] < 0) locArray[0] = minLoc;
        for (int i = 1; i < n; i++) {
            if (locArray[i] < 0) locArray[i] = locArray[i - 1] + 1; else locArray[i] += i;
            if (locArray[i] < locArray[i - 1]) doSort = true;
        }
        if (doSort) {
            for (int i = 1; i < n; i++) { const int v = locArray[i]; int j = i - 1; while (j >= 0 && locArray[j] > v) { locArray[j + 1] = locArray[j]; j--; } locArray[j + 1] = v; }
        }
        for (int i = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) gaps++;
        int len = 0;
        if (gaps >= 1) {
            len = 2 + gaps * 2;
            if (len > BBIDX_MAX_GAPS) len = -1;
            else {
                S.gaps[0] = locArray[0]; S.gaps[len - 1] = locArray[n - 1];
                for (int i = 1, j = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) { S.gaps[j] = locArray[i - 1]; S.gaps[j + 1] = locArray[i]; j += 2; }
            }
        }
        S.ngaps = len;
    }
    wsync();
    return __builtin_amdgcn_readfirstlane(S.ngaps);
}

// SiteScore.setPerfect (current/stream/SiteScore.java:239-292): order-independent form
template <int WLEN> __device__ __forceinline__ void setPerfectW(const U &u, const WaveLds<WLEN> &S, int chrom, int strand, int start, int stop, int &perfectOut, int &semiOut) {
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
    const int total = uni(mx - refloc + 1);                 // bases compared; lanes past the end re-read the last one
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

// BBIndex.slowWalk3 :1219-1706
template <bool LONG, int WLEN> __device__ __forceinline__ void slowWalk3W(U &u, WaveLds<WLEN> &S, CycleLds *C, WL &L, int strand, int numKeys, int mqs, int baseChrom_,
                           SiteOut &ssl, int *bestScores, bool allBasesCovered, int maxScore, bool fullyDefined) {
    const bbidx_params &p = u.ix->p;
    const int blen = u.blen, lane = u.lane;
    // maxQuickScore(offsets, keyScores) of this strand: the minus-strand arrays are the mirrored plus-strand ones, whose sum,
    // coverage and span are the same, so the caller's value serves both strands
    const int baseChrom = u.c.baseChrom(baseChrom_);
    const int numHits = L.n;
    const bool filter_by_qscore = numKeys >= 5;
    const int minScore = (int)(MIN_SCORE_MULT * maxScore);
    const int minQuickScore = (int)(MIN_QSCORE_MULT * mqs);
    int currentTopScore = bestScores[0];
    int cutoff = max(minScore, (int)(currentTopScore * DYN_SCORE));
    int qcutoff = max(bestScores[2], minQuickScore);
    int bestqscore = bestScores[3], maxHits = bestScores[1], perfectsFound = bestScores[5];
    int approxHitsCutoff = calcApproxHitsCutoff(p, numKeys, maxHits, p.minApproxHitsToKeep, currentTopScore >= maxScore);
    if (approxHitsCutoff > numHits) return;
    const bool shortCircuit = allBasesCovered && numKeys == numHits && filter_by_qscore;
    if (currentTopScore >= maxScore) qcutoff = max(qcutoff, (int)(mqs * DYN_QSCORE_PERFECT));

    PrevSite pv; pv.idx = -1; pv.chrom = pv.strand = pv.start = pv.stop = pv.score = pv.perfect = pv.semiperfect = pv.ngaps = 0;
    bool finished = false;
    // loop-carried uniform state, re-declared uniform at the top of every round (see wavep::uni)
    auto reuni = [&]() {
        approxHitsCutoff = uni(approxHitsCutoff); cutoff = uni(cutoff); qcutoff = uni(qcutoff); currentTopScore = uni(currentTopScore);
        maxHits = uni(maxHits); perfectsFound = uni(perfectsFound); bestqscore = uni(bestqscore); L.nlive = uni(L.nlive);
        pv.idx = uni(pv.idx); pv.chrom = uni(pv.chrom); pv.strand = uni(pv.strand); pv.start = uni(pv.start); pv.stop = uni(pv.stop);
        pv.score = uni(pv.score); pv.perfect = uni(pv.perfect); pv.semiperfect = uni(pv.semiperfect); pv.ngaps = uni(pv.ngaps);
        ssl.n = uni(ssl.n); ssl.overflow = uni(ssl.overflow); finished = uni(finished); u.cWalk = uni(u.cWalk); u.cExtend = uni(u.cExtend); u.cRefBytes = uni(u.cRefBytes);
    };
    // one site of the merged lists with every list's head / value in L (the reference's loop body between two polls)
    bool seqMode = false;                                      // (phase timers: the sequential walk is accounted separately)
    auto visit = [&](const int site) -> int {
        const int minsite = site - p.maxIndel, maxsite = site + p.maxIndel2;
        const bool inr = (unsigned)L.value - (unsigned)minsite <= (unsigned)(maxsite - minsite);
        const int approxHits = popc(mask_ule((unsigned)L.value - (unsigned)minsite, (unsigned)(maxsite - minsite)));
        PH_WALK(u, seqMode ? 4 : 1);
        if (approxHits >= approxHitsCutoff) {
            const int centerIndex = __builtin_ctzll(mask_eq(L.hv, site));
            const int maxNearbySite = wmax(inr ? L.value : site);
            int score;
            int qscore = filter_by_qscore ? quickScoreW(u, L.value, L.ksc, L.offs, centerIndex, site, approxHits, numHits) : qcutoff;
            qscore += scoreZ2W(u, L.value, L.offs, site, approxHits, numHits);
            int mapStart = site, mapStop = maxNearbySite;
            bool locArrayValid = false;
            if (qscore < qcutoff) score = -1;
            else {
                const int chrom = u.c.chromOf(site, baseChrom);
                if (shortCircuit && qscore == mqs) score = maxScore;
                else {
                    PH_MARK(u, 3); PH_WALK(u, seqMode ? 4 : 2);
                    bool hopeless = false;
                    if (BBIDX_EXTEND_BOUND && approxHits <= 3) {
                        const int ub = uni(extendBoundW(u, S, strand, L.value, numHits, chrom, centerIndex));
                        hopeless = ub < cutoff && ub < maxScore;
                    }
                    if (hopeless) { score = -1; WORK(u.cExtend, 1u); }
                    else {
                        score = extendScoreW(u, S, strand, L.value, L.offs, numHits, chrom, centerIndex);
                        locArrayValid = true;
                        int mn = INT_MAX, mx = INT_MIN;
                        for (int i = lane; i < blen; i += 64) { const int x = S.loc[i]; if (x > -1) { mn = min(mn, x); mx = max(mx, x); } }
                        mn = wmin(mn); mx = wmax(mx);
                        if (mn < 0 || mx < 0) score = -99999;
                        mapStart = u.c.toNumber(mn, chrom);
                        mapStop = u.c.toNumber(mx, chrom);
                    }
                    PH_MARK(u, 4); PH_SKIPW(u);
                }
                if (score == maxScore) {
                    qcutoff = max(qcutoff, (int)(mqs * DYN_QSCORE_PERFECT));
                    approxHitsCutoff = calcApproxHitsCutoff(p, numKeys, maxHits, p.minApproxHitsToKeep, true);
                }
                if (score >= cutoff) { qcutoff = max(qcutoff, (int)(qscore * DYN_QSCORE)); bestqscore = max(qscore, bestqscore); }
            }
            PH_WALK(u, seqMode ? 4 : 2);
            if (score >= cutoff) {
                if (score > currentTopScore) {
                    maxHits = max(approxHits, maxHits);
                    approxHitsCutoff = calcApproxHitsCutoff(p, numKeys, maxHits, approxHitsCutoff, currentTopScore >= maxScore);
                    cutoff = max(cutoff, (int)(score * DYN_SCORE));
                    if (score >= maxScore) cutoff = max(cutoff, (int)(score * 0.95f));
                    currentTopScore = score;
                }
                const int chrom = u.c.chromOf(mapStart, baseChrom);
                const int site2 = u.c.siteOf(mapStart);
                const int site3 = u.c.siteOf(mapStop) + blen - 1;
                int ngaps = 0;
                if (site3 - site2 >= MINGAP + blen && locArrayValid) {
                    ngaps = makeGapArrayW(u, S, site2, MINGAP);
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
                            setPerfectW(u, S, pv.chrom, pv.strand, pv.start, pv.stop, pv.perfect, pv.semiperfect);
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
                            setPerfectW(u, S, pv.chrom, pv.strand, pv.start, pv.stop, pv.perfect, pv.semiperfect);
                        }
                        pv.score = betterScore;
                        wb = 3;
                    } else makeNew = true;
                    // the merged site goes back to the list after the if-chain: a lane-0 store inside an arm would share its
                    // join block with the chain, and every value merged there would count as divergent
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
                    if (!perfect1) setPerfectW(u, S, chrom, strand, site2, site3, sp, ssemi);
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
            PH_WALK(u, seqMode ? 4 : 2);
        }
        return approxHits;
    };
    bool cycled = false;
    // A walk in which EVERY list has to hit (after a perfect site the hit cutoff is the number of lists; in a read with a perfect
    // site in an early cycle that is every later cycle, none of which the prescan has seen): a site needs an entry of the
    // shortest list with an entry of every other list no farther from it than MAX_INDEL + MAX_INDEL2.  If no entry of the
    // shortest list has that, the cycle holds no site and the walk has nothing to do -- found from the gathered entries alone,
    // without presence maps, candidates or a single visit.
    if (LONG && BBIDX_CYCLE && BBIDX_ALL_LISTS && C != nullptr && approxHitsCutoff >= numHits && numHits >= 2 && max(p.maxIndel, p.maxIndel2) <= 32768) {
        CycleLanes cl; int E;
        PH_WALK(u, 0);
        if (cycleGather<false>(u, *C, S, L, baseChrom, cl, E)) {
            const int mylen = lane < numHits ? cl.len : INT_MAX;
            const int minLen = wmin(mylen);
            if (minLen <= 8) {
                const int lo0 = rl(cl.lo, __builtin_ctzll(__ballot(mylen == minLen)));
                const int W = p.maxIndel + p.maxIndel2;
                bool possible = false;
                for (int t = 0; t < minLen && !possible; t++) {
                    const int w = C->ent[lo0 + t];
                    int a = 0, b = lane < numHits ? cl.len : 0;                  // first entry of this lane's list >= w - W
                    while (__ballot(b > a)) {
                        if (b > a) { const int m = (a + b) >> 1; if (C->ent[cl.lo + m] < w - W) a = m + 1; else b = m; }
                    }
                    const bool ok = lane >= numHits || (a < cl.len && C->ent[cl.lo + a] <= w + W);
                    possible = __ballot(ok) == ~0ull;
                }
                if (!possible) { cycled = true; WORK(u.cWalk, (unsigned)E); }
            }
        }
        PH_WALK(u, 0);
    }
    // A walk that starts at a hit cutoff of 1 (reads with few hit keys) looks at EVERY entry: an isolated one is a site with one
    // hit, whose quick score is the fixed keyScore + scoreZ1Key (isoq) and whose extendScore follows from its diagonal alone.
    const bool lowCutoff = approxHitsCutoff <= 1;
    const bool lowOk = BBIDX_CYCLE_LOW && filter_by_qscore && p.kfilter <= 1 && wmin(lane < numHits ? L.ksc : INT_MAX) > 0 &&
                       wmax(lane < numHits ? L.ksc : 0) + u.scoreZ1Key < 65535;
    if (!cycled && LONG && BBIDX_CYCLE && C != nullptr && (!lowCutoff || lowOk) && max(p.maxIndel, p.maxIndel2) <= 32768) {
        // whole-cycle form (see findMaxQscore2Cycle): at a hit cutoff >= 2 an isolated entry can neither be scored nor change
        // any state, so only the candidates are visited, in site order, each followed by popSite's exit rule
        CycleLanes cl; int E;
        PH_WALK(u, 0);
        const int ncand = cycleLoad(u, *C, S, L, baseChrom, cl, E);
        PH_WALK(u, 0);
        int npass = 0;
        int *passv = S.xch[1];                                 // the load phase's scratch rows are free again (cs[] is S.xch[0])
        if (ncand >= 0 && lowCutoff) {
            // The isolated entries that pass the quick-score filter are scored exactly, one per lane; the few that reach the score
            // cutoff (or a case singleKeyScoreLane does not cover) join the candidates and are visited in site order like them.
            // The thresholds only rise during the walk, so the ones that fail here would fail at their turn too.
            for (int base = 0; base < E && npass <= 64; base += 64) {
                const int e = base + lane;
                const int iq = e < E ? (int)C->isoq[e] : 0;
                const bool need = iq != 0 && iq >= qcutoff;
                if (!__ballot(need)) continue;
                const int v = need ? C->ent[e] : 0;
                const int chrom = u.c.chromOf(v, baseChrom), refbase = u.c.siteOf(v);
                const int reflen = need ? u.ix->chromArrLen[chrom] : 0;
                const bool plain = need && refbase > 0 && refbase + blen + 4 < reflen;
                const uint8_t *ref = plain ? u.ix->chromArr[chrom] : nullptr;
                const int sc = singleKeyScoreLane(u, S, strand, ref, refbase, plain);
                const bool pass = need && (!plain || sc >= cutoff || sc == maxScore);
                WORK(u.cExtend, (unsigned)popc(__ballot(plain && !pass)));
                WORK(u.cRefBytes, (unsigned)(blen * popc(__ballot(plain))));
                const u64 PM = __ballot(pass);
                if (pass) { const int slot = npass + popc(PM & lt_mask(lane)); if (slot < 64) passv[slot] = v; }
                npass += popc(PM);
            }
            wsync();
            PH_WALK(u, 3);
        }
        if (ncand >= 0 && npass <= 64) {
            cycled = true;
            WORK(u.cWalk, (unsigned)E);
            // the candidate sites sit in S.xch, which nothing inside visit() touches (compaction and the greedy trim are over)
            const int *cs = S.xch[0];
            int prev = INT_MIN;
            for (;;) {
                reuni(); prev = uni(prev);
                const int nextC = cycleNext(u, cs, ncand, prev);
                const int site = approxHitsCutoff <= 1 ? min(nextC, cycleNext(u, passv, npass, prev)) : nextC;
                // at a cutoff >= 2 the pops between two candidates are isolated, and the loop may end among them (or at `prev`)
                if (approxHitsCutoff >= 2) {
                    const int X = cycleExitSite(u, cl, numHits, prev, approxHitsCutoff);
                    if (X < site || (X == prev && prev != INT_MIN)) break;
                }
                if (site == INT_MAX) break;
                cycleSeek(u, *C, cl, L, site);
                visit(site);
                if (uni(finished)) break;
                prev = site;
            }
        }
    }
    PH_WALK(u, 1);
    seqMode = !cycled;
    if (LONG && BBIDX_CYCLE && !cycled) refillLists(L);          // makeListsW left the look-ahead buffers to this path
    while (!cycled && L.nlive > 0 && !finished) {
        reuni();
        if (LONG && approxHitsCutoff >= 2 && L.bwait == 0) {
            int unusedQ = 0, unusedH = 0;
            if (batchPop(u, L, p.maxIndel, p.maxIndel2, approxHitsCutoff, false, unusedQ, unusedH, 0, baseChrom, u.cWalk) > 0) continue;
            L.bwait = BATCH_RETRY;
        } else if (LONG && L.bwait > 0) L.bwait--;
        const int site = wmin(L.hv);
        const int approxHits = visit(site);
        if (uni(finished)) break;
        if (LONG && approxHits < approxHitsCutoff && approxHitsCutoff >= 2 && L.bulk == 0) { bulkSkip(u, L, site, p.maxIndel, approxHitsCutoff, baseChrom, u.cWalk); continue; }
        if (LONG && L.bulk > 0) L.bulk--;
        BST(u, 4, 1);
        popSite(u, L, site, approxHitsCutoff, false, baseChrom, u.cWalk);
    }
    PH_WALK(u, seqMode ? 4 : 1);
    bestScores[0] = max(bestScores[0], currentTopScore);
    bestScores[1] = max(bestScores[1], maxHits);
    bestScores[2] = max(bestScores[2], qcutoff);
    bestScores[3] = max(bestScores[3], bestqscore);
    bestScores[4] = mqs;
    bestScores[5] = perfectsFound;
}

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

// BBIndex.trimExcessHitListsByGreedy :266-350 (+ Solver.findWorstGreedy :46-95): lane j evaluates list position j,
// the "first strict prefix minimum below the early-termination score" rule comes from an exclusive prefix-min scan.
// x = lengths[lane] (COUNTS of the lane's key), key = keys[lane]; both are updated in place.
template <int WLEN> __device__ __forceinline__ int trimByGreedyW(const U &u, WaveLds<WLEN> &S, int off, int ksc, int n, int maxHitLists, int &key, int &x) {
    const DevIndex &ix = *u.ix;
    const bbidx_params &p = ix.p;
    const int lane = u.lane;
    const float inv = __fdiv_rn(1.0f, (float)u.baseKeyHitScore);
    const int limit = max(SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch]) * n;
    const int limit2 = max(SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch2]);
    const int limit3 = max(SMALL_LIST, ix.lengthHistogram[p.maxShortestListToSearch]);
    const bool act = lane < n;
    if (!act) x = 0;
    int sum = wsum(x);
    const int initialHitCount = popc(__ballot(act && x != 0));
    const int shortest = wmin((act && x > 0) ? x : INT_MAX - 1);
    if (initialHitCount < p.minApproxHitsToKeep) return initialHitCount;
    if (shortest > limit3 && !p.slow) { key = -1; return 0; }
    // every list shorter than SMALL_LIST: whichever list the first round picks as worst, the loop returns there
    // (`lengths[worst] < SMALL_LIST`), so the values need not be computed
    if (wmax(x) < SMALL_LIST) return initialHitCount;
    int *listsL = S.xch[0], *offL = S.xch[1], *lenL = S.xch[2];
    if (act) offL[lane] = off;
    int hitsCount = initialHitCount;
    const long long EARLY = -50LL * 2000;
    while (hitsCount >= p.minApproxHitsToKeep && (sum > limit || sum / initialHitCount > limit2 || hitsCount > maxHitLists)) {
        const u64 M = __ballot(act && x > 0);
        if (act) { lenL[lane] = x; if (x > 0) listsL[popc(M & lt_mask(lane))] = lane; }
        wsync();
        long long v = LLONG_MAX;
        if (lane < hitsCount) v = valueOfElement(offL, n, lenL, __fmul_rn((float)ksc, inv), p.k, listsL, hitsCount, lane, p.pointsPerSite);
        long long pm = v;                                   // inclusive prefix minimum
        for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(pm, d); if (lane >= d) pm = min(pm, t); }
        long long ex = __shfl_up(pm, 1);
        if (lane == 0) ex = LLONG_MAX;
        const bool upd = lane < hitsCount && v < ex;
        const u64 earlyM = __ballot(upd && lane != 0 && ex < EARLY);
        const int worstIndex = earlyM ? __builtin_ctzll(earlyM) : hibit(__ballot(upd));
        // readlane, not ds_bpermute: the value steers uniform control flow, and only a readlane tells the compiler so
        const long long worstValue64 = (long long)(((u64)(unsigned)rl((int)(v >> 32), worstIndex) << 32) | (unsigned)rl((int)v, worstIndex));
        const int worstValue = worstValue64 < INT_MIN ? INT_MIN : (worstValue64 > INT_MAX ? INT_MAX : (int)worstValue64);
        const int worst = listsL[worstIndex];
        const int lenWorst = rl(x, worst);
        sum -= lenWorst;
        wsync();
        if (worstValue > 0 || lenWorst < SMALL_LIST) return hitsCount;
        hitsCount--;
        if (lane == worst) { x = 0; key = -1; }
    }
    return hitsCount;
}

// Compaction: lanes with keep==true move to lanes 0..count-1.  One LDS round trip publishes, for every destination
// lane, the lane it takes its values from; the values themselves then move with ds_bpermute (no LDS storage).
template <int WLEN> __device__ __forceinline__ int compactSrc(WaveLds<WLEN> &S, int lane, bool keep, int &count) {
    const u64 M = __ballot(keep);
    wsync();
    if (keep) S.xch[0][popc(M & lt_mask(lane))] = lane;
    wsync();
    const int src = S.xch[0][lane];
    wsync();
    count = popc(M);
    return lane < count ? src : lane;
}

// what BBIndex.getHits needs for one key on one strand, taken from the fused KeyEntry of the key (plus strand) or of
// the key it is the reverse complement of (minus strand)
struct KeyHit { int cnt, start, len, first; };

// the minus-strand view of lane i is the reverse-complement half of the record held by lane n-1-i
__device__ __forceinline__ KeyHit minusView(int lane, int n, int cntRC, int startR, int lenR, int firstR) {
    const int src = (lane < n) ? n - 1 - lane : lane;
    KeyHit h;
    h.cnt = __shfl(cntRC, src); h.start = __shfl(startR, src); h.len = __shfl(lenR, src); h.first = __shfl(firstR, src);
    return h;
}

// BBIndex.getHits (:354-391) + the heap fill at the top of slowWalk3/findMaxQscore2: builds the compacted lists
template <bool LONG, int WLEN> __device__ __forceinline__ int makeListsW(const U &u, WaveLds<WLEN> &S, WL &L, int block, int baseChrom, const KeyHit &h, int off, int ksc, int n, int minHits, bool refill = true) {
    const bool hit = u.lane < n && h.cnt > 0 && h.len > 0 && h.first != -1;
    const u64 M = __ballot(hit);
    const int nh = popc(M);
    if (nh < minHits) return nh;
    int cnt;
    const int src = compactSrc(S, u.lane, hit, cnt);
    L.n = L.nlive = nh; L.sites = (GlobalInts)u.ix->sites[block];
    const bool live = u.lane < nh;
    L.row = __shfl(h.start, src); L.stop = L.row + __shfl(h.len, src); L.offs = __shfl(off, src); L.ksc = __shfl(ksc, src);
    const int first = __shfl(h.first, src);
    L.value = live ? adjustSite(u, first, L.offs, baseChrom) : LANE_UNUSED;
    L.hv = live ? L.value : INT_MAX;
#pragma unroll
    for (int j = 0; j < NB; j++) L.nb[j] = 0;
    L.nbuf = 0;
    if (refill) refillLists(L);                                 // the whole-cycle walk gathers every entry itself
    L.bulk = L.bwait = -1;
    if (LONG) {
        const int entries = wsum(live ? L.stop - L.row : 0);
        L.bulk = entries >= BULK_MIN_ENTRIES ? 0 : -1;
        L.bwait = entries >= BATCH_MIN_ENTRIES ? 0 : -1;
    }
    return nh;
}

#ifndef BBIDX_WAVE_OCC
#define BBIDX_WAVE_OCC 6
#endif
#ifndef BBIDX_LONG_SHORT_OCC
#define BBIDX_LONG_SHORT_OCC 5      // long-list variant with short reads: the whole-cycle walk's LDS (7.7 KB per wave) allows 5 waves per SIMD
#endif
template <bool LONG, int WLEN> __global__ __launch_bounds__(64 * WAVES_PER_BLOCK, WLEN <= WSHORTLEN ? (LONG ? BBIDX_LONG_SHORT_OCC : 8) : (LONG ? 4 : BBIDX_WAVE_OCC)) void probe_wave_kernel(const Params P) {
    __shared__ WaveLds<WLEN> lds[WAVES_PER_BLOCK];
    __shared__ CycleLds cyc[LONG ? WAVES_PER_BLOCK : 1];   // whole-cycle walk: the long-list variant only
    __shared__ unsigned blockStats[5];
    __shared__ uint8_t compLut[256];      // AminoAcid.baseToComplementExtended
    __shared__ int8_t numLut[256];        // AminoAcid.baseToNumber
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < 5) blockStats[threadIdx.x] = 0;
    compLut[threadIdx.x] = (uint8_t)complement_extended((int)threadIdx.x);
    numLut[threadIdx.x] = (int8_t)(threadIdx.x < 128 ? base_num((int)threadIdx.x) : -1);
    __syncthreads();
    WaveLds<WLEN> &S = lds[wave];
    const DevIndex &ix = P.ix;
    const bbidx_params &p = ix.p;
    U u;
    u.ix = &ix;
    u.c.shift = 31 - p.chromBits; u.c.siteMask = (int)(0xFFFFFFFFu >> (p.chromBits + 1));
    u.c.cpb = 1 << p.chromBits; u.c.lowMask = u.c.cpb - 1; u.c.highMask = ~u.c.lowMask;
    u.k = p.k; u.baseKeyHitScore = BASE_HIT_SCORE * p.k;
    u.indelPenalty = (u.baseKeyHitScore / 2) - 1; u.indelPenaltyMult = 20;
    u.maxPenalty = u.baseKeyHitScore - (1 + u.baseKeyHitScore / 8);
    u.scoreZ1Key = Z_MULT * p.k;
    u.lane = lane; u.blen = 0;
    u.cPrescan = u.cWalk = u.cExtend = u.cRefBytes = 0;
#if defined(BBIDX_PHASE_TIMERS) || defined(BBIDX_BATCH_STATS) || defined(BBIDX_CYC_STATS)
    for (int j = 0; j < 5; j++) u.ph[j] = 0;
    u.phT = __builtin_readcyclecounter();
#endif
    unsigned cSites = 0;

    // persistent waves: a read costs anything from a few microseconds to milliseconds (repeats), and a wave that is done
    // would otherwise idle until the slowest of its block's four reads finishes
    // (The plain variant takes eight reads per pull: same-address atomics come back one every few nanoseconds, which is the whole
    // kernel on a small genome.  With long lists a read costs 0.1 - 10 ms of wave time: one per pull.)
    unsigned rNext = 0, rEnd = 0;
    for (;;) {
    long long r = 0;
    if (LONG) {
        if (lane == 0) r = (long long)atomicAdd(&P.queue[2], 1u);
        r = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)r);
        if (r >= P.nreads) break;
    } else {
        if (rNext >= rEnd) {
            unsigned r0 = 0;
            if (lane == 0) r0 = atomicAdd(&P.queue[2], 8u);
            rNext = (unsigned)__builtin_amdgcn_readfirstlane((int)r0);
            if ((long long)rNext >= P.nreads) break;
            rEnd = (long long)rNext + 8 < P.nreads ? rNext + 8u : (unsigned)P.nreads;
        }
        r = (long long)rNext++;
    }
    int result = 0;                      // what goes to nsites[r]
    bool done = false;
    bbidx_read rr; rr.len = 0; rr.nkeys = 0; rr.bases_off = 0; rr.keys_off = 0;
    if (!done) rr = P.reads[r];
    const int blen = rr.len;
    int n = rr.nkeys;
    u.blen = blen;
    if (!done) {
        if (n < 1 || blen < p.k) { result = 0; done = true; }
        else if (n > KB || blen > MAXLEN) { result = -2; done = true; }
        else if (n > 64 || blen > WLEN) { result = NSITES_PENDING; done = true; }
    }
    // one uniform do { } while (0) body per read: `break` = finished with `result`
    if (!done) do {
        const uint8_t *bP = P.bases + rr.bases_off;
        const int8_t *qP = P.baseScores + rr.bases_off;
        const int *koff = P.keyinfo + rr.keys_off, *kscore = koff + n;
        int sumBS = 0; bool undefinedBase = false;
        for (int i = lane; i < blen; i += 64) {
            const int b = bP[i], q = qP[i];
            S.base[0][i] = (uint8_t)b; S.base[1][blen - 1 - i] = compLut[b]; S.bsc[i] = (int8_t)q;
            sumBS += q;
            const int cd = numLut[b];
            S.code[i] = (int8_t)cd;
            if (cd < 0) undefinedBase = true;
        }
        sumBS = wsum(sumBS);
        const bool fullyDefined = __ballot(undefinedBase) == 0;
        wsync();
        if (P.rcOut) {                                       // AminoAcid.reverseComplementBases for the DP stage, coalesced
            uint8_t *rc = P.rcOut + rr.bases_off;
            for (int i = lane; i < blen; i += 64) rc[i] = S.base[1][i];
        }

        // KeyRing.makeKeys; lane i owns key i
        int off = 0, ksc = 0, keyOrig = -1;
        if (lane < n) {
            off = koff[lane]; ksc = kscore[lane];
            int key = 0;
            for (int q = off; q < off + p.k; q++) { const int x = S.code[q]; if (x < 0) { key = -1; break; } key = (key << 2) | x; }
            keyOrig = key;
        }
        {   // the wave kernel's coverage arithmetic needs non-decreasing offsets (KeyRing.makeOffsets gives them)
            const int prevOff = __shfl_up(off, 1);
            if (__ballot(lane > 0 && lane < n && off < prevOff)) { result = NSITES_PENDING; break; }
        }
        // one 32-byte fused record per key: COUNTS and the list heads of the key and of its reverse complement
        KeyEntry e; e.cnt = e.cntRC = e.startF = e.lenF = e.firstF = e.startR = e.lenR = e.firstR = 0;
        if (keyOrig >= 0) e = ix.fused[0][keyOrig];
        const int cntOrig = e.cnt;
        int key = keyOrig;
        auto countHits = [&](int maxLen) -> int {
            const bool v = key >= 0 && cntOrig > 0 && cntOrig < maxLen;
            if (!v) key = -1;
            return popc(__ballot(v));
        };
        const int maxLen = p.maxUsableLength;
        int numHits = countHits(maxLen);
        if (numHits > 0) {
            const int trigger = (3 * n) / 4;
            if (numHits < 4 && numHits < trigger) { key = keyOrig; numHits = countHits((maxLen * 3) / 2); }
            if (numHits < 3 && numHits < trigger) { key = keyOrig; numHits = countHits(maxLen * 2); }
            if (numHits < 3 && numHits < trigger) { key = keyOrig; numHits = countHits(maxLen * 3); }
            if (numHits < 2 && numHits < trigger) { key = keyOrig; numHits = countHits(maxLen * 5); }
        }
        PH_MARK(u, 0);
        const int nOriginal = n;
        int cnt = cntOrig;
        auto compactKeys = [&]() {
            const int src = compactSrc(S, lane, lane < n && key >= 0, n);
            off = __shfl(off, src); key = __shfl(key, src); ksc = __shfl(ksc, src); cnt = __shfl(cnt, src);
            e.cntRC = __shfl(e.cntRC, src); e.startF = __shfl(e.startF, src); e.lenF = __shfl(e.lenF, src); e.firstF = __shfl(e.firstF, src);
            e.startR = __shfl(e.startR, src); e.lenR = __shfl(e.lenR, src); e.firstR = __shfl(e.firstR, src);
            if (lane >= n) key = -1;
        };
        if (numHits < n) compactKeys();
        if (p.trimByGreedy) {
            const int maxLists = max((int)(HIT_FRACTION_TO_RETAIN * n), MIN_LISTS_RETAIN);
            numHits = trimByGreedyW(u, S, off, ksc, n, maxLists, key, cnt);
        }
        if (numHits < p.minApproxHitsToKeep) { result = 0; break; }
        if (numHits < n) compactKeys();
        // minus strand: KeyRing.reverseComplementKeys / reverseOffsets
        int offM = 0, keyM = -1, kscM = 0;
        {
            const int src = (lane < n) ? n - 1 - lane : lane;
            const int so = __shfl(off, src), sk = __shfl(key, src), ss = __shfl(ksc, src);
            if (lane < n) { offM = blen - (so + p.k); keyM = rc_key(sk, p.k); kscM = ss; }
        }
        // every (block, strand) cycle reloads the fused records of its keys: 32 bytes per key lane, two vector loads.  (Parking block
        // 0's records in LDS saved the reload but cost 2 KB per wave, which the cycle's presence maps make better use of.  Round 3:
        // one load per block held in registers across its two strands, with the next block's records requested ahead, cost 8 more
        // live VGPRs: 62.8 -> 71.7 ms at 5 waves per SIMD (spills), 88.7 -> 72.2 ms at 4 -- the reload is the cheaper choice.)
        auto keyHits = [&](int block, int strand) -> KeyHit {
            KeyEntry eb; eb.cnt = eb.cntRC = eb.startF = eb.lenF = eb.firstF = eb.startR = eb.lenR = eb.firstR = 0;
            if (lane < n && key >= 0) eb = ix.fused[block][key];
            if (strand) return minusView(lane, n, eb.cntRC, eb.startR, eb.lenR, eb.firstR);
            KeyHit h; h.cnt = eb.cnt; h.start = eb.startF; h.len = eb.lenF; h.first = eb.firstF;
            return h;
        };
        const int mqs = maxQuickScoreW(u, off, ksc, n);
        int bestScores[6] = {0, 0, 0, 0, 0, 0};
        const bool prescan = p.prescanQscore && numHits >= 5;
        int hitsCutoff = 0, qscoreCutoff = (int)(MIN_QSCORE_MULT * mqs);
        bool allBasesCovered = true, pretend;
        {
            const int off0 = rl(off, 0), offLast = rl(off, n - 1);
            const int prevOff = __shfl_up(off, 1);
            if (off0 != 0 || offLast != blen - p.k) allBasesCovered = false;
            else if (__ballot(lane > 0 && lane < n && off > prevOff + p.k)) allBasesCovered = false;
            pretend = allBasesCovered || n >= nOriginal - 4 || (n >= 9 && (offLast - off0 + p.k) > max(40, (int)(blen * .75f)));
        }

        PH_MARK(u, 1);
        const int cpb = u.c.cpb;
        int ncycles = 0;
        for (int chrom = p.minChrom; chrom <= p.maxChrom; chrom = ((chrom & u.c.highMask) + cpb)) ncycles += 2;
        if (ncycles > 64) { result = -2; break; }
        WL L;
        L.row = L.stop = L.offs = L.ksc = L.nbuf = 0; L.value = LANE_UNUSED; L.hv = INT_MAX; for (int j = 0; j < NB; j++) L.nb[j] = 0; L.n = L.nlive = 0; L.bulk = -1; L.bwait = -1; L.sites = nullptr;
        int precount = n, prescore = mqs;                 // lane c holds the prescan result of cycle c
        bool dead = false;
        if (prescan) {
            int bestqscore = 0, maxHits = 0, minHitsToScore = p.minApproxHitsToKeep, cycle = 0; bool earlyOut = false;
            for (int chrom = p.minChrom; chrom <= p.maxChrom && !earlyOut; chrom = ((chrom & u.c.highMask) + cpb)) {
                const int baseChrom = u.c.baseChrom(chrom);
                const int block = baseChrom >> p.chromBits;
                for (int pmi = 0; pmi < 2 && !earlyOut; pmi++, cycle++) {
                    PH_PRE(u, 3);
                    const int nh = makeListsW<LONG>(u, S, L, block, baseChrom, keyHits(block, pmi), pmi ? offM : off, pmi ? kscM : ksc, n, minHitsToScore,
                                                    !(LONG && BBIDX_CYCLE));
                    PH_PRE(u, 0);
                    if (nh < minHitsToScore) { if (lane == cycle) { prescore = -9999; precount = 0; } }
                    else {
                        int tq, th;
                        const bool perfectOnly = bestqscore >= mqs && pretend;
                        if (!(LONG && BBIDX_CYCLE && !perfectOnly && findMaxQscore2Cycle(u, cyc[LONG ? wave : 0], S, L, baseChrom, minHitsToScore, n, mqs, tq, th))) {
                            PH_PRE(u, 3);
                            if (LONG && BBIDX_CYCLE) refillLists(L);
                            findMaxQscore2W<LONG>(u, L, baseChrom, minHitsToScore, perfectOnly, n, mqs, tq, th);
                            PH_PRE(u, 4);
                        }
                        if (lane == cycle) { prescore = tq; precount = th; }
                        bestqscore = max(tq, bestqscore); maxHits = max(maxHits, th);
                        if (bestqscore >= mqs && pretend) { minHitsToScore = max(minHitsToScore, maxHits); earlyOut = true; }
                    }
                }
            }
            bestScores[1] = max(bestScores[1], maxHits);
            bestScores[3] = max(bestScores[3], bestqscore);
            if (bestScores[1] < p.minApproxHitsToKeep) dead = true;
            else if ((float)bestScores[3] < __fmul_rn((float)mqs, MIN_QSCORE_MULT2)) dead = true;
            else if (bestScores[3] >= mqs && pretend) {
                hitsCutoff = calcApproxHitsCutoff(p, n, bestScores[1], p.minApproxHitsToKeep, true);
                qscoreCutoff = max(qscoreCutoff, (int)(bestScores[3] * DYN_QSCORE_PERFECT));
            } else {
                hitsCutoff = calcApproxHitsCutoff(p, n, bestScores[1], p.minApproxHitsToKeep, false);
                qscoreCutoff = max(qscoreCutoff, (int)(bestScores[3] * PRESCAN_QSCORE_THRESH));
            }
        }
        PH_PRE(u, 3); PH_MARK(u, 2);
        if (uni(dead)) { result = 0; break; }
        hitsCutoff = uni(hitsCutoff); qscoreCutoff = uni(qscoreCutoff); n = uni(n);

        const int maxScore = 70 + (blen - 1) * 100 + sumBS;               // msa.maxQuality(baseScores)
        SiteOut ssl; ssl.v = P.sites + r * (long long)P.maxSites; ssl.n = 0; ssl.cap = P.maxSites; ssl.overflow = false;
        int cycle = 0; bool quit = false;
        for (int chrom = p.minChrom; chrom <= p.maxChrom && !quit; chrom = ((chrom & u.c.highMask) + cpb)) {
            const int baseChrom = u.c.baseChrom(chrom);
            const int block = baseChrom >> p.chromBits;
            for (int strand = 0; strand < 2 && !quit; strand++, cycle++) {
                for (int j = 0; j < 6; j++) bestScores[j] = uni(bestScores[j]);
                ssl.n = uni(ssl.n); ssl.overflow = uni(ssl.overflow); cycle = uni(cycle); quit = uni(quit);
                if (!prescan || rl(precount, cycle) >= hitsCutoff || rl(prescore, cycle) >= qscoreCutoff) {
                    PH_WALK(u, 1);
                    const int nh = makeListsW<LONG>(u, S, L, block, baseChrom, keyHits(block, strand), strand ? offM : off, strand ? kscM : ksc, n, p.minApproxHitsToKeep,
                                                    !(LONG && BBIDX_CYCLE));
                    if (nh >= p.minApproxHitsToKeep)
                        slowWalk3W<LONG>(u, S, LONG ? &cyc[wave] : nullptr, L, strand, n, mqs, chrom, ssl, bestScores, allBasesCovered, maxScore, fullyDefined);
                }
                if (p.quitAfterTwoPerfects && bestScores[5] >= 2) quit = true;
            }
        }
        result = ssl.overflow ? -1 : ssl.n;
        PH_MARK(u, 3);
        cSites += (unsigned)ssl.n;
    } while (0);

    if (lane == 0) {
        P.nsites[r] = result;
        if (result == NSITES_PENDING) atomicAdd(&P.queue[1], 1u);
    }
    wsync();                             // the next read reuses this wave's LDS
    }   // next read
    if (P.stats) {
        if (lane == 0) {
#if defined(BBIDX_BATCH_STATS) || defined(BBIDX_CYC_STATS)
            for (int j = 0; j < 5; j++) atomicAdd(&blockStats[j], u.ph[j]);
#elif defined(BBIDX_PHASE_TIMERS)
            for (int j = 0; j < 5; j++) atomicAdd(&blockStats[j], u.ph[j] >> 4);
#else
            atomicAdd(&blockStats[0], u.cPrescan); atomicAdd(&blockStats[1], u.cWalk); atomicAdd(&blockStats[2], u.cExtend);
            atomicAdd(&blockStats[3], u.cRefBytes); atomicAdd(&blockStats[4], cSites);
#endif
        }
        __syncthreads();
        if (threadIdx.x < 5 && blockStats[threadIdx.x])
            atomicAdd(&P.stats[8 * (blockIdx.x % STAT_SHARDS) + threadIdx.x], (unsigned long long)blockStats[threadIdx.x]);
    }
}

}  // namespace bbidxw

int bbidx_launch_wave(const bbidx::Params &P, hipStream_t stream, bool longLists, int maxReadLen) {
    using namespace bbidxw;
    const bool shortReads = maxReadLen <= WSHORTLEN;
    // as many blocks as fit the device at once; they pull reads from a queue (Params.queue[2])
    static int perCU[4] = {0, 0, 0, 0}, numCUs = 0;
    const int variant = (longLists ? 2 : 0) + (shortReads ? 1 : 0);
    if (perCU[variant] == 0) {
        const void *fn = longLists ? (shortReads ? (const void *)probe_wave_kernel<true, WSHORTLEN> : (const void *)probe_wave_kernel<true, WMAXLEN>)
                                   : (shortReads ? (const void *)probe_wave_kernel<false, WSHORTLEN> : (const void *)probe_wave_kernel<false, WMAXLEN>);
        int per = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, 64 * WAVES_PER_BLOCK, 0) != hipSuccess || per < 1) per = 1;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) numCUs = prop.multiProcessorCount;
        if (numCUs < 1) numCUs = 256;
        perCU[variant] = per;
    }
    long long blocks = (P.nreads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks > (long long)numCUs * perCU[variant]) blocks = (long long)numCUs * perCU[variant];
    const dim3 g((unsigned)blocks), b(64 * WAVES_PER_BLOCK);
    if (longLists) {
        if (shortReads) hipLaunchKernelGGL((probe_wave_kernel<true, WSHORTLEN>), g, b, 0, stream, P);
        else hipLaunchKernelGGL((probe_wave_kernel<true, WMAXLEN>), g, b, 0, stream, P);
    } else {
        if (shortReads) hipLaunchKernelGGL((probe_wave_kernel<false, WSHORTLEN>), g, b, 0, stream, P);
        else hipLaunchKernelGGL((probe_wave_kernel<false, WMAXLEN>), g, b, 0, stream, P);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        static thread_local char msg[256];
        snprintf(msg, sizeof msg, "probe_wave_kernel launch failed: %s", hipGetErrorString(e));
        bbmap_set_error(msg);
        return BBMAP_E_HIP;
    }
    return BBMAP_OK;
}
