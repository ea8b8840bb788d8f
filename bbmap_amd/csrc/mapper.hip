// Mapper control flow around the index probe and the DP, device-resident (include/bbmap_amd.h, "Mapper control flow").
//
// The reference runs this logic in Java, one read (pair) at a time, between its two hot kernels:
//   BBMapThread.processRead      current/align2/BBMapThread.java:389-490
//   BBMapThread.processReadPair  current/align2/BBMapThread.java:943-1098
// A GPU cannot go back to the host between the probe and every single alignment, so the same decisions are taken here by
// small kernels over the whole batch.  Design:
//   * one thread per read (per pair where the two mates interact, per rescue search in the rescue stage): the logic is
//     short, branchy, list-shaped code over a handful of 128-byte site records; a read's records are contiguous in HBM;
//   * scoreSlow is a per-read SEQUENCE (a site's minScore depends on the results of the sites before it, and a fill can ask
//     for a second, wider fill), so every read carries a small state machine and the DP kernels run in ROUNDS: round j
//     aligns the j-th fill of every read that still has one.  Round 1 holds nearly all the work (most reads have one
//     candidate); later rounds are small.  The host only reads three counters per round;
//   * rescue is a stage per anchor mate (mate 1, then mate 2: the second pass sees the sites the first one added):
//     plan (which anchor sites search) -> quick_rescue_kernel (pipeline.hip) -> prepare (ungapped score, tip deletions, DP
//     job) -> DP -> finish (retain / pair / append / merge duplicates).
// Every function names the reference lines it follows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "bbmap_amd.h"
#include "index_ctx.h"

void bbmap_set_error(const char *msg);
void bbmsa_use_narrow(bbmsa_ctx *c, bool on);          // msa_host.hip (internal, see msa_ctx.h)
int bbmsa_wait_first_pass(bbmsa_ctx *c, void *waiter);
void bbmsa_sort_by_width(bbmsa_ctx *c, bool on);
int bbmsa_set_latency_jobs(bbmsa_ctx *c, int64_t n);

namespace bbmapper {

typedef bbmap_msite Site;

constexpr int GAPBUFFER2 = 128, GAPLEN = 128, MINGAP = 256;                       // Shared.java:21-26
constexpr int TIP_MAX_TIPLEN = 8, OUTER_DIST_MULT = 14, OUTER_DIST_DIV = 32;      // AbstractMapThread.java:2987-2993
constexpr int MIN_TRIM_SINGLE = 3, MIN_TRIM_PAIRED = 2;                           // BBMapThread.java:62-63
constexpr int GAPPED_BIT = 1 << 30;
constexpr int DEAD_MARK = 0x7fffffff;

struct Settings {
    float minRatio, ratioPaired, ratioPreRescue;
    int slowAlignPadding, slowRescuePadding, extraPadding, tipSearchDist, maxPairDist, averagePairDist, maxRescueDist,
        maxRescueMismatches, maxTrimSitesToRetain, trimList, doRescue, alignColumns, clearzone3, maxIndel, expLimit, paired;
    // the aligner class's points (MultiStateAligner11ts: jni/MultiStateAligner11tsJNI.c:18-98; MultiStateAligner9PacBio:
    // current/align2/MultiStateAligner9PacBio.java:2375-2407): POINTS_MATCH, POINTS_MATCH2, POINTS_SUB / SUB2 / SUB3,
    // min(POINTS_DEL, POINTS_INS - POINTS_MATCH2) of maxImperfectScore, and CLEARZONE1e = 2*MATCH2 - MATCH - SUB + 1
    // (AbstractMapThread.java:142)
    int ptsMatch, ptsMatch2, ptsSub, ptsSub2, ptsSub3, impDelta, clearzone1e;
    int msaMaxColumns;          // columns of the reference's MSA instance (realign_new's padding rules read msa.maxColumns)
    int finalStage;
};

struct SlowState {      // scoreSlow's loop state of one read
    int idx;            // site being worked on
    int phase;          // 0 = look at site idx, 1 = first fill in flight, 2 = wider refill in flight, 3 = finished
    int minMsaLimit;
    int pending;        // job index of the fill in flight (GAPPED_BIT for the gapped log)
    int oldJob;         // phase 2: the first fill
    int expectedLen;
    int minscore;
    int seq;            // fills issued for this read so far
};

struct PairResc {       // rescue(): per pair and pass
    int first, count;   // its searches in the rescue job list
    int maxMismatches, retainLimit, retainLimit2, findTip;
    int unpaired2;      // pass A remembers mate 2's unpaired count for pass B (BBMapThread.java:1075-1081 runs before both)
    int ran;            // this pass's `if(unpaired>0 && numSites>0)` block runs (its mergeDuplicateSites of the loose list included)
};

struct RescInfo { int pair, anchorSite, strand, job; };   // per rescue search; job = DP job index or -1

struct Dev {
    Settings S;
    const bbidx_read *reads;
    const uint8_t *bases;
    long long minusDelta, nreads;
    const uint8_t *const *chromArr;
    const int *chromArrLen;
    const uint8_t *refsBase;
    const bbidx_site *psites; const int *pnsites; int maxSites;
    Site *ms; int *mcount; int cap;
    int *nearArr;
    SlowState *slow;
    const int *activeIn; int *activeOut; int nActiveIn;
    unsigned *counters;         // [0] plain fills, [1] gapped fills, [2] next active count, [3] overflowed reads, [4] rescue searches,
                                // [5] reads without site, [6] refills, [7] rescue fills, [8] fills ahead of time that were dropped
    bbmsa_job *jobs; bbmap_jobinfo *jinfo; const bbmsa_result *results; long long jobCap;
    bbmsa_job *gjobs; bbmsa_gaps *ggaps; bbmap_jobinfo *ginfo; const bbmsa_result *gresults; long long gjobCap;
    bbresc_job *rjobs; RescInfo *rinfo; const bbresc_result *rres; PairResc *pres; long long rescCap;
    Site *rsite;                // per rescue search: the SiteScore under construction
    int pass;                   // rescue pass: 0 = mate 1 anchors, 1 = mate 2 anchors
    int plainColumns;           // widest window the first DP context takes
    int fillAhead;              // scoreSlow rounds: fill the sites behind the one in flight ahead of time
    // the final alignment stage (mapper_final.h)
    struct FinalRead *fin; bbmap_final *finalOut;
    uint8_t *pool; long long poolUnits;         // match strings: bump-allocated in 4-byte units, counters[20] = units in use
    const uint8_t *match, *gmatch; int matchStride, gmatchStride;
                                // counters: [20] pool units handed out (beyond the capacity once a request failed), [21] units in use when the first
                                // request failed, [22] requests that failed, [24] reads that need toLocalAlignment, [25] pool units those may take
};

__device__ inline int imin(int a, int b) { return a < b ? a : b; }
__device__ inline int imax(int a, int b) { return a > b ? a : b; }
__device__ inline int iabsdif(int a, int b) { return a > b ? a - b : b - a; }
__device__ inline int max_quality(const Settings &S, int len) { return S.ptsMatch + (len - 1) * S.ptsMatch2; }      // MSA.maxQuality
__device__ inline int max_imperfect(const Settings &S, int len) { return max_quality(S, len) + S.impDelta; }      // maxImperfectScore

// ---------------------------------------------------------------------------------------------- SiteScore / GapTools
__device__ void fix_gaps2(Site &ss) {                                                        // GapTools.fixGaps2 :127-175
    int ra[BBMSA_MAX_GAPS / 2], rb[BBMSA_MAX_GAPS / 2];
    bool alive[BBMSA_MAX_GAPS / 2];
    const int nr = ss.ngaps / 2;
    for (int i = 0; i < nr; i++) { ra[i] = ss.gaps[2 * i]; rb[i] = ss.gaps[2 * i + 1]; alive[i] = true; }
    for (int i = 1; i < nr; i++)
        if (alive[i - 1] && ra[i] - rb[i - 1] <= MINGAP) { ra[i] = imin(ra[i - 1], ra[i]); rb[i] = imax(rb[i - 1], rb[i]); alive[i - 1] = false; }
    int m = 0;
    for (int i = 0; i < nr; i++) if (alive[i]) { ss.gaps[2 * m] = ra[i]; ss.gaps[2 * m + 1] = rb[i]; m++; }
    ss.ngaps = m < 2 ? 0 : 2 * m;
}
__device__ void fix_gaps(Site &ss) {                                                         // GapTools.fixGaps :27-72
    if (ss.ngaps == 0) return;
    const int a = ss.start, b = ss.stop, n = ss.ngaps;
    int *g = ss.gaps;
    if (!(g[0] <= b && g[n - 1] >= a)) { ss.ngaps = 0; return; }
    int changed = 0;
    if (g[0] != a) { g[0] = a; changed++; }
    if (g[n - 1] != b) { g[n - 1] = b; changed++; }
    for (int i = 0; i < n; i++) { if (g[i] < a) { g[i] = a; changed++; } else if (g[i] > b) { g[i] = b; changed++; } }
    for (int i = 1; i < n; i++) if (g[i - 1] > g[i]) { g[i] = g[i - 1]; changed++; }
    if (changed == 0) return;
    g[0] = a; g[n - 1] = b;
    int remove = 0;
    for (int i = 0; i < n; i += 2) {
        g[i] = imin(imax(g[i], a), b); g[i + 1] = imin(imax(g[i + 1], a), b);
        if (g[i] == g[i + 1]) remove++;
    }
    if (remove) fix_gaps2(ss);
}
__device__ bool check_gaps(const Site &ss) {                                                 // SiteScore.CHECKGAPS
    if (ss.ngaps == 0) return true;
    if (ss.ngaps & 1) return false;
    for (int i = 1; i < ss.ngaps; i++) if (ss.gaps[i - 1] > ss.gaps[i]) return false;
    return ss.gaps[0] == ss.start && ss.gaps[ss.ngaps - 1] == ss.stop;
}
__device__ void set_limits(Site &ss, int a, int b) {                                         // SiteScore.java:905-914
    ss.start = a; ss.stop = b;
    if (ss.ngaps) { ss.gaps[0] = a; ss.gaps[ss.ngaps - 1] = b; if (!check_gaps(ss)) fix_gaps(ss); }
}
__device__ void set_start(Site &ss, int a) {                                                 // :933-942
    ss.start = a;
    if (ss.ngaps) { ss.gaps[0] = a; if (ss.gaps[0] > ss.gaps[1]) fix_gaps(ss); }
}
__device__ void set_stop(Site &ss, int b) {                                                  // :943-951
    ss.stop = b;
    if (ss.ngaps) { ss.gaps[ss.ngaps - 1] = b; fix_gaps(ss); }
}
__device__ void set_slow_score(Site &ss, int x) {                                            // :962-983
    if (x <= 0) ss.pairedScore = x;
    else if (ss.pairedScore > 0) ss.pairedScore = ss.slowScore > 0 ? x + (ss.pairedScore - ss.slowScore) : x + 1;
    ss.slowScore = x;
}
__device__ int calc_gref_len(const Site &ss) {                                               // GapTools.calcGrefLen :80-92
    int total = ss.stop - ss.start + 1;
    for (int i = 2; i < ss.ngaps; i += 2) total -= imax(0, (ss.gaps[i] - ss.gaps[i - 1] - GAPBUFFER2) / GAPLEN) * (GAPLEN - 1);
    return total;
}

// ---------------------------------------------------------------------------------------------- byte scans (one thread)
// A thread's consecutive bytes through aligned 32-bit loads: one thread per read means 64 lanes in 64 different cache lines, and a
// byte load costs the L1 as much as a dword load -- four bytes per access instead of one.  Only words that hold a byte below `n`
// are touched.
struct Words {
    const unsigned *w; unsigned lo; int sh, n, k;
    __device__ void init(const uint8_t *p, int n_) {
        sh = (int)((unsigned long long)p & 3ull); w = (const unsigned *)(p - sh); n = n_; k = 0;
        lo = n > 0 ? w[0] : 0u;
    }
    __device__ unsigned next() {                       // bytes p[4k .. 4k+3] of the k-th call
        k++;
        const unsigned hi = (4 * k - sh < n) ? w[k] : 0u;
        const unsigned x = sh ? __builtin_amdgcn_alignbyte(hi, lo, (unsigned)sh) : lo;
        lo = hi;
        return x;
    }
};
// MSA.scoreNoIndels(read, ref, refStart) (MultiStateAligner11tsJNI.java:1034-1089; MultiStateAligner9PacBio.java:1876-1937 is the
// same statement with its own points)
__device__ int score_no_indels(const Settings &S, const uint8_t *read, int len, const uint8_t *ref, int reflen, int refStart) {
    int readStart = 0, readStop = len;
    if (refStart < 0) readStart = -refStart;
    if (refStart + len > reflen) readStop -= (refStart + len - reflen);
    int score = 0, mode = -1, t = 0;                  // mode 0 = match streak, 1 = substitution streak
    const int n = readStop - readStart;
    if (n <= 0) return 0;
    Words A, B;
    A.init(read + readStart, n); B.init(ref + refStart + readStart, n);
    for (int i = 0; i < n; i += 4) {
        const unsigned c4 = A.next(), r4 = B.next();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (i + q < n) {
                const int c = (int)((c4 >> (8 * q)) & 255u), r = (int)((r4 >> (8 * q)) & 255u);
                if (c == r && c != 'N') { if (mode == 0) { t++; score += S.ptsMatch2; } else { t = 0; score += S.ptsMatch; } mode = 0; }
                else if (c >= 128 || c == 'N') { }
                else if (r >= 128 || r == 'N') { }
                else { if (mode == 1) t++; else t = 0; score += (t + 1 > 5 ? S.ptsSub3 : (t + 1 > 1 ? S.ptsSub2 : S.ptsSub)); mode = 1; }
            }
        }
    }
    return score;
}
// SiteScore.setPerfect(bases) (current/stream/SiteScore.java:239-292)
__device__ void set_perfect(Site &ss, const uint8_t *bases, int len, const uint8_t *ref, int reflen) {
    ss.perfect = 0; ss.semiperfect = 0;
    if (len != ss.stop - ss.start + 1) return;
    bool perfect = true, semi = true;
    int refloc = ss.start, readloc = 0, N = 0;
    const int mx = imin(ss.stop, reflen - 1), nlimit = len / 2;
    if (ss.start < 0) { N -= ss.start; readloc -= ss.start; refloc -= ss.start; perfect = false; }
    if (ss.stop >= reflen) { N += (ss.stop - reflen + 1); perfect = false; }
    if (N > nlimit) return;
    const int n = mx - refloc + 1;
    if (n > 0) {
        Words A, B;
        A.init(bases + readloc, n); B.init(ref + refloc, n);
        for (int i = 0; i < n; i += 4) {
            const unsigned c4 = A.next(), r4 = B.next();
            if (c4 == r4 && !(((c4 ^ 0x4E4E4E4Eu) - 0x01010101u) & ~(c4 ^ 0x4E4E4E4Eu) & 0x80808080u) && i + 4 <= n) continue;   // four equal bases, none of them 'N'
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (i + q < n) {
                    const int c = (int)((c4 >> (8 * q)) & 255u), r = (int)((r4 >> (8 * q)) & 255u);
                    if (c != r || c == 'N') {
                        perfect = false;
                        if (c == 'N') semi = false;
                        if (r != 'N' || (N = N + 1) > nlimit) return;
                    }
                }
            }
        }
    }
    semi = semi && N <= nlimit;
    perfect = perfect && semi && N == 0;
    ss.perfect = perfect; ss.semiperfect = semi;
}
// findTipDeletionsRight / Left (AbstractMapThread.java:2178-2292); ChromosomeArray.minIndex is 0 for these arrays
__device__ int tip_right(const uint8_t *bases, int len, const uint8_t *ref, int reflen, int originalStop, int searchDist, int tiplen) {
    if (originalStop < tiplen - 1) return 0;
    int bestStart = originalStop, lastMismatch = 0, originalMismatches = 0, contig = 0;
    const int tipCoord = len - 1;
    for (int i = 0; i < tiplen && contig < 5; i++) {
        if (bases[tipCoord - i] != ref[originalStop - i]) { originalMismatches++; lastMismatch = i; contig = 0; } else contig++;
    }
    if (originalMismatches < 3) return 0;
    int minMismatches = originalMismatches;
    tiplen = lastMismatch + 1;
    if (tiplen < 4) return 0;
    searchDist = imin(searchDist, 30 * originalMismatches);
    const int last = imin(reflen - 1, originalStop + searchDist);
    // The tip (byte j = bases[tipCoord - j]) against a sliding window of the reference (byte j = ref[start - j]): one reference byte
    // enters per position and the mismatches of the first `tiplen` bytes are counted in one go -- the same count the byte loop
    // reaches whenever it stays below minMismatches, which is all the comparison uses.
    unsigned long long T = 0, Wd = 0;
    for (int j = 0; j < 8; j++) T |= (unsigned long long)bases[tipCoord - j] << (8 * j);
    for (int j = 1; j < 8; j++) Wd |= (unsigned long long)ref[originalStop + 1 - j] << (8 * (j - 1));      // the window of start - 1, about to shift
    const unsigned long long keep = tiplen >= 8 ? 0x8080808080808080ull : ((1ull << (8 * tiplen)) - 1) & 0x8080808080808080ull;
    for (int start = originalStop + 1; start <= last && minMismatches > 0; start++) {
        Wd = (Wd << 8) | ref[start];
        const unsigned long long x = Wd ^ T;
        const int mm = __builtin_popcountll((((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & keep);
        if (mm < minMismatches) { bestStart = start; minMismatches = mm; }
    }
    if (minMismatches > 2 || originalMismatches - minMismatches < 2) return 0;
    return bestStart - originalStop;
}
__device__ int tip_left(const uint8_t *bases, const uint8_t *ref, int reflen, int originalStart, int searchDist, int tiplen) {
    if (originalStart + tiplen >= reflen) return 0;
    if (0 >= originalStart) return 0;
    int bestStart = originalStart, lastMismatch = 0, originalMismatches = 0, contig = 0;
    for (int i = 0; i < tiplen && contig < 5; i++) {
        if (bases[i] != ref[originalStart + i]) { originalMismatches++; lastMismatch = i; contig = 0; } else contig++;
    }
    if (originalMismatches < 3) return 0;
    int minMismatches = originalMismatches;
    tiplen = lastMismatch + 1;
    if (tiplen < 4) return 0;
    searchDist = imin(searchDist, 16 + 16 * originalMismatches + 8 * tiplen);
    const int last = imax(0, originalStart - searchDist);
    // as in tip_right: tip byte j = bases[j], window byte j = ref[start + j]; the window moves left one reference byte at a time
    unsigned long long T = 0, Wd = 0;
    for (int j = 0; j < 8; j++) T |= (unsigned long long)bases[j] << (8 * j);
    for (int j = 1; j < 8; j++) Wd |= (unsigned long long)ref[originalStart - 1 + j] << (8 * (j - 1));     // the window of start + 1, about to shift
    const unsigned long long keep = tiplen >= 8 ? 0x8080808080808080ull : ((1ull << (8 * tiplen)) - 1) & 0x8080808080808080ull;
    for (int start = originalStart - 1; start >= last && minMismatches > 0; start--) {
        Wd = (Wd << 8) | ref[start];
        const unsigned long long x = Wd ^ T;
        const int mm = __builtin_popcountll((((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & keep);
        if (mm < minMismatches) { bestStart = start; minMismatches = mm; }
    }
    if (minMismatches > 2 || originalMismatches - minMismatches < 2) return 0;
    return originalStart - bestStart;
}
// findTipDeletions(ss, bases, maxImperfectScore, lookRight, lookLeft) (AbstractMapThread.java:1107-1141)
__device__ bool find_tip_deletions(const Settings &S, Site &ss, const uint8_t *bases, int len, const uint8_t *ref, int reflen,
                                   int maxImp, bool lookRight, bool lookLeft) {
    if (ss.slowScore >= maxImp) return false;
    if (len <= 2 * TIP_MAX_TIPLEN) return false;
    int maxSearch = imin(S.tipSearchDist, S.alignColumns - (S.slowRescuePadding + 8 + imax(len, ss.stop - ss.start)));
    if (maxSearch < 1) return false;
    bool changed = false;
    if (lookRight) {
        const int x = tip_right(bases, len, ref, reflen, ss.stop, maxSearch, TIP_MAX_TIPLEN);
        if (x > 0) {
            set_stop(ss, ss.stop + x); changed = true;
            maxSearch = imin(maxSearch, S.alignColumns - (S.slowRescuePadding + 8 + imax(len, ss.stop - ss.start)));
            if (maxSearch < 1) return changed;
        }
    }
    if (lookLeft) {
        const int y = tip_left(bases, ref, reflen, ss.start, maxSearch, TIP_MAX_TIPLEN);
        if (y > 0) { set_start(ss, ss.start - y); changed = true; }
    }
    return changed;
}

// ---------------------------------------------------------------------------------------------- list tools
__device__ inline int cmp_score(const Site &a, const Site &b) {                              // SiteScore.compareTo :55-73
    int x = b.score - a.score; if (x) return x;
    x = b.slowScore - a.slowScore; if (x) return x;
    x = b.pairedScore - a.pairedScore; if (x) return x;
    x = b.quickScore - a.quickScore; if (x) return x;
    x = a.chrom - b.chrom; if (x) return x;
    return a.start - b.start;
}
__device__ inline int cmp_pos(const Site &a, const Site &b) {                                // PositionComparator :379-395
    if (a.chrom != b.chrom) return a.chrom - b.chrom;
    if (a.start != b.start) return a.start - b.start;
    if (a.stop != b.stop) return a.stop - b.stop;
    if (a.strand != b.strand) return a.strand - b.strand;
    if (a.score != b.score) return b.score - a.score;
    if (a.slowScore != b.slowScore) return b.slowScore - a.slowScore;
    if (a.quickScore != b.quickScore) return b.quickScore - a.quickScore;
    if (a.perfect != b.perfect) return a.perfect ? -1 : 1;
    if (a.rescued != b.rescued) return a.rescued ? 1 : -1;
    return 0;
}
// stable insertion sort (Collections.sort is a stable merge sort: same order)
template <bool BYPOS> __device__ void sort_sites(Site *s, int n) {
    for (int i = 1; i < n; i++) {
        if ((BYPOS ? cmp_pos(s[i - 1], s[i]) : cmp_score(s[i - 1], s[i])) <= 0) continue;       // already in place (the common case)
        const Site t = s[i];
        int j = i - 1;
        while (j >= 0 && (BYPOS ? cmp_pos(s[j], t) : cmp_score(s[j], t)) > 0) { s[j + 1] = s[j]; j--; }
        s[j + 1] = t;
    }
}
// Entries marked for removal (Tools.condenseStrict's nulls): the first 64 list positions in a register mask, positions beyond
// (only the overflow tier has lists that long) as a mark in the record itself (reserved[1], zero in every list entry otherwise).
struct DeadSet {
    unsigned long long lo = 0; bool hi = false;
    __device__ void mark(Site *s, int i) { if (i < 64) lo |= 1ull << i; else { s[i].reserved[1] = DEAD_MARK; hi = true; } }
    __device__ bool dead(const Site *s, int i) const { return i < 64 ? ((lo >> i) & 1) != 0 : (hi && s[i].reserved[1] == DEAD_MARK); }
    __device__ bool any() const { return lo != 0 || hi; }
};
// order-preserving removal of the marked entries
__device__ int condense(Site *s, int n, const DeadSet &dead) {
    if (!dead.any()) return n;
    int m = 0;
    for (int i = 0; i < n; i++) if (!dead.dead(s, i)) { if (m != i) s[m] = s[i]; m++; }
    return m;
}
// Tools.trimSitesBelowCutoff (Tools.java:1113-1161)
__device__ int trim_below_cutoff(Site *s, int n, int cutoff, bool retainPaired, int minRetain, int maxRetain) {
    if (n <= minRetain) return n;
    if (n > maxRetain) n = maxRetain;
    DeadSet dead;
    int removed = 0;
    const int maxToRemove = n - minRetain;
    for (int i = n - 1; i >= 0; i--) {
        if (!s[i].semiperfect && s[i].score < cutoff && (!retainPaired || s[i].pairedScore <= 0)) {       // retainSemiperfect is always true here
            dead.mark(s, i); removed++;
            if (removed >= maxToRemove) break;
        }
    }
    return condense(s, n, dead);
}
// Tools.trimSiteList (Tools.java:654-674)
__device__ int trim_site_list(Site *s, int &n, float fraction, bool retainPaired, int minRetain, int maxRetain) {
    if (n == 0) return -999999;
    if (n == 1) return s[0].score;
    int maxScore = -999999;
    if (minRetain > 1 && minRetain < n) maxScore = s[0].score;
    else for (int i = 0; i < n; i++) maxScore = imax(maxScore, s[i].score);
    n = trim_below_cutoff(s, n, (int)__fmul_rn((float)maxScore, fraction), retainPaired, minRetain, maxRetain);
    return maxScore;
}
// BBMapThread.trimList, USE_AFFINE_SCORE branch (BBMapThread.java:140-197)
__device__ void trim_list(Site *s, int &n, bool retainPaired, int maxScore, bool specialCasePerfect, int minRetain, int maxRetain) {
    if (n < 2) return;
    const int highest = trim_site_list(s, n, .6f, retainPaired, minRetain, maxRetain);
    if (highest == maxScore && specialCasePerfect) {
        trim_site_list(s, n, .94f, retainPaired, minRetain, maxRetain);
        if (n > 8) trim_site_list(s, n, .99f, retainPaired, minRetain, maxRetain);
        return;
    }
    const int mstr2 = minRetain <= 1 ? 1 : minRetain + 1;
    if (n > 4) trim_site_list(s, n, .65f, retainPaired, minRetain, maxRetain);
    if (n > 8) trim_site_list(s, n, .7f, retainPaired, minRetain, maxRetain);
    if (n > 12) trim_site_list(s, n, .75f, retainPaired, minRetain, maxRetain);
    if (n > 16) trim_site_list(s, n, .8f, retainPaired, minRetain, maxRetain);
    if (n > 20) trim_site_list(s, n, .85f, retainPaired, minRetain, maxRetain);
    if (n > 24) trim_site_list(s, n, .9f, retainPaired, minRetain, maxRetain);
    if (n > 32) trim_site_list(s, n, .95f, retainPaired, minRetain, maxRetain);
    if (n > 40) trim_site_list(s, n, .97f, retainPaired, mstr2, maxRetain);
    if (n > 48) trim_site_list(s, n, .99f, retainPaired, mstr2, maxRetain);
}
__device__ bool positional_match(const Site &a, const Site &b, bool testGaps) {              // SiteScore.java:353-365
    if (a.chrom != b.chrom || a.strand != b.strand || a.start != b.start || a.stop != b.stop) return false;
    if (!testGaps || (a.ngaps == 0 && b.ngaps == 0)) return true;
    if (a.ngaps != b.ngaps) return false;
    for (int i = 0; i < a.ngaps; i++) if (a.gaps[i] != b.gaps[i]) return false;
    return true;
}
// Tools.mergeDuplicateSites(list, true, true) (Tools.java:697-759)
__device__ int merge_duplicate_sites(Site *s, int n) {
    if (n < 2) return n;
    sort_sites<true>(s, n);
    DeadSet dead;
    int ai = 0;
    for (int i = 1; i < n; i++) {
        Site &a = s[ai];
        const Site &b = s[i];
        const bool exact = positional_match(a, b, true);
        if (exact || positional_match(a, b, false)) {
            bool takeB = false;                        // different gaps: the better of the two lends its gap array
            if (!exact) {
                if (a.score != b.score) takeB = b.score > a.score;
                else if (a.slowScore != b.slowScore) takeB = b.slowScore > a.slowScore;
                else if (a.pairedScore != b.pairedScore) takeB = b.pairedScore > a.pairedScore;
            }
            set_slow_score(a, imax(a.slowScore, b.slowScore));
            a.pairedScore = (a.pairedScore <= a.slowScore && b.pairedScore <= a.slowScore) ? 0 : imax(0, imax(a.pairedScore, b.pairedScore));
            a.score = imax(a.score, b.score);
            a.perfect = (a.perfect || b.perfect);
            a.semiperfect = (a.semiperfect || b.semiperfect);
            if (takeB) { a.ngaps = b.ngaps; for (int q = 0; q < BBMSA_MAX_GAPS; q++) a.gaps[q] = b.gaps[q]; }
            dead.mark(s, i);
        } else ai = i;
    }
    return condense(s, n, dead);
}
// Tools.removeLowQualitySitesPaired (Tools.java:934-960)
__device__ int remove_low_quality_paired(Site *s, int n, int maxSw, float multSingle, float multPaired) {
    if (n == 0) return 0;
    const int thresh = (int)__fmul_rn((float)maxSw, multSingle), threshPaired = (int)__fmul_rn((float)maxSw, multPaired);
    if (s[0].score < threshPaired) return 0;
    DeadSet dead;
    for (int i = n - 1; i >= 0; i--) {
        if (s[i].pairedScore > 0) { if (s[i].slowScore < threshPaired) dead.mark(s, i); }
        else if (s[i].slowScore < thresh) dead.mark(s, i);
    }
    return condense(s, n, dead);
}

// ---------------------------------------------------------------------------------------------- stage 1: begin
// quickMap's tail for one read: probe records -> SiteScores, removeOutOfBounds (AbstractMapThread.java:2444-2476)
__device__ int load_sites(const Dev &D, long long r, Site *s) {
    const int ns = D.pnsites[r];
    if (ns < 0) return -1;                                 // the probe ran out of room (or declined the read): reported, not mapped
    const bbidx_site *ps = D.psites + r * (long long)D.maxSites;
    const int len = D.reads[r].len;
    int n = 0;
    for (int i = 0; i < ns; i++) {
        Site ss;
        ss.chrom = ps[i].chrom; ss.strand = ps[i].strand; ss.start = ps[i].start; ss.stop = ps[i].stop; ss.hits = ps[i].hits;
        ss.quickScore = ss.score = ps[i].score; ss.slowScore = 0; ss.pairedScore = 0;
        ss.perfect = ps[i].perfect; ss.semiperfect = ps[i].semiperfect; ss.rescued = 0;
        ss.ngaps = ps[i].ngaps;
        for (int q = 0; q < BBMSA_MAX_GAPS; q++) ss.gaps[q] = ps[i].gaps[q];
        ss.match_job = -1; ss.reserved[0] = ss.reserved[1] = 0;
        const int mx = D.chromArrLen[ss.chrom] - 1;
        if (ss.start < 0 || ss.stop > mx) continue;
        if (calc_gref_len(ss) >= D.S.expLimit) { set_stop(ss, ss.start + imin(len + 40, D.S.expLimit)); if (ss.ngaps) fix_gaps(ss); }
        s[n++] = ss;
    }
    return n;
}

// pairSiteScoresInitial (BBMapThread.java:736-940); REQUIRE_CORRECT_STRANDS_PAIRS = true, SAME_STRAND_PAIRS = false
__device__ void pair_initial(const Settings &S, Site *s1, int &n1, Site *s2, int &n2, int len1, int len2) {
    if (n1 < 1 || n2 < 1) return;
    sort_sites<true>(s1, n1); sort_sites<true>(s2, n2);
    for (int i = 0; i < n1; i++) s1[i].pairedScore = 0;
    for (int i = 0; i < n2; i++) s2[i].pairedScore = 0;
    int maxPaired1 = -1, maxPaired2 = -1, numPerfectPairs = 0;
    const int ilimit = n1 - 1, jlimit = n2 - 1, maxReadLen = imax(len1, len2);
    const int outerDistLimit = (maxReadLen * OUTER_DIST_MULT) / OUTER_DIST_DIV, innerDistLimit = S.maxPairDist;
    const int expectedFragLength = S.averagePairDist + len1 + len2;
    for (int i = 0, j = 0; i <= ilimit && j <= jlimit; i++) {
        Site &a = s1[i];
        while (j < jlimit && (s2[j].chrom < a.chrom || (s2[j].chrom == a.chrom && a.start - s2[j].stop > innerDistLimit))) j++;
        for (int k = j; k <= jlimit; k++) {
            Site &b = s2[k];
            if (b.chrom > a.chrom) break;
            if (b.start - a.stop > innerDistLimit) break;
            int innerdist, outerdist;
            if (a.strand != b.strand) {
                if (a.strand == 0) { innerdist = b.start - a.stop; outerdist = b.stop - a.start; }
                else { innerdist = a.start - b.stop; outerdist = a.stop - b.start; }
            } else if (a.start <= b.start) { innerdist = b.start - a.stop; outerdist = b.stop - a.start; }
            else { innerdist = a.start - b.stop; outerdist = a.stop - b.start; }
            if (outerdist >= outerDistLimit && innerdist <= innerDistLimit && a.strand != b.strand) {
                const int deviation = iabsdif(S.averagePairDist, innerdist);
                const int ps1 = a.score + 1 + imax(1, b.score / 2 - ((deviation * b.score) / (32 * expectedFragLength + 100)));
                const int ps2 = b.score + 1 + imax(1, a.score / 2 - ((deviation * a.score) / (32 * expectedFragLength + 100)));
                bool p1 = false, p2 = false;
                if (ps1 > a.pairedScore) { p1 = true; a.pairedScore = ps1; maxPaired1 = imax(a.score, maxPaired1); }
                if (ps2 > b.pairedScore) { p2 = true; b.pairedScore = ps2; maxPaired2 = imax(b.score, maxPaired2); }
                if (p1 && p2 && outerdist >= maxReadLen && deviation <= expectedFragLength && a.perfect && b.perfect) numPerfectPairs++;
            }
        }
    }
    for (int i = 0; i < n1; i++) if (s1[i].pairedScore > s1[i].score) s1[i].score = s1[i].pairedScore;
    for (int i = 0; i < n2; i++) if (s2[i].pairedScore > s2[i].score) s2[i].score = s2[i].pairedScore;
    if (S.trimList) {
        if (numPerfectPairs > 0) {
            n1 = trim_below_cutoff(s1, n1, (int)__fmul_rn((float)maxPaired1, .94f), false, 1, S.maxTrimSitesToRetain);
            n2 = trim_below_cutoff(s2, n2, (int)__fmul_rn((float)maxPaired2, .94f), false, 1, S.maxTrimSitesToRetain);
        } else {
            if (n1 > 4) n1 = trim_below_cutoff(s1, n1, (int)__fmul_rn((float)maxPaired1, .9f), true, 1, S.maxTrimSitesToRetain);
            if (n2 > 4) n2 = trim_below_cutoff(s2, n2, (int)__fmul_rn((float)maxPaired2, .9f), true, 1, S.maxTrimSitesToRetain);
        }
    }
}

// one thread per read (single) or per pair: BBMapThread.java:405-431 / :953-1017
__global__ __launch_bounds__(128) void begin_kernel(const Dev D) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (D.S.paired) {
        if (2 * u + 1 >= D.nreads) return;
        const long long r1 = 2 * u, r2 = r1 + 1;
        Site *s1 = D.ms + r1 * D.cap, *s2 = D.ms + r2 * D.cap;
        int n1 = load_sites(D, r1, s1), n2 = load_sites(D, r2, s2);
        if (n1 < 0 || n2 < 0) {                            // one mate's probe overflowed: the pair is reported, not mapped
            atomicAdd(&D.counters[3], (unsigned)((n1 < 0) + (n2 < 0)));
            D.mcount[r1] = n1 < 0 ? -1 : -2; D.mcount[r2] = n2 < 0 ? -1 : -2;
            return;
        }
        const int len1 = D.reads[r1].len, len2 = D.reads[r2].len;
        pair_initial(D.S, s1, n1, s2, n2, len1, len2);
        if (D.S.trimList) {
            if (n1 > MIN_TRIM_PAIRED) sort_sites<false>(s1, n1);
            if (n2 > MIN_TRIM_PAIRED) sort_sites<false>(s2, n2);
            trim_list(s1, n1, true, max_quality(D.S, len1), false, MIN_TRIM_PAIRED, D.S.maxTrimSitesToRetain);
            trim_list(s2, n2, true, max_quality(D.S, len2), false, MIN_TRIM_PAIRED, D.S.maxTrimSitesToRetain);
        }
        for (int i = 0; i < n1; i++) s1[i].score = s1[i].quickScore;
        for (int i = 0; i < n2; i++) s2[i].score = s2[i].quickScore;
        D.mcount[r1] = n1; D.mcount[r2] = n2;
        if (n1 == 0) atomicAdd(&D.counters[5], 1u);
        if (n2 == 0) atomicAdd(&D.counters[5], 1u);
    } else {
        if (u >= D.nreads) return;
        Site *s = D.ms + u * D.cap;
        int n = load_sites(D, u, s);
        if (n < 0) { atomicAdd(&D.counters[3], 1u); D.mcount[u] = -1; return; }
        if (D.S.trimList && n > 1) {
            sort_sites<false>(s, n);
            trim_list(s, n, false, max_quality(D.S, D.reads[u].len), true, MIN_TRIM_SINGLE, D.S.maxTrimSitesToRetain);
        }
        D.mcount[u] = n;
        if (n == 0) atomicAdd(&D.counters[5], 1u);
    }
}

// ---------------------------------------------------------------------------------------------- stage 2: scoreNoIndels + sort + tip deletions
// AbstractMapThread.scoreNoIndels (:762-856), Collections.sort, findTipDeletions (:1075-1105), and scoreSlow's opening
// (BBMapThread.java:255-260).  One thread per read.
__global__ __launch_bounds__(128) void score_kernel(const Dev D) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= D.nreads) return;
    SlowState st; st.idx = 0; st.phase = 3; st.minMsaLimit = 0; st.pending = -1; st.oldJob = -1; st.expectedLen = 0; st.minscore = 0; st.seq = 0;
    const int n = D.mcount[r];
    if (n <= 0) { D.slow[r] = st; return; }
    const bbidx_read rr = D.reads[r];
    const int len = rr.len, maxSw = max_quality(D.S, len), maxImp = max_imperfect(D.S, len);
    Site *s = D.ms + r * D.cap;
    int near = 0; bool force = false;
    for (int j = 0; j < n; j++) {
        Site &ss = s[j];                                   // in place: a site is 128 bytes, the loop touches a dozen of its fields
        const int oldScore = ss.score;
        const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
        if (ss.perfect) { near++; set_slow_score(ss, maxSw); ss.score = maxSw; ss.ngaps = 0; }
        else {
            const uint8_t *ref = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom];
            int sw = score_no_indels(D.S, bases, len, ref, reflen, ss.start);
            if (sw < oldScore && oldScore >= maxImp && ss.stop - ss.start + 1 != len) {            // :806-813
                const int sw2 = score_no_indels(D.S, bases, len, ref, reflen, ss.stop - len + 1);
                if (sw2 >= maxImp) { sw = sw2; set_start(ss, ss.stop - len + 1); set_perfect(ss, bases, len, ref, reflen); }
            }
            set_slow_score(ss, sw); ss.score = sw;
            if (sw >= maxImp) {
                near++;
                set_stop(ss, ss.start + len - 1); ss.ngaps = 0;
                if (sw >= maxSw) ss.perfect = ss.semiperfect = 1;
                else set_perfect(ss, bases, len, ref, reflen);
            } else if (oldScore >= maxImp) force = true;
        }
    }
    const int numNear = force ? -near : near;
    sort_sites<false>(s, n);
    if (numNear < 1 && D.S.tipSearchDist > 0) {
        for (int j = 0; j < n; j++) {
            if (!s[j].semiperfect && s[j].slowScore < maxImp) {
                Site ss = s[j];
                const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
                const uint8_t *ref = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom];
                if (find_tip_deletions(D.S, ss, bases, len, ref, reflen, maxImp, true, true)) {
                    ss.match_job = -1;
                    set_slow_score(ss, score_no_indels(D.S, bases, len, ref, reflen, ss.start));
                    if (ss.slowScore == maxSw) { set_stop(ss, ss.start + len - 1); ss.perfect = ss.semiperfect = 1; }
                    else { ss.perfect = 0; set_perfect(ss, bases, len, ref, reflen); }
                    s[j] = ss;
                }
            }
        }
    }
    D.nearArr[r] = numNear;
    if (D.S.paired || numNear < 1) {                       // single-ended: scoreSlow only without a near-perfect site (:466)
        st.phase = 0;
        st.minMsaLimit = -D.S.clearzone1e + (int)__fmul_rn(D.S.paired ? D.S.ratioPreRescue : D.S.minRatio, (float)maxSw);
    }
    D.slow[r] = st;
}

// ---------------------------------------------------------------------------------------------- stage 3: scoreSlow rounds
__device__ inline bbmsa_job make_job(const Dev &D, const bbidx_read &rr, const Site &ss, int pad, int minscore) {
    bbmsa_job j;
    j.read_off = rr.bases_off + (ss.strand ? D.minusDelta : 0);
    j.ref_off = (long long)(D.chromArr[ss.chrom] - D.refsBase);
    j.read_len = rr.len; j.ref_len = D.chromArrLen[ss.chrom];
    j.refStartLoc = ss.start - pad; j.refEndLoc = ss.stop + pad;
    j.minScore = minscore;
    j.flags = BBMSA_FILL_AND_SCORE_LIMITED | BBMSA_DO_TRACEBACK;
    return j;
}
// appends one fill to the plain or (sites with a gap array) the gapped log; returns its index (GAPPED_BIT marks the gapped log), or
// NO_ROOM when that log is full: the caller then leaves its state untouched and asks again in the next round, before which the host
// has grown the log (the reference's lists have no capacity; nothing may be lost or the batch refused because a log was sized too small)
constexpr int NO_ROOM = -2;
__device__ int emit_fill(const Dev &D, long long r, const bbidx_read &rr, const Site &ss, int site, int pad, int minscore, int kind, int seq) {
    bbmap_jobinfo info; info.read = (int)r; info.seq = seq; info.kind = kind; info.site = site;
    const bbmsa_job j = make_job(D, rr, ss, pad, minscore);
    // the wide list (second DP context: BBMap's 3000 columns) takes the sites with a gap array and the windows wider than the
    // first context's column limit; a job without gaps is an ordinary job there
    if (ss.ngaps || (imin(j.ref_len - 1, j.refEndLoc) - imax(0, j.refStartLoc) + 1) > D.plainColumns) {
        const unsigned k = atomicAdd(&D.counters[1], 1u);
        if ((long long)k >= D.gjobCap) return NO_ROOM;
        D.gjobs[k] = j; D.ginfo[k] = info;
        bbmsa_gaps g; g.ngaps = ss.ngaps;
        for (int q = 0; q < BBMSA_MAX_GAPS; q++) g.gaps[q] = q < ss.ngaps ? ss.gaps[q] : 0;
        D.ggaps[k] = g;
        return (int)k | GAPPED_BIT;
    }
    const unsigned k = atomicAdd(&D.counters[0], 1u);
    if ((long long)k >= D.jobCap) return NO_ROOM;
    D.jobs[k] = j; D.jinfo[k] = info;
    return (int)k;
}
__device__ inline const bbmsa_result &fill_result(const Dev &D, int job) {
    return (job & GAPPED_BIT) ? D.gresults[job & ~GAPPED_BIT] : D.results[job];
}
// the tail of scoreSlow's loop body (BBMapThread.java:361-381); job < 0: no (successful) fill
__device__ void finish_site(const Dev &D, SlowState &st, Site &ss, int job, const uint8_t *bases, int len, int maxSw) {
    if (job >= 0) {
        const bbmsa_result &res = fill_result(D, job);
        set_slow_score(ss, res.score[0]); set_limits(ss, res.score[1], res.score[2]); ss.match_job = job;
    }
    ss.reserved[0] = ss.reserved[1] = 0;
    ss.score = ss.slowScore;
    st.minMsaLimit = imax(st.minMsaLimit, ss.slowScore - D.S.clearzone3);
    ss.perfect = (ss.slowScore == maxSw);
    if (ss.perfect) ss.semiperfect = 1;
    else if (!ss.semiperfect) set_perfect(ss, bases, len, D.chromArr[ss.chrom], D.chromArrLen[ss.chrom]);
}

// scoreSlow's per-site opening (BBMapThread.java:278-303): a site whose span differs from the read length loses its ungapped
// score and flags; an over-long expected window is cut.  Depends on nothing but the site itself, so it gives the same answer
// whether it is evaluated ahead of time (on a copy) or when the loop reaches the site.
__device__ inline int prepare_site(const Settings &S, Site &ss, int len, int &expectedLen, bool &needsFill) {
    if (ss.stop - ss.start != len - 1) { set_slow_score(ss, 0); ss.semiperfect = 0; ss.perfect = 0; }
    const int sw = ss.slowScore;
    needsFill = sw < max_imperfect(S, len) && !ss.semiperfect;
    expectedLen = 0;
    if (needsFill) {
        expectedLen = calc_gref_len(ss);
        if (expectedLen >= S.expLimit) set_stop(ss, ss.start + imin(len + 40, S.expLimit));
    }
    return sw;
}
__device__ inline bbmap_jobinfo &job_info(const Dev &D, int job) { return (job & GAPPED_BIT) ? D.ginfo[job & ~GAPPED_BIT] : D.jinfo[job]; }

// One round: every active read consumes the result of its fill in flight and moves on to its next fill (or finishes).
// activeIn == nullptr: every read of the batch (round 1).
//
// Fills ahead of time.  Strictly one fill per read and round would make a read with m candidate sites take m rounds, and a
// round costs the latency of a whole DP launch sequence however few jobs it holds.  A site's fill depends on the sites before
// it through ONE number, the running minMsaLimit = max(initial, best slowScore so far - CLEARZONE3) (:376), and once a read's
// first site is done that number rarely moves (the list is sorted by score, best first).  So while site idx >= 1 is in
// flight, the sites behind it are filled too, with the limit as it stands (Site.reserved[0] = 1 + job, reserved[1] = the
// minScore used).  When the loop reaches such a site it recomputes the minScore the reference would use: equal -> the finished
// fill IS the reference's fill and is adopted (numbered in the read's sequence at that moment); different -> it is dropped
// (its log entry keeps seq = -1) and the site is filled again.  Results are those of the sequential loop in every case.
__global__ __launch_bounds__(128) void slow_round_kernel(const Dev D) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long count = D.activeIn ? D.nActiveIn : D.nreads;
    bool stillActive = false;
    long long r = -1;
    if (t < count) {
        r = D.activeIn ? D.activeIn[t] : t;
        SlowState st = D.slow[r];
        if (st.phase != 3) {
            const bbidx_read rr = D.reads[r];
            const int len = rr.len, maxSw = max_quality(D.S, len);
            const int n = D.mcount[r];
            Site *s = D.ms + r * D.cap;
            while (st.idx < n) {
                Site ss = s[st.idx];
                const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
                if (st.phase == 1) {                                               // first fill came back (:312-335)
                    const bbmsa_result &res = fill_result(D, st.pending);
                    const int nsc = res.score_len;
                    if (nsc > 6 && (res.score[3] + res.score[4] + st.expectedLen < D.S.expLimit)) {
                        set_limits(ss, ss.start - res.score[6], ss.stop + res.score[7]);
                        const int job = emit_fill(D, r, rr, ss, st.idx, D.S.slowAlignPadding + D.S.extraPadding, st.minscore, 1, st.seq);
                        stillActive = true;
                        if (job == NO_ROOM) break;                                     // log full: the same step again next round
                        st.seq++; st.oldJob = st.pending; st.pending = job;
                        atomicAdd(&D.counters[6], 1u);
                        st.phase = 2; s[st.idx] = ss;
                        break;
                    }
                    finish_site(D, st, ss, nsc > 0 ? st.pending : -1, bases, len, maxSw);
                    s[st.idx] = ss; st.idx++; st.phase = 0;
                    continue;
                }
                if (st.phase == 2) {                                               // the wider refill came back: keep the better one
                    const bbmsa_result &res = fill_result(D, st.pending), &old = fill_result(D, st.oldJob);
                    const int job = (res.score_len == 0 || res.score[0] < old.score[0]) ? st.oldJob : st.pending;
                    finish_site(D, st, ss, job, bases, len, maxSw);
                    s[st.idx] = ss; st.idx++; st.phase = 0;
                    continue;
                }
                // phase 0: look at site idx (:267-309)
                bool needsFill;
                const int early = ss.reserved[0], earlyMin = ss.reserved[1];
                const int sw = prepare_site(D.S, ss, len, st.expectedLen, needsFill);
                if (needsFill) {
                    st.minscore = imax(sw, st.minMsaLimit);
                    if (early && earlyMin == st.minscore) {                        // filled ahead of time with the very same bound: adopt
                        st.pending = early - 1;
                        bbmap_jobinfo &ji = job_info(D, st.pending);
                        ji.seq = st.seq++; ji.site = st.idx;
                        st.phase = 1; s[st.idx] = ss;
                        continue;                                                  // its result is there already
                    }
                    const int job = emit_fill(D, r, rr, ss, st.idx, D.S.slowAlignPadding, st.minscore, 0, st.seq);
                    stillActive = true;
                    if (job == NO_ROOM) break;                                         // log full: this site again next round
                    if (early) atomicAdd(&D.counters[8], 1u);                      // a fill ahead of time that the sequence does not contain
                    st.seq++; st.pending = job;
                    st.phase = 1; s[st.idx] = ss;
                    break;
                }
                if (early) atomicAdd(&D.counters[8], 1u);
                finish_site(D, st, ss, -1, bases, len, maxSw);
                s[st.idx] = ss; st.idx++;
            }
            if (!stillActive) st.phase = 3;
            else if (st.idx >= 1 && D.fillAhead && (st.phase == 1 || st.phase == 2)) {
                for (int j = st.idx + 1; j < n; j++) {
                    Site tmp = s[j];
                    bool needsFill; int expectedLen;
                    const int sw = prepare_site(D.S, tmp, len, expectedLen, needsFill);
                    if (!needsFill) continue;
                    const int minscore = imax(sw, st.minMsaLimit);
                    if (tmp.reserved[0] && tmp.reserved[1] == minscore) continue;  // already in the log with this bound
                    const int job = emit_fill(D, r, rr, tmp, j, D.S.slowAlignPadding, minscore, 0, -1);
                    if (job == NO_ROOM) break;                                         // no room for fills ahead of time this round
                    if (tmp.reserved[0]) atomicAdd(&D.counters[8], 1u);
                    s[j].reserved[0] = job + 1; s[j].reserved[1] = minscore;
                }
            }
            D.slow[r] = st;
        }
    }
    // next round's read list: one reservation per wavefront
    const unsigned long long m = __ballot(stillActive);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&D.counters[2], (unsigned)__builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (stillActive) D.activeOut[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)r;
    }
}

// ---------------------------------------------------------------------------------------------- stage 4: after scoreSlow
// Tools.mergeDuplicateSites (+ Collections.sort for single-ended reads, BBMapThread.java:483-489 / :1042, :1058)
__global__ __launch_bounds__(128) void finish_kernel(const Dev D) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= D.nreads) return;
    int n = D.mcount[r];
    if (n <= 0) return;
    Site *s = D.ms + r * D.cap;
    n = merge_duplicate_sites(s, n);
    if (!D.S.paired) sort_sites<false>(s, n);
    D.mcount[r] = n;
}

// ---------------------------------------------------------------------------------------------- stage 5: rescue
// processReadPair :1065-1095 + rescue() up to the quickRescue call (AbstractMapThread.java:1144-1220).  One thread per pair.
__global__ __launch_bounds__(128) void rescue_plan_kernel(const Dev D) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p + 1 >= D.nreads) return;
    PairResc pr = D.pres[p];
    pr.first = 0; pr.count = 0; pr.ran = 0;
    const long long ra = 2 * p + D.pass, rl = 2 * p + (1 - D.pass);             // anchor read, loose read
    int na = D.mcount[ra], nl = D.mcount[rl];
    if (na < 0 || nl < 0) { D.pres[p] = pr; return; }                            // overflowed pair
    Site *sa = D.ms + ra * D.cap;
    const Site *sl = D.ms + rl * D.cap;
    int unpairedA;
    if (D.pass == 0) {
        int u1 = 0, u2 = 0;
        for (int i = 0; i < na; i++) if (sa[i].pairedScore == 0) u1++;
        for (int i = 0; i < nl; i++) if (sl[i].pairedScore == 0) u2++;
        pr.unpaired2 = u2; unpairedA = u1;
    } else unpairedA = pr.unpaired2;
    if (!(unpairedA > 0 && na > 0)) { D.pres[p] = pr; return; }
    pr.ran = 1;
    const int lenA = D.reads[ra].len, L = D.reads[rl].len;
    sort_sites<false>(sa, na);
    na = remove_low_quality_paired(sa, na, max_quality(D.S, lenA), D.S.ratioPreRescue, D.S.ratioPreRescue);
    D.mcount[ra] = na;
    const int searchDist = imin(D.S.maxPairDist, 2 * D.S.averagePairDist + 100);
    if (searchDist > D.S.maxRescueDist || na == 0) { D.pres[p] = pr; return; }
    const int maxLooseSw = max_quality(D.S, L), maxAnchorSw = max_quality(D.S, lenA), maxImp = max_imperfect(D.S, L);
    const int bestLoose = nl == 0 ? 0 : sl[0].slowScore, bestAnchor = sa[0].slowScore;
    if (bestLoose == maxLooseSw && bestAnchor == maxAnchorSw && sa[0].pairedScore > 0) { D.pres[p] = pr; return; }
    const int rescueScoreLimit = (int)__fmul_rn(0.95f, (float)bestAnchor);
    pr.retainLimit = imax((int)__fmul_rn(0.68f, (float)bestLoose), (int)__fmul_rn(0.4f, (float)maxLooseSw));
    pr.retainLimit2 = imax((int)__fmul_rn(0.95f, (float)bestLoose), (int)__fmul_rn(0.55f, (float)maxLooseSw));
    pr.maxMismatches = bestLoose > maxImp ? 5 : imin(D.S.maxRescueMismatches, (int)__fsub_rn(__fmul_rn(0.60f, (float)L), 1.0f));
    pr.findTip = (D.S.tipSearchDist > 0 && bestLoose < maxImp) ? 1 : 0;
    int cnt = 0;
    for (int i = 0; i < na; i++) { if (sa[i].slowScore < rescueScoreLimit) break; if (sa[i].pairedScore == 0 && !sa[i].rescued) cnt++; }
    if (cnt == 0) { D.pres[p] = pr; return; }
    const unsigned first = atomicAdd(&D.counters[4], (unsigned)cnt);
    pr.first = (int)first; pr.count = cnt;
    if ((long long)first + cnt <= D.rescCap) {
        const bbidx_read rrl = D.reads[rl];
        int k = 0;
        for (int i = 0; i < na; i++) {
            const Site &ssa = sa[i];
            if (ssa.slowScore < rescueScoreLimit) break;
            if (!(ssa.pairedScore == 0 && !ssa.rescued)) continue;
            const int searchIntoAnchor = ssa.stop - ssa.start - 1 + (lenA * 11 / 16);
            const int strand = ssa.strand ^ 1;
            bbresc_job j;
            j.read_off = rrl.bases_off + (strand ? D.minusDelta : 0);              // the loose read on the strand to search
            j.read_len = L; j.chrom = ssa.chrom;
            if (ssa.strand == 0) { j.loc = ssa.stop - searchIntoAnchor; j.idealStart = ssa.stop + D.S.averagePairDist; }
            else { j.loc = ssa.start + searchIntoAnchor; j.idealStart = ssa.start - D.S.averagePairDist; }
            j.searchDist = searchDist + searchIntoAnchor; j.maxAllowedMismatches = pr.maxMismatches;
            j.flags = strand == 1 ? 1 : 0; j.reserved = 0;
            D.rjobs[first + k] = j;
            RescInfo ri; ri.pair = (int)p; ri.anchorSite = i; ri.strand = strand; ri.job = -1;
            D.rinfo[first + k] = ri;
            k++;
        }
    }
    D.pres[p] = pr;
}

// rescue()'s body after quickRescue + slowRescue up to its fill (AbstractMapThread.java:1222-1226, :1246-1265).  One thread per
// pair, its searches in anchor order (a read's fills are numbered in the order the reference issues them).
__global__ __launch_bounds__(128) void rescue_prep_kernel(const Dev D) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p + 1 >= D.nreads) return;
    const PairResc pr = D.pres[p];
    if (pr.count == 0 || (long long)pr.first + pr.count > D.rescCap) return;
    const long long rl = 2 * p + (1 - D.pass);
    const bbidx_read rrl = D.reads[rl];
    const int L = rrl.len, maxImp = max_imperfect(D.S, L), maxScore = max_quality(D.S, L);
    int seq = D.slow[rl].seq;
    for (int k = 0; k < pr.count; k++) {
        const long long q = pr.first + k;
        RescInfo ri = D.rinfo[q];
        const bbresc_result res = D.rres[q];
        const bbresc_job rj = D.rjobs[q];
        Site ss;
        ss.chrom = rj.chrom; ss.strand = ri.strand; ss.start = res.start; ss.stop = res.stop; ss.hits = 0;
        ss.quickScore = ss.score = res.score; ss.slowScore = 0; ss.pairedScore = 0;
        ss.perfect = res.perfect; ss.semiperfect = res.semiperfect; ss.rescued = 1; ss.ngaps = 0;
        for (int i = 0; i < BBMSA_MAX_GAPS; i++) ss.gaps[i] = 0;
        ss.match_job = -1; ss.reserved[0] = 0; ss.reserved[1] = 0;
        ri.job = -1;
        const int reflen = D.chromArrLen[rj.chrom];
        // reserved[0]: 0 = dropped, 1 = slowRescue finished without a fill, 2 = fill in flight; reserved[1] = ungapped score
        if (res.found == 1 && ss.start >= 0 && ss.stop <= reflen - 1 && res.mismatches <= pr.maxMismatches) {
            const uint8_t *bases = D.bases + rj.read_off;
            const uint8_t *ref = D.chromArr[rj.chrom];
            int sw = score_no_indels(D.S, bases, L, ref, reflen, ss.start);
            if (sw < maxImp && D.S.maxIndel > 0) {
                set_slow_score(ss, sw);
                if (pr.findTip && find_tip_deletions(D.S, ss, bases, L, ref, reflen, maxImp, true, true)) sw = score_no_indels(D.S, bases, L, ref, reflen, ss.start);
                const int minMsaLimit = -D.S.clearzone1e + (int)__fmul_rn(D.S.ratioPaired, (float)maxScore);
                ri.job = emit_fill(D, rl, rrl, ss, -1, D.S.slowRescuePadding, imax(sw, minMsaLimit), 2, seq++);
                atomicAdd(&D.counters[7], 1u);
                ss.reserved[0] = 2; ss.reserved[1] = sw;
            } else {
                set_slow_score(ss, sw); ss.score = ss.slowScore; set_stop(ss, ss.start + L - 1);
                ss.reserved[0] = 1; ss.reserved[1] = sw;
            }
        }
        D.rsite[q] = ss;
        D.rinfo[q] = ri;
    }
    D.slow[rl].seq = seq;
}

// slowRescue's tail (:1267-1305), rescue()'s retain / pair logic (:1227-1236) and mergeDuplicateSites of the loose list
// (BBMapThread.java:1087 / :1094).  One thread per pair.
__global__ __launch_bounds__(128) void rescue_finish_kernel(const Dev D) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p + 1 >= D.nreads) return;
    const PairResc pr = D.pres[p];
    if (!pr.ran) return;
    const long long ra = 2 * p + D.pass, rl = 2 * p + (1 - D.pass);
    Site *sa = D.ms + ra * D.cap, *sl = D.ms + rl * D.cap;
    int nl = D.mcount[rl];
    const int nsearch = ((long long)pr.first + pr.count > D.rescCap) ? 0 : pr.count;
    const bbidx_read rrl = D.reads[rl];
    const int L = rrl.len, maxScore = max_quality(D.S, L);
    bool overflow = false;
    for (int k = 0; k < nsearch; k++) {
        const long long q = pr.first + k;
        Site ss = D.rsite[q];
        if (ss.reserved[0] == 0) continue;
        const RescInfo ri = D.rinfo[q];
        const uint8_t *bases = D.bases + D.rjobs[q].read_off;
        const uint8_t *ref = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom];
        if (ss.reserved[0] == 2) {
            const bbmsa_result &res = fill_result(D, ri.job);
            if (res.score_len > 0) { set_slow_score(ss, res.score[0]); ss.score = ss.slowScore; set_start(ss, res.score[1]); set_stop(ss, res.score[2]); ss.match_job = ri.job; }
            else { const int oldStart = D.rres[q].start; set_slow_score(ss, ss.reserved[1]); ss.score = ss.slowScore; set_start(ss, oldStart); set_stop(ss, ss.start + L - 1); }
        }
        ss.reserved[0] = ss.reserved[1] = 0;
        ss.pairedScore = ss.score + 1;
        ss.perfect = (ss.slowScore == maxScore);
        if (ss.perfect) ss.semiperfect = 1; else set_perfect(ss, bases, L, ref, reflen);
        if (ss.score > pr.retainLimit && ss.start >= 0 && ss.stop <= reflen - 1) {
            Site &ssa = sa[ri.anchorSite];
            if (ss.score > pr.retainLimit2) {
                ss.pairedScore = imax(ss.pairedScore, ss.slowScore + ssa.slowScore / 4);
                ssa.pairedScore = imax(ssa.pairedScore, ssa.slowScore + ss.slowScore / 4);
            }
            if (nl < D.cap) sl[nl++] = ss; else overflow = true;
        }
    }
    if (overflow) { atomicAdd(&D.counters[3], 1u); D.mcount[rl] = -1; return; }
    D.mcount[rl] = merge_duplicate_sites(sl, nl);
}

#include "mapper_final.h"

// ---------------------------------------------------------------------------------------------- overflow tier
// units (reads, or pairs in paired mode) whose site list did not fit: appended in any order, sorted on the host
__global__ __launch_bounds__(128) void collect_overflow_kernel(const int *mcount, long long nunits, int paired, int *ids, unsigned *count) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nunits) return;
    const bool over = paired ? (mcount[2 * u] < 0 || mcount[2 * u + 1] < 0) : mcount[u] < 0;
    if (over) ids[atomicAdd(count, 1u)] = (int)u;
}
__global__ __launch_bounds__(128) void gather_reads_kernel(const bbidx_read *reads, const int *ids, int nunits, int paired, bbidx_read *sub, int *readIds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nunits) return;
    if (paired) {
        const int r = 2 * ids[i];
        sub[2 * i] = reads[r]; sub[2 * i + 1] = reads[r + 1]; readIds[2 * i] = r; readIds[2 * i + 1] = r + 1;
    } else { sub[i] = reads[ids[i]]; readIds[i] = ids[i]; }
}
// a read the tier mapped is marked in the main list (BBMAP_NSITES_IN_TIER); counts the reads that had overflowed and now have a list
__global__ __launch_bounds__(128) void mark_tier_kernel(int *mcount, const int *tierCount, const int *readIds, int n, unsigned *resolved) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || tierCount[i] < 0) return;
    const int r = readIds[i];
    if (mcount[r] == -1) atomicAdd(resolved, 1u);
    mcount[r] = BBMAP_NSITES_IN_TIER;
}

// ---------------------------------------------------------------------------------------------- packed output
__global__ __launch_bounds__(256) void pack_counts_kernel(const int *mcount, long long n, int *counts) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { const int m = mcount[r]; counts[r] = m > 0 ? m : 0; }
}
// one 16-byte piece of a 128-byte site record per thread: eight consecutive lanes move one record
__global__ __launch_bounds__(256) void pack_sites_kernel(const Site *ms, const int *mcount, const long long *offsets, long long n, int cap, long long packedCap, Site *packed) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long slot = t >> 3; const int piece = (int)(t & 7);
    const long long r = slot / cap; const int j = (int)(slot - r * cap);
    if (r >= n || j >= mcount[r]) return;
    const long long dst = offsets[r] + j;
    if (dst >= packedCap) return;
    ((uint4 *)(packed + dst))[piece] = ((const uint4 *)(ms + r * cap + j))[piece];
}

struct ToLL { __host__ __device__ long long operator()(int x) const { return (long long)x; } };

}  // namespace bbmapper

// ================================================================================================= host side
struct bbmap_ctx {
    bbmap_config cfg;
    bbidx_ctx *index;
    hipStream_t hostStream = nullptr;   // bbmap_map_batch (host buffers in and out) runs on it
    bbidx_launch probeLs;       // this context's probe launches: queue, work counters, events (the index is shared, read-only; index_ctx.h)
    bbmsa_ctx *msa, *msaGapped;
    bbmapper::Settings S;
    std::vector<void *> allocs;
    // device buffers
    bbidx_site *d_psites; int *d_pnsites;
    bbmap_msite *d_ms; int *d_mcount, *d_near;
    bbmapper::SlowState *d_slow;
    int *d_active[2];
    unsigned *d_counters;
    bbmsa_job *d_jobs; bbmap_jobinfo *d_jinfo; bbmsa_result *d_results; uint8_t *d_match;
    bbmsa_job *d_gjobs; bbmsa_gaps *d_ggaps; bbmap_jobinfo *d_ginfo; bbmsa_result *d_gresults; uint8_t *d_gmatch;
    bbresc_job *d_rjobs; bbmapper::RescInfo *d_rinfo; bbresc_result *d_rres; bbmapper::PairResc *d_pres; bbmap_msite *d_rsite;
    const uint8_t *const *d_chromArr; const int *d_chromArrLen; const uint8_t *refsBase;
    int *d_chromMin;
    long long *d_chromOff;
    long long jobCap, gjobCap, rescCap;
    bbmapper::FinalRead *d_fin; bbmap_final *d_final; uint8_t *d_pool; long long poolUnits, poolUsed, finalFills;
    int matchStride, gmatchStride, maxRows, plainColumns;
    unsigned *h_counters;           // pinned
    hipEvent_t ev[12];
    bbmap_stats stats;
    long long nJobs, nGapped;
    bool ran;
    // overflow tier: a second, small context with long site lists for the reads whose list did not fit max_sites
    void *d_packTmp; size_t packTmpBytes;     // bbmap_pack_sites_device's scan scratch (allocated on first use)
    bbmap_ctx *tier;
    bool ownsMsa;
    int *d_tierUnits; bbidx_read *d_tierReads; int *d_tierReadIds;
    long long tierReads;            // reads the tier mapped in the last batch
    // The tier's pass runs beside the main pass (its reads are known once begin_kernel has run): its own stream, driven by its
    // own host thread, joined at the end of the batch.
    hipStream_t tierStream;
    // second-context fills (few jobs, wide windows: a handful of waves per CU) run on a stream of their own beside the plain ones
    hipStream_t dpStream; hipEvent_t evFork, evJoin;
    std::thread tierThread;
    bool tierStarted;
    int tierRc; char tierErr[320];
    long long overAfterBegin;       // reads flagged by the probe (counters[3] after begin_kernel)
    struct BatchArgs { int64_t n_reads; const bbidx_read *reads; uint8_t *bases; int64_t minus_delta; const int8_t *baseScores; const int32_t *keyinfo; } batch;
    // bbmap_map_batch (host buffers in, packed site lists out): device copies the context keeps between calls, grown on demand
    struct HostIO { void *p[7]; size_t cap[7]; } hio;      // reads, bases (both strands), base scores, keyinfo, counts, offsets, packed
};

static thread_local char g_merr[320];
#define MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_merr, sizeof g_merr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_merr); return BBMAP_E_HIP; } } while (0)
#define MTRY(expr) do { const int rc_ = (expr); if (rc_ != BBMAP_OK) return rc_; } while (0)
static int mfail(int code, const char *msg) { bbmap_set_error(msg); return code; }

extern "C" int bbmap_default_config_profile(int32_t profile, bbmap_config *c) {
    if (!c || (profile != BBIDX_PROFILE_BBMAP && profile != BBIDX_PROFILE_PACBIO)) return mfail(BBMAP_E_ARG, "bbmap_default_config: bad argument");
    memset(c, 0, sizeof *c);
    c->paired = 0; c->max_reads = 0; c->max_sites = 32;
    c->extraPadding = 10; c->maxPairDist = 32000; c->averagePairDist = 100; c->maxRescueDist = 1200; c->maxRescueMismatches = 32;
    c->maxTrimSitesToRetain = 800; c->trimList = 1; c->doRescue = 1; c->clearzone3 = 800; c->fastCols = 0; c->jobsPerRead = 0;
    c->finalStage = profile == BBIDX_PROFILE_BBMAP ? 1 : 0;
    if (profile == BBIDX_PROFILE_PACBIO) {      // BBMapPacBio.setDefaults (BBMapPacBio.java:47-69), BBMapThreadPacBio.java:27-28
        c->max_read_len = 6016; c->minRatio = 0.46f; c->slowAlignPadding = 8; c->slowRescuePadding = 16; c->tipSearchDist = 15;
        c->alignColumns = 7600; c->msaMaxColumns = 7600;
    } else {                                    // BBMap.setDefaults (BBMap.java:45-65), BBMapThread.java:27-28
        c->max_read_len = 150; c->minRatio = 0.56f; c->slowAlignPadding = 4; c->slowRescuePadding = 8; c->tipSearchDist = 100;
        c->alignColumns = 3000; c->msaMaxColumns = 3000;
    }
    c->reserved[3] = profile;
    return BBMAP_OK;
}
extern "C" int bbmap_default_config(bbmap_config *c) { return bbmap_default_config_profile(BBIDX_PROFILE_BBMAP, c); }

template <class T> static int dalloc(bbmap_ctx *c, T **p, size_t count) {
    void *d = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&d, bytes) != hipSuccess) return mfail(BBMAP_E_NOMEM, "bbmap_create: device allocation failed");
    c->allocs.push_back(d);
    *p = (T *)d;
    return BBMAP_OK;
}

extern "C" void bbmap_destroy(bbmap_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->tierThread.joinable()) c->tierThread.join();
    if (c->tier) bbmap_destroy(c->tier);
    bbidx_launch_free(&c->probeLs);
    if (c->d_packTmp) (void)hipFree(c->d_packTmp);
    for (int i = 0; i < 7; i++) if (c->hio.p[i]) (void)hipFree(c->hio.p[i]);
    if (c->hostStream) (void)hipStreamDestroy(c->hostStream);
    if (c->tierStream) (void)hipStreamDestroy(c->tierStream);
    if (c->dpStream) (void)hipStreamDestroy(c->dpStream);
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    if (c->evJoin) (void)hipEventDestroy(c->evJoin);
    if (c->ownsMsa && c->msaGapped && c->msaGapped != c->msa) bbmsa_destroy(c->msaGapped);
    if (c->ownsMsa && c->msa) bbmsa_destroy(c->msa);
    for (int i = 0; i < 12; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete c;
}

extern "C" int bbidx_get_chrom_table(bbidx_ctx *ix, int32_t *nchroms, const uint8_t **chromArr, int32_t *chromArrLen, int32_t cap) {
    if (!ix || !nchroms) return mfail(BBMAP_E_ARG, "bbidx_get_chrom_table: null argument");
    *nchroms = ix->dev.nchroms;
    if (!chromArr && !chromArrLen) return BBMAP_OK;
    if (cap < ix->dev.nchroms + 1) return mfail(BBMAP_E_ARG, "bbidx_get_chrom_table: buffers too small (need nchroms + 1 entries)");
    MHIP(hipSetDevice(ix->device));
    if (chromArr) MHIP(hipMemcpy(chromArr, ix->dev.chromArr, sizeof(void *) * (size_t)(ix->dev.nchroms + 1), hipMemcpyDeviceToHost));
    if (chromArrLen) MHIP(hipMemcpy(chromArrLen, ix->dev.chromArrLen, 4 * (size_t)(ix->dev.nchroms + 1), hipMemcpyDeviceToHost));
    return BBMAP_OK;
}

// parent != null: the overflow tier of `parent` (longer job logs per read)
static int create_impl(bbidx_ctx *index, const bbmap_config *cfg, bbmap_ctx *parent, bbmap_ctx **out) {
    if (!index || !cfg || !out) return mfail(BBMAP_E_ARG, "bbmap_create: null argument");
    *out = nullptr;
    const int profile = cfg->reserved[3];
    if (profile != BBIDX_PROFILE_BBMAP && profile != BBIDX_PROFILE_PACBIO) return mfail(BBMAP_E_ARG, "bbmap_create: unknown profile (bbmap_config.reserved[3])");
    if (profile != index->dev.p.profile) return mfail(BBMAP_E_ARG, "bbmap_create: the index was built for the other profile (BBIDX_PROFILE_*)");
    const bool pacbio = profile == BBIDX_PROFILE_PACBIO;
    if (cfg->max_reads < 1 || cfg->max_read_len < 1 || cfg->max_read_len > (pacbio ? BBIDX_PACBIO_MAX_READ_LEN : 600))
        return mfail(BBMAP_E_ARG, "bbmap_create: max_reads >= 1 and max_read_len in 1..600 (1..6016 for BBIDX_PROFILE_PACBIO)");
    if (cfg->max_sites < 1 || cfg->max_sites > BBMAP_MAX_SITES_LIMIT) return mfail(BBMAP_E_ARG, "bbmap_create: max_sites must be 1..4096");
    if (cfg->paired && (cfg->max_reads & 1)) return mfail(BBMAP_E_ARG, "bbmap_create: paired mode takes an even number of reads");
    if (cfg->msaMaxColumns < 64 || cfg->msaMaxColumns > (pacbio ? 8192 : 4096)) return mfail(BBMAP_E_ARG, "bbmap_create: msaMaxColumns must be 64..4096 (..8192 for BBIDX_PROFILE_PACBIO)");
    if (cfg->device != index->device) return mfail(BBMAP_E_ARG, "bbmap_create: the index lives on another device");
    MHIP(hipSetDevice(cfg->device));
    bbmap_ctx *c = new (std::nothrow) bbmap_ctx();
    if (!c) return mfail(BBMAP_E_NOMEM, "bbmap_create: out of host memory");
    c->cfg = *cfg; c->index = index;
    int rc = BBMAP_OK;
    auto bail = [&](int code) { bbmap_destroy(c); return code; };
    if ((rc = bbidx_launch_init(index, &c->probeLs)) != BBMAP_OK) return bail(rc);
    // settings
    bbmapper::Settings &S = c->S;
    const float R = cfg->minRatio;
    S.minRatio = R;
    { const float a = R * .80f, b = 1.0f - ((1.0f - R) * 1.4f); S.ratioPaired = a > b ? a : b; }                // AbstractMapThread.java:106
    { const float a = R * .60f, b = 1.0f - ((1.0f - R) * 1.8f); S.ratioPreRescue = a > b ? a : b; }             // :107
    S.slowAlignPadding = cfg->slowAlignPadding; S.slowRescuePadding = cfg->slowRescuePadding; S.extraPadding = cfg->extraPadding;
    S.tipSearchDist = cfg->tipSearchDist; S.maxPairDist = cfg->maxPairDist; S.averagePairDist = cfg->averagePairDist;
    S.maxRescueDist = cfg->maxRescueDist; S.maxRescueMismatches = cfg->maxRescueMismatches; S.maxTrimSitesToRetain = cfg->maxTrimSitesToRetain;
    S.trimList = cfg->trimList; S.doRescue = cfg->doRescue; S.alignColumns = cfg->alignColumns; S.clearzone3 = cfg->clearzone3;
    S.maxIndel = index->dev.p.maxIndel; S.paired = cfg->paired;
    if (pacbio) { S.ptsMatch = 90; S.ptsMatch2 = 100; S.ptsSub = -137; S.ptsSub2 = -49; S.ptsSub3 = -25; S.impDelta = -305; }   // min(-292, -205 - 100)
    else { S.ptsMatch = 70; S.ptsMatch2 = 100; S.ptsSub = -127; S.ptsSub2 = -51; S.ptsSub3 = -25; S.impDelta = -495; }          // min(-472, -395 - 100)
    S.clearzone1e = 2 * S.ptsMatch2 - S.ptsMatch - S.ptsSub + 1;
    S.msaMaxColumns = cfg->msaMaxColumns;
    if (cfg->finalStage && pacbio) return bail(mfail(BBMAP_E_ARG, "bbmap_create: the final alignment stage follows BBMapThread (BBIDX_PROFILE_BBMAP) only; set finalStage = 0 for BBIDX_PROFILE_PACBIO"));
    S.finalStage = cfg->finalStage ? 1 : 0;
    // BBMap.java:434: `if(paired){BBIndex.QUIT_AFTER_TWO_PERFECTS=false;}` -- a static of the index class in the reference, so the
    // borrowed index context is switched the same way (and back for a single-ended mapper)
    index->dev.p.quitAfterTwoPerfects = cfg->paired ? 0 : 1;
    S.expLimit = (cfg->alignColumns * 17) / 20 - (2 * (cfg->slowAlignPadding + 10));                            // EXPECTED_LEN_LIMIT, :92
    // DP contexts: the plain one takes every ungapped window (first pass for the common narrow ones, the wide pass for the rest)
    const int maxRows = ((cfg->max_read_len + 31) / 32) * 32;
    c->maxRows = maxRows;
    bbmsa_config mc; memset(&mc, 0, sizeof mc);
    // two DP contexts.  The first takes the ordinary windows (read length + a few dozen columns): its LDS tables and column
    // buffers are sized for `fastCols` columns, which is what lets four blocks share a CU.  The second has the reference's own
    // 3000 columns (BBMapThread.java:27-28) and takes what does not fit the first: gapped references and wide windows.
    mc.device = cfg->device; mc.maxRows = maxRows;
    bbmsa_config gc;
    if (pacbio) {
        // mapPacBio: ONE context with the MultiStateAligner9PacBio scheme (strip-tiled wavefront kernel, msa_fill_strip.hip) and the
        // reference's 7600 columns for every fill, with or without a gap array; its traceback records and scratch matrices take tens
        // of GB, so the overflow tier borrows its parent's context and runs after the main pass instead of beside it
        mc.maxRows = cfg->max_read_len + 4 > 6100 ? 6100 : cfg->max_read_len + 4;
        c->maxRows = mc.maxRows;
        mc.maxColumns = cfg->msaMaxColumns;
        mc.reserved[2] = BBMSA_SCHEME_9PACBIO;
        gc = mc;
        c->plainColumns = mc.maxColumns;
        if (parent) { c->msa = parent->msa; c->msaGapped = parent->msaGapped; c->ownsMsa = false; }
        else {
            c->ownsMsa = true;
            if ((rc = bbmsa_create(&mc, &c->msa)) != BBMAP_OK) return bail(rc);
            c->msaGapped = c->msa;
        }
    } else {
        mc.maxColumns = cfg->fastCols > 0 ? cfg->fastCols : 256;
        if (mc.maxColumns > cfg->msaMaxColumns) mc.maxColumns = cfg->msaMaxColumns;
        c->plainColumns = mc.maxColumns;
        gc = mc;
        gc.maxColumns = cfg->msaMaxColumns;
        gc.reserved[0] = 32; gc.reserved[1] = 640 < gc.maxColumns ? 640 : gc.maxColumns;      // (32 lanes x 5 rows per job: 80 vs 85 ms of scoreSlow with sh/randomreads.sh's deletions; bbmsa_create widens the group for longer reads)
        if (const char *e = getenv("BBMAP_G2_LANES")) { if (*e) gc.reserved[0] = atoi(e); }          // experiments: geometry of the second context
        if (const char *e = getenv("BBMAP_G2_COLS")) { if (*e) gc.reserved[1] = atoi(e) < gc.maxColumns ? atoi(e) : gc.maxColumns; }
        (void)parent;                   // the tier runs beside its parent's pass: DP contexts of its own
        c->ownsMsa = true;
        if ((rc = bbmsa_create(&mc, &c->msa)) != BBMAP_OK) return bail(rc);
        if ((rc = bbmsa_create(&gc, &c->msaGapped)) != BBMAP_OK) return bail(rc);
    }
    const long long n = cfg->max_reads;
    const int cap = cfg->max_sites;
    if (!pacbio) {
        // launches of a few hundred fills (the late rounds of scoreSlow and of the final stage) are one wavefront's latency: they take
        // the 64-lane geometry, whose step is the shorter chain (msa_ctx.h; 236 -> 231 ms per step for the second context alone)
        const long long lat = getenv("BBMAP_LATENCY_JOBS") ? atoll(getenv("BBMAP_LATENCY_JOBS")) : 4096;
        if ((rc = bbmsa_set_latency_jobs(c->msa, lat)) != BBMAP_OK) return bail(rc);
        if (c->msaGapped != c->msa && (rc = bbmsa_set_latency_jobs(c->msaGapped, lat)) != BBMAP_OK) return bail(rc);
    }
    // starting capacities of the two fill logs; they grow when a batch needs more (grow_logs)
    const int jpr = cfg->jobsPerRead > 0 ? cfg->jobsPerRead : 3;
    c->jobCap = n * jpr + 1024;
    c->gjobCap = parent ? n * 16 + 4096 : (n * jpr) / 24 + 4096;
    if (cfg->jobsPerRead < 0) c->jobCap = c->gjobCap = -(long long)cfg->jobsPerRead;       // exact starting capacity (tests of the growth path)
    c->rescCap = parent ? n * 64 + 1024 : n * 2 + 1024;
    c->matchStride = ((maxRows + c->plainColumns + 15) / 16) * 16;
    // a gapped match string expands every gap symbol to 128 'D's (traceback, MultiStateAligner11tsJNI.java:481-493)
    c->gmatchStride = ((maxRows + gc.maxColumns + 2 + 128 * 8 + 15) / 16) * 16;
    // the plain log rarely needs more than rows + columns of a NARROW window: cap its slot at what first-pass windows need, and
    // let the rare wide window report match_len = -1?  No: slots are sized for the widest window the context accepts.
#define DA(ptr, count) if ((rc = dalloc(c, &(ptr), (size_t)(count))) != BBMAP_OK) return bail(rc)
    DA(c->d_psites, n * cap); DA(c->d_pnsites, n);
    DA(c->d_ms, n * cap); DA(c->d_mcount, n); DA(c->d_near, n);
    DA(c->d_slow, n);
    DA(c->d_active[0], n); DA(c->d_active[1], n);
    DA(c->d_counters, 64);
    DA(c->d_jobs, c->jobCap); DA(c->d_jinfo, c->jobCap); DA(c->d_results, c->jobCap); DA(c->d_match, c->jobCap * c->matchStride);
    DA(c->d_gjobs, c->gjobCap); DA(c->d_ggaps, c->gjobCap); DA(c->d_ginfo, c->gjobCap); DA(c->d_gresults, c->gjobCap); DA(c->d_gmatch, c->gjobCap * c->gmatchStride);
    DA(c->d_rjobs, c->rescCap); DA(c->d_rinfo, c->rescCap); DA(c->d_rres, c->rescCap); DA(c->d_rsite, c->rescCap);
    DA(c->d_pres, n / 2 + 1);
    if (S.finalStage) {
        // match strings of the final stage: one of the read's length per perfect read, about two per imperfect one; grows on demand
        c->poolUnits = (n * (long long)(3 * (cfg->max_read_len + 16)) + 65536) / 4;
        if (const char *e = getenv("BBMAP_FINAL_POOL_UNITS")) { if (*e && atoll(e) >= 64) c->poolUnits = atoll(e); }      // (tests of the growth path)
        DA(c->d_fin, n); DA(c->d_final, n); DA(c->d_pool, c->poolUnits * 4);
    }
    const int nch = index->dev.nchroms;
    DA(c->d_chromMin, nch + 1); DA(c->d_chromOff, nch + 1);
#undef DA
    c->d_chromArr = index->dev.chromArr; c->d_chromArrLen = index->dev.chromArrLen;
    {
        std::vector<const uint8_t *> hc((size_t)nch + 1);
        if (hipMemcpy(hc.data(), index->dev.chromArr, sizeof(void *) * hc.size(), hipMemcpyDeviceToHost) != hipSuccess) return bail(mfail(BBMAP_E_HIP, "bbmap_create: reading the chromosome table failed"));
        c->refsBase = hc[1];
        std::vector<long long> off((size_t)nch + 1, 0);
        for (int i = 1; i <= nch; i++) off[(size_t)i] = (long long)(hc[(size_t)i] - hc[1]);
        if (hipMemcpy(c->d_chromOff, off.data(), 8 * off.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(mfail(BBMAP_E_HIP, "bbmap_create: upload failed"));
        if (hipMemset(c->d_chromMin, 0, 4 * ((size_t)nch + 1)) != hipSuccess) return bail(mfail(BBMAP_E_HIP, "bbmap_create: memset failed"));
    }
    if (hipHostMalloc((void **)&c->h_counters, 64 * 4) != hipSuccess) return bail(mfail(BBMAP_E_NOMEM, "bbmap_create: pinned allocation failed"));
    for (int i = 0; i < 12; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) return bail(mfail(BBMAP_E_HIP, "bbmap_create: hipEventCreate failed"));
    if (!getenv("BBMAP_SERIAL_DP") && !pacbio) {
        if (hipStreamCreateWithFlags(&c->dpStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) != hipSuccess) return bail(mfail(BBMAP_E_HIP, "bbmap_create: stream / event creation failed"));
    }
    *out = c;
    return BBMAP_OK;
}

extern "C" int bbmap_create(bbidx_ctx *index, const bbmap_config *cfg, bbmap_ctx **out) {
    bbmap_ctx *c = nullptr;
    MTRY(create_impl(index, cfg, nullptr, &c));
    // reserved[1]: reads the overflow tier holds (0 = 4096, < 0 = no tier); reserved[2]: its max_sites (0 = 1024)
    if (cfg->reserved[1] >= 0) {
        bbmap_config tc = *cfg;
        long long tn = cfg->reserved[1] > 0 ? cfg->reserved[1] : 4096;
        if (tn > cfg->max_reads) tn = cfg->max_reads;
        if (cfg->paired) tn &= ~1ll;
        tc.max_reads = (int32_t)tn;
        tc.max_sites = cfg->reserved[2] > 0 ? cfg->reserved[2] : 1024;
        tc.jobsPerRead = 128;
        tc.reserved[1] = -1;
        if (tn >= (cfg->paired ? 2 : 1) && tc.max_sites > cfg->max_sites) {
            const int rc = create_impl(index, &tc, c, &c->tier);
            if (rc != BBMAP_OK) { bbmap_destroy(c); return rc; }
            const long long units = cfg->paired ? cfg->max_reads / 2 : cfg->max_reads;
            if (dalloc(c, &c->d_tierUnits, (size_t)units + 1) != BBMAP_OK || dalloc(c, &c->d_tierReads, (size_t)tn) != BBMAP_OK ||
                dalloc(c, &c->d_tierReadIds, (size_t)tn) != BBMAP_OK) { bbmap_destroy(c); return BBMAP_E_NOMEM; }
            if (hipStreamCreateWithFlags(&c->tierStream, hipStreamNonBlocking) != hipSuccess) { bbmap_destroy(c); return mfail(BBMAP_E_HIP, "bbmap_create: hipStreamCreate failed"); }
        }
    }
    *out = c;
    return BBMAP_OK;
}

static int read_counters(bbmap_ctx *c, hipStream_t stream) {
    MHIP(hipMemcpyAsync(c->h_counters, c->d_counters, 64 * 4, hipMemcpyDeviceToHost, stream));
    MHIP(hipStreamSynchronize(stream));
    return BBMAP_OK;
}

// Replaces a device array by a larger one (contents kept), in stream order; the old one is freed once the stream has passed.
template <class T> static int regrow(bbmap_ctx *c, hipStream_t stream, T **p, size_t oldCount, size_t newCount, std::vector<void *> &dead) {
    void *d = nullptr;
    if (hipMalloc(&d, (newCount ? newCount : 1) * sizeof(T)) != hipSuccess) return mfail(BBMAP_E_NOMEM, "bbmap_map_batch_device: growing a fill log failed (device memory)");
    if (oldCount) MHIP(hipMemcpyAsync(d, *p, oldCount * sizeof(T), hipMemcpyDeviceToDevice, stream));
    for (void *&q : c->allocs) if (q == (void *)*p) q = d;
    dead.push_back((void *)*p);
    *p = (T *)d;
    return BBMAP_OK;
}
// The reference's per-read lists of fills have no capacity.  When a round asks for more log entries than there are (the kernels
// then hold the affected reads back, emit_fill's NO_ROOM), the logs are grown here before the next round; `needJobs` / `needGapped`
// = entries that must fit.  The device counters are set back to the number of entries that were really written.
// arrays a regrow has replaced: freed when the guard goes out of scope, after the stream has passed the copies (error paths included)
struct DeadArrays {
    hipStream_t stream; std::vector<void *> v;
    explicit DeadArrays(hipStream_t s) : stream(s) {}
    ~DeadArrays() { if (!v.empty()) { (void)hipStreamSynchronize(stream); for (void *q : v) (void)hipFree(q); } }
};
static int grow_logs(bbmap_ctx *c, hipStream_t stream, bbmapper::Dev &D, long long needJobs, long long needGapped, long long usedJobs, long long usedGapped) {
    DeadArrays guard(stream);
    std::vector<void *> &dead = guard.v;
    if (needJobs > c->jobCap) {
        long long nc = c->jobCap * 2; if (nc < needJobs) nc = needJobs + needJobs / 4 + 1024;
        MTRY(regrow(c, stream, &c->d_jobs, (size_t)usedJobs, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_jinfo, (size_t)usedJobs, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_results, (size_t)usedJobs, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_match, (size_t)usedJobs * c->matchStride, (size_t)nc * c->matchStride, dead));
        c->jobCap = nc;
    }
    if (needGapped > c->gjobCap) {
        long long nc = c->gjobCap * 2; if (nc < needGapped) nc = needGapped + needGapped / 4 + 1024;
        MTRY(regrow(c, stream, &c->d_gjobs, (size_t)usedGapped, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_ggaps, (size_t)usedGapped, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_ginfo, (size_t)usedGapped, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_gresults, (size_t)usedGapped, (size_t)nc, dead));
        MTRY(regrow(c, stream, &c->d_gmatch, (size_t)usedGapped * c->gmatchStride, (size_t)nc * c->gmatchStride, dead));
        c->gjobCap = nc;
    }
    c->h_counters[32] = (unsigned)usedJobs; c->h_counters[33] = (unsigned)usedGapped;
    MHIP(hipMemcpyAsync(c->d_counters, c->h_counters + 32, 8, hipMemcpyHostToDevice, stream));
    MHIP(hipStreamSynchronize(stream));
    D.jobs = c->d_jobs; D.jinfo = c->d_jinfo; D.results = c->d_results; D.jobCap = c->jobCap;
    D.gjobs = c->d_gjobs; D.ggaps = c->d_ggaps; D.ginfo = c->d_ginfo; D.gresults = c->d_gresults; D.gjobCap = c->gjobCap;
    c->stats.log_growths += 1.0f;              // (how often the logs grew in this batch)
    return BBMAP_OK;
}

// launches the DP over the fills appended since (jobBase, gjobBase)
static int run_fills(bbmap_ctx *c, hipStream_t stream, const uint8_t *bases, long long jobBase, long long nNew, long long gBase, long long gNew,
                     bool finalStage = false) {
    // The one-job-per-lane narrow kernel runs in front of the wavefront kernel on the same stream and is a ~1.5 ms dependent chain
    // however few jobs there are: worth it only for the big first rounds of scoreSlow (163 k of 459 k fills finish there in 4.8 ms
    // on the bench workload).  The final stage's fills never fit its band (see msa_ctx.h), nor do the second context's wide windows.
    static const long long narrowMin = getenv("BBMAP_NARROW_MIN_JOBS") ? atoll(getenv("BBMAP_NARROW_MIN_JOBS")) : 32768;
    static const bool sortWide = !(getenv("BBMAP_SORT_WIDE") && atoi(getenv("BBMAP_SORT_WIDE")) == 0);      // (experiments: 0 switches the width order off)
    bbmsa_use_narrow(c->msa, !finalStage && nNew >= narrowMin);
    // (not the first context's: its windows span 162..256 columns, and sorted its pass ends 4 ms earlier -- leaving the second
    // context's latency-bound wide pass to finish on its own: final stage 79.6 -> 85.5 ms)
    if (c->msaGapped != c->msa) { bbmsa_use_narrow(c->msaGapped, false); bbmsa_sort_by_width(c->msaGapped, sortWide); }
    // the second context's launches first, on their own stream: its blocks take their share of the CUs and the plain context's
    // persistent blocks fill the rest (and the slots the others free)
    hipStream_t gs = (c->dpStream && nNew > 0) ? c->dpStream : stream;
    if (gNew > 0) {
        if (gs != stream) { MHIP(hipEventRecord(c->evFork, stream)); MHIP(hipStreamWaitEvent(gs, c->evFork, 0)); }
        MTRY(bbmsa_align_gapped_batch_device(c->msaGapped, gs, gNew, c->d_gjobs + gBase, c->d_ggaps + gBase, bases, c->refsBase,
                                             c->d_gresults + gBase, c->d_gmatch + gBase * c->gmatchStride, c->gmatchStride));
        if (gs != stream) { MHIP(hipEventRecord(c->evJoin, gs)); if (nNew > 0) MTRY(bbmsa_wait_first_pass(c->msaGapped, stream)); }
    }
    if (nNew > 0)
        MTRY(bbmsa_align_batch_device(c->msa, stream, nNew, c->d_jobs + jobBase, bases, c->refsBase, c->d_results + jobBase,
                                      c->d_match + jobBase * c->matchStride, c->matchStride));
    if (gNew > 0 && gs != stream) MHIP(hipStreamWaitEvent(stream, c->evJoin, 0));
    return BBMAP_OK;
}

static void add_dp_ms(bbmap_ctx *c, bool plain, bool gapped) {
    float k3[3];
    if (plain && bbmsa_last_kernel_ms3(c->msa, k3) == BBMAP_OK) { c->stats.ms_dp_narrow += k3[0]; c->stats.ms_dp_wave += k3[1]; c->stats.ms_dp_generic += k3[2];
                                                                 if (k3[1] > c->stats.ms_dp_wave_max) c->stats.ms_dp_wave_max = k3[1]; }
    // (mapPacBio has ONE context for both logs: its kernel times are the plain launch's already, a second reading would count them twice)
    if (gapped && c->msaGapped != c->msa && bbmsa_last_kernel_ms3(c->msaGapped, k3) == BBMAP_OK) c->stats.ms_dp_gapped += k3[0] + k3[1] + k3[2];
    static const bool show = getenv("BBMAP_DP_COUNTS") != nullptr;      // where the fills of a launch sequence ended up (experiments)
    if (show) {
        int64_t n4[4];
        if (plain && bbmsa_last_counts(c->msa, n4) == BBMAP_OK)
            fprintf(stderr, "dp counts plain : narrow finished %lld, narrow handed on %lld, wavefront list %lld, to the wide/generic pass %lld\n",
                    (long long)n4[0], (long long)n4[1], (long long)n4[2], (long long)n4[3]);
        if (gapped && bbmsa_last_counts(c->msaGapped, n4) == BBMAP_OK)
            fprintf(stderr, "dp counts second: narrow finished %lld, narrow handed on %lld, wavefront list %lld, to the wide/generic pass %lld\n",
                    (long long)n4[0], (long long)n4[1], (long long)n4[2], (long long)n4[3]);
    }
}

static void tier_start_async(bbmap_ctx *c, long long found);

// The final alignment stage (mapper_final.h) over the site lists in c->d_ms: policy, genMatchString in rounds, policy,
// toLocalAlignment.  jobBase / gBase: entries of the two fill logs already used (and run).
static int run_final_stage(bbmap_ctx *c, hipStream_t stream, bbmapper::Dev &D, int64_t n_reads, const uint8_t *bases, long long jobBase, long long gBase,
                           long long &finalRounds, long long &finalLocal) {
    const unsigned TB = 128;
    const long long units = c->cfg.paired ? n_reads / 2 : n_reads;
    {
        D.fin = c->d_fin; D.finalOut = c->d_final; D.pool = c->d_pool; D.poolUnits = c->poolUnits;
        D.match = c->d_match; D.gmatch = c->d_gmatch; D.matchStride = c->matchStride; D.gmatchStride = c->gmatchStride;
        hipLaunchKernelGGL(bbmapper::final_begin_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, D);
        MHIP(hipGetLastError());
        long long nActive = n_reads; bool first = true; int cur = 0;
        bool ranPlain = false, ranGapped = false;
        for (int round = 0; nActive > 0; round++) {
            if (round > 64 * c->cfg.max_sites + 64) return mfail(BBMAP_E_HIP, "bbmap_map_batch_device: the final stage does not come to an end (internal error)");
            MHIP(hipMemsetAsync(c->d_counters + 2, 0, 4, stream));
            MHIP(hipMemsetAsync(c->d_counters + 21, 0xff, 4, stream));
            MHIP(hipMemsetAsync(c->d_counters + 22, 0, 4, stream));
            D.activeIn = first ? nullptr : c->d_active[cur]; D.nActiveIn = (int)nActive; D.activeOut = c->d_active[1 - cur];
            hipLaunchKernelGGL(bbmapper::final_round_kernel, dim3((unsigned)((nActive + TB - 1) / TB)), dim3(TB), 0, stream, D);
            MHIP(hipGetLastError());
            MTRY(read_counters(c, stream));
            add_dp_ms(c, ranPlain, ranGapped);
            const long long asked = c->h_counters[0], gasked = c->h_counters[1];
            const long long total = asked < c->jobCap ? asked : c->jobCap, gtotal = gasked < c->gjobCap ? gasked : c->gjobCap;
            MTRY(run_fills(c, stream, bases, jobBase, total - jobBase, gBase, gtotal - gBase, true));
            ranPlain = total > jobBase; ranGapped = gtotal > gBase;
            jobBase = total; gBase = gtotal;
            if (asked > c->jobCap || gasked > c->gjobCap) {
                MTRY(grow_logs(c, stream, D, asked, gasked, total, gtotal));
                D.match = c->d_match; D.gmatch = c->d_gmatch;
            }
            if (c->h_counters[22] > 0) {           // the match-string pool was full for some reads: they repeat their step next round
                const long long used = c->h_counters[21];                      // units handed out before the first request that failed
                long long nu = c->poolUnits * 2, need = used + ((long long)c->h_counters[20] - used) * 2 + 65536;
                if (nu < need) nu = need;
                if (nu > 0x7ffffff0LL) return mfail(BBMAP_E_NOMEM, "bbmap_map_batch_device: the final stage's match strings exceed 8 GB; map smaller batches");
                DeadArrays guard(stream);                          // (frees the old pool when this scope is left, also on an error return)
                MTRY(regrow(c, stream, &c->d_pool, (size_t)used * 4, (size_t)nu * 4, guard.v));
                c->h_counters[34] = (unsigned)used;
                MHIP(hipMemcpyAsync(c->d_counters + 20, c->h_counters + 34, 4, hipMemcpyHostToDevice, stream));
                MHIP(hipStreamSynchronize(stream));
                c->poolUnits = nu; D.pool = c->d_pool; D.poolUnits = nu;
            }
            nActive = c->h_counters[2];
            cur = 1 - cur; first = false;
            finalRounds++;
        }
        MHIP(hipStreamSynchronize(stream));
        add_dp_ms(c, ranPlain, ranGapped);
        MHIP(hipMemsetAsync(c->d_counters + 24, 0, 8, stream));
        hipLaunchKernelGGL(bbmapper::final_end_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, D);
        MHIP(hipGetLastError());
        MTRY(read_counters(c, stream));
        finalLocal = c->h_counters[24];
        if ((long long)c->h_counters[20] + (long long)c->h_counters[25] + 64 > c->poolUnits) {      // room for toLocalAlignment's strings
            const long long nu = (long long)c->h_counters[20] + (long long)c->h_counters[25] + 65536;
            if (nu > 0x7ffffff0LL) return mfail(BBMAP_E_NOMEM, "bbmap_map_batch_device: the final stage's match strings exceed 8 GB; map smaller batches");
            DeadArrays guard(stream);
            MTRY(regrow(c, stream, &c->d_pool, (size_t)c->h_counters[20] * 4, (size_t)nu * 4, guard.v));
            MHIP(hipStreamSynchronize(stream));
            c->poolUnits = nu; D.pool = c->d_pool; D.poolUnits = nu;
        }
        hipLaunchKernelGGL(bbmapper::final_local_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, D);
        MHIP(hipGetLastError());
    }
    return BBMAP_OK;
}


static void fill_dev(bbmap_ctx *c, bbmapper::Dev &D, int64_t n_reads, const bbidx_read *reads, uint8_t *bases, int64_t minus_delta) {
    memset(&D, 0, sizeof D);
    D.S = c->S; D.reads = reads; D.bases = bases; D.minusDelta = minus_delta; D.nreads = n_reads;
    D.chromArr = c->d_chromArr; D.chromArrLen = c->d_chromArrLen; D.refsBase = c->refsBase;
    D.psites = c->d_psites; D.pnsites = c->d_pnsites; D.maxSites = c->cfg.max_sites;
    D.ms = c->d_ms; D.mcount = c->d_mcount; D.cap = c->cfg.max_sites; D.nearArr = c->d_near; D.slow = c->d_slow;
    D.counters = c->d_counters; D.plainColumns = c->plainColumns; D.fillAhead = c->cfg.reserved[0] ? 0 : 1;
    D.jobs = c->d_jobs; D.jinfo = c->d_jinfo; D.results = c->d_results; D.jobCap = c->jobCap;
    D.gjobs = c->d_gjobs; D.ggaps = c->d_ggaps; D.ginfo = c->d_ginfo; D.gresults = c->d_gresults; D.gjobCap = c->gjobCap;
    D.rjobs = c->d_rjobs; D.rinfo = c->d_rinfo; D.rres = c->d_rres; D.pres = c->d_pres; D.rescCap = c->rescCap; D.rsite = c->d_rsite;
}

// one context's pass over `n_reads` read records
static int map_records(bbmap_ctx *c, hipStream_t stream, int64_t n_reads, const bbidx_read *reads, uint8_t *bases,
                       int64_t minus_delta, const int8_t *baseScores, const int32_t *keyinfo, bool writeRc) {
    memset(&c->stats, 0, sizeof c->stats);
    c->stats.reads = n_reads;
    MHIP(hipMemsetAsync(c->d_counters, 0, 64 * 4, stream));
    MHIP(hipEventRecord(c->ev[0], stream));
    // ---- probe (BBIndex.findAdvanced); reverse complements are written on the way
    MTRY(bbidx_find_batch_device_with(c->index, &c->probeLs, stream, n_reads, reads, bases, baseScores, keyinfo, c->d_psites, c->cfg.max_sites,
                                      c->d_pnsites, writeRc ? bases + minus_delta : nullptr));
    MHIP(hipEventRecord(c->ev[1], stream));
    bbmapper::Dev D;
    fill_dev(c, D, n_reads, reads, bases, minus_delta);
    const unsigned TB = 128;
    const long long units = c->cfg.paired ? n_reads / 2 : n_reads;
    hipLaunchKernelGGL(bbmapper::begin_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, D);
    MHIP(hipGetLastError());
    if (c->tier) {                       // the units the probe flagged: known now, so the tier can work beside the rest of this pass
        hipLaunchKernelGGL(bbmapper::collect_overflow_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, c->d_mcount, units, c->cfg.paired,
                           c->d_tierUnits, c->d_counters + 16);
        MHIP(hipGetLastError());
    }
    MHIP(hipEventRecord(c->ev[2], stream));
    hipLaunchKernelGGL(bbmapper::score_kernel, dim3((unsigned)((n_reads + TB - 1) / TB)), dim3(TB), 0, stream, D);
    MHIP(hipGetLastError());
    MHIP(hipEventRecord(c->ev[3], stream));
    // ---- scoreSlow in rounds
    long long jobBase = 0, gBase = 0, nActive = n_reads;
    int cur = 0;
    bool first = true, ranPlainPrev = false, ranGappedPrev = false;
    for (int round = 0; nActive > 0 && round < 4 * c->cfg.max_sites + 4; round++) {
        MHIP(hipMemsetAsync(c->d_counters + 2, 0, 4, stream));
        D.activeIn = first ? nullptr : c->d_active[cur]; D.nActiveIn = (int)nActive; D.activeOut = c->d_active[1 - cur];
        hipLaunchKernelGGL(bbmapper::slow_round_kernel, dim3((unsigned)((nActive + TB - 1) / TB)), dim3(TB), 0, stream, D);
        MHIP(hipGetLastError());
        MTRY(read_counters(c, stream));
        if (round == 0) {
            // the probe is over: its statistics are read now
            float pms = 0; long long ps[5];       // (not for the tier's own pass: the synchronous copy inside would wait for the main stream)
            if (writeRc && bbidx_last_stats_with(c->index, &c->probeLs, (int64_t *)ps, &pms) == BBMAP_OK) for (int i = 0; i < 5; i++) c->stats.probe_stats[i] = ps[i];
            c->overAfterBegin = c->h_counters[3];
            if (c->tier && c->h_counters[16] > 0 && c->tier->msa != c->msa) tier_start_async(c, c->h_counters[16]);     // (a tier that borrows the DP context runs after the pass)
        }
        add_dp_ms(c, ranPlainPrev, ranGappedPrev);
        const long long asked = c->h_counters[0], gasked = c->h_counters[1];
        const long long total = asked < c->jobCap ? asked : c->jobCap, gtotal = gasked < c->gjobCap ? gasked : c->gjobCap;     // entries really written
        MTRY(run_fills(c, stream, bases, jobBase, total - jobBase, gBase, gtotal - gBase));
        ranPlainPrev = total > jobBase; ranGappedPrev = gtotal > gBase;
        jobBase = total; gBase = gtotal;
        // a log was too small: the reads that found no room ask again next round (emit_fill's NO_ROOM), after it has grown
        if (asked > c->jobCap || gasked > c->gjobCap) MTRY(grow_logs(c, stream, D, asked, gasked, total, gtotal));
        nActive = c->h_counters[2];
        cur = 1 - cur; first = false;
        c->stats.rounds++;
    }
    MHIP(hipEventRecord(c->ev[4], stream));
    hipLaunchKernelGGL(bbmapper::finish_kernel, dim3((unsigned)((n_reads + TB - 1) / TB)), dim3(TB), 0, stream, D);
    MHIP(hipGetLastError());
    MHIP(hipEventRecord(c->ev[5], stream));
    // ---- rescue: mate 1 anchors, then mate 2
    if (c->cfg.paired && c->cfg.doRescue) {
        const long long pairs = n_reads / 2;
        for (int pass = 0; pass < 2; pass++) {
            D.pass = pass;
            // each pass gets its own region of the search list: reset the search counter, keep the fills' counters
            MHIP(hipMemsetAsync(c->d_counters + 4, 0, 4, stream));
            hipLaunchKernelGGL(bbmapper::rescue_plan_kernel, dim3((unsigned)((pairs + TB - 1) / TB)), dim3(TB), 0, stream, D);
            MHIP(hipGetLastError());
            MTRY(read_counters(c, stream));
            const long long nsearch = c->h_counters[4];
            if (nsearch > c->rescCap) return mfail(BBMAP_E_NOMEM, "bbmap_map_batch_device: rescue list full");
            c->stats.rescue_scans += nsearch;
            if (nsearch == 0) {
                hipLaunchKernelGGL(bbmapper::rescue_finish_kernel, dim3((unsigned)((pairs + TB - 1) / TB)), dim3(TB), 0, stream, D);
                MHIP(hipGetLastError());
                continue;
            }
            hipEvent_t q0 = c->ev[8], q1 = c->ev[9];
            MHIP(hipEventRecord(q0, stream));
            MTRY(bbpipe_quick_rescue_device(stream, nsearch, c->d_rjobs, bases, (const int64_t *)c->d_chromOff, c->d_chromArrLen, c->d_chromMin, c->refsBase,
                                            c->d_rres, c->S.ptsMatch, c->S.ptsMatch2, 1, 100));
            MHIP(hipEventRecord(q1, stream));
            // every search issues at most one fill, into either log: room for all of them before the kernel that writes them
            if (jobBase + nsearch > c->jobCap || gBase + nsearch > c->gjobCap) MTRY(grow_logs(c, stream, D, jobBase + nsearch, gBase + nsearch, jobBase, gBase));
            hipLaunchKernelGGL(bbmapper::rescue_prep_kernel, dim3((unsigned)((pairs + TB - 1) / TB)), dim3(TB), 0, stream, D);
            MHIP(hipGetLastError());
            MTRY(read_counters(c, stream));
            { float ms = 0; if (hipEventElapsedTime(&ms, q0, q1) == hipSuccess) c->stats.ms_quick_rescue += ms; }
            const long long total = c->h_counters[0], gtotal = c->h_counters[1];
            if (total > c->jobCap || gtotal > c->gjobCap) return mfail(BBMAP_E_HIP, "bbmap_map_batch_device: rescue fills beyond the reserved log entries (internal error)");
            MTRY(run_fills(c, stream, bases, jobBase, total - jobBase, gBase, gtotal - gBase));
            const bool ranPlain = total > jobBase, ranGapped = gtotal > gBase;
            jobBase = total; gBase = gtotal;
            hipLaunchKernelGGL(bbmapper::rescue_finish_kernel, dim3((unsigned)((pairs + TB - 1) / TB)), dim3(TB), 0, stream, D);
            MHIP(hipGetLastError());
            MHIP(hipStreamSynchronize(stream));
            add_dp_ms(c, ranPlain, ranGapped);
        }
    }
    MHIP(hipEventRecord(c->ev[6], stream));
    // ---- the final alignment stage
    c->finalFills = 0; c->poolUsed = 0;
    long long finalRounds = 0, finalLocal = 0;
    if (c->S.finalStage) MTRY(run_final_stage(c, stream, D, n_reads, bases, jobBase, gBase, finalRounds, finalLocal));
    MHIP(hipEventRecord(c->ev[7], stream));
    MTRY(read_counters(c, stream));
    c->poolUsed = 4ll * c->h_counters[20];
    c->nJobs = c->h_counters[0]; c->nGapped = c->h_counters[1];
    c->finalFills = c->S.finalStage ? (c->nJobs + c->nGapped) - (jobBase + gBase) : 0;       // (jobBase / gBase: the logs before the final stage)
    bbmap_stats &st = c->stats;
    st.reads_overflowed = c->h_counters[3]; st.reads_without_site = c->h_counters[5];
    st.fills = c->nJobs; st.gapped_fills = c->nGapped; st.refills = c->h_counters[6]; st.rescue_fills = c->h_counters[7]; st.fills_dropped = c->h_counters[8];
    (void)hipEventElapsedTime(&st.ms_probe, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&st.ms_begin, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&st.ms_score, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&st.ms_slow, c->ev[3], c->ev[4]);
    (void)hipEventElapsedTime(&st.ms_finish, c->ev[4], c->ev[5]);
    (void)hipEventElapsedTime(&st.ms_rescue, c->ev[5], c->ev[6]);
    (void)hipEventElapsedTime(&st.ms_final, c->ev[6], c->ev[7]);
    (void)hipEventElapsedTime(&st.ms_total, c->ev[0], c->ev[7]);
    st.final_fills = c->finalFills; st.final_rounds = finalRounds; st.final_local = finalLocal;
    c->ran = true;
    return BBMAP_OK;
}

// The reference's ArrayList<SiteScore> has no capacity (BBIndex.java:1537-1604).  Reads whose list did not fit max_sites are
// mapped again, from the probe on, by the tier context with its long lists (pairs as pairs); a read the tier cannot hold either
// stays flagged.  The reads the PROBE flagged are known once begin_kernel has run, and the tier maps them on its own stream
// beside the rest of the main pass.  A list can also outgrow max_sites when rescue appends to it (rare): then the tier runs once
// more after the main pass, over all flagged reads.
static int tier_pass(bbmap_ctx *c, hipStream_t s, long long found) {
    bbmap_ctx *t = c->tier;
    const bbmap_ctx::BatchArgs &B = c->batch;
    const int paired = c->cfg.paired;
    const unsigned TB = 128;
    c->tierReads = 0; t->ran = false;
    std::vector<int> ids((size_t)found);
    MHIP(hipMemcpyAsync(ids.data(), c->d_tierUnits, 4 * (size_t)found, hipMemcpyDeviceToHost, s));
    MHIP(hipStreamSynchronize(s));
    std::sort(ids.begin(), ids.end());
    const long long room = paired ? t->cfg.max_reads / 2 : t->cfg.max_reads;
    const long long take = found < room ? found : room;           // the first `room` units in read order; the rest stay flagged
    MHIP(hipMemcpyAsync(c->d_tierUnits, ids.data(), 4 * (size_t)take, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bbmapper::gather_reads_kernel, dim3((unsigned)((take + TB - 1) / TB)), dim3(TB), 0, s, B.reads, c->d_tierUnits, (int)take, paired,
                       c->d_tierReads, c->d_tierReadIds);
    MHIP(hipGetLastError());
    MHIP(hipStreamSynchronize(s));                                // `ids` is done with
    const long long tn = paired ? 2 * take : take;
    // the reverse complements of these reads are in place (the main probe wrote them)
    MTRY(map_records(t, s, tn, c->d_tierReads, B.bases, B.minus_delta, B.baseScores, B.keyinfo, false));
    c->tierReads = tn;
    return BBMAP_OK;
}

static void tier_start_async(bbmap_ctx *c, long long found) {
    c->tierStarted = true; c->tierRc = BBMAP_OK; c->tierErr[0] = 0;
    c->tierThread = std::thread([c, found]() {
        int rc = hipSetDevice(c->cfg.device) == hipSuccess ? BBMAP_OK : BBMAP_E_HIP;
        if (rc == BBMAP_OK) rc = tier_pass(c, c->tierStream, found);
        if (rc != BBMAP_OK) { snprintf(c->tierErr, sizeof c->tierErr, "overflow tier: %s", bbmap_last_error()); }
        c->tierRc = rc;
    });
}

// after the tier's pass: its reads are marked in the main list, its counts join the batch's statistics
static int tier_finish(bbmap_ctx *c, hipStream_t stream) {
    bbmap_ctx *t = c->tier;
    const long long tn = c->tierReads;
    if (tn == 0) return BBMAP_OK;
    const unsigned TB = 128;
    MHIP(hipMemsetAsync(c->d_counters + 17, 0, 4, stream));
    hipLaunchKernelGGL(bbmapper::mark_tier_kernel, dim3((unsigned)((tn + TB - 1) / TB)), dim3(TB), 0, stream, c->d_mcount, t->d_mcount, c->d_tierReadIds, (int)tn,
                       c->d_counters + 17);
    MHIP(hipGetLastError());
    MTRY(read_counters(c, stream));
    bbmap_stats &st = c->stats; const bbmap_stats &ts = t->stats;
    st.reads_reprobed = tn;
    st.reads_overflowed -= (long long)c->h_counters[17];
    st.reads_without_site += ts.reads_without_site;
    st.fills += ts.fills; st.gapped_fills += ts.gapped_fills; st.refills += ts.refills; st.rescue_scans += ts.rescue_scans;
    st.rescue_fills += ts.rescue_fills; st.fills_dropped += ts.fills_dropped;
    st.final_fills += ts.final_fills; st.final_local += ts.final_local;
    if (getenv("BBMAP_TIER_DEBUG"))
        fprintf(stderr, "[bbmap tier] reads %lld: probe %.2f begin %.2f score %.2f slow %.2f (rounds %lld) finish %.2f rescue %.2f total %.2f\n",
                tn, ts.ms_probe, ts.ms_begin, ts.ms_score, ts.ms_slow, (long long)ts.rounds, ts.ms_finish, ts.ms_rescue, ts.ms_total);
    return BBMAP_OK;
}

extern "C" int bbmap_map_batch_device(bbmap_ctx *c, void *stream_, int64_t n_reads, const bbidx_read *reads, uint8_t *bases,
                                      int64_t minus_delta, const int8_t *baseScores, const int32_t *keyinfo) {
    if (!c) return mfail(BBMAP_E_ARG, "bbmap_map_batch_device: null context");
    if (n_reads < 0 || n_reads > c->cfg.max_reads) return mfail(BBMAP_E_ARG, "bbmap_map_batch_device: more reads than the context was made for");
    if (c->cfg.paired && (n_reads & 1)) return mfail(BBMAP_E_ARG, "bbmap_map_batch_device: paired mode takes an even number of reads");
    if (n_reads == 0) { c->ran = false; return BBMAP_OK; }
    if (!reads || !bases || !baseScores || !keyinfo) return mfail(BBMAP_E_ARG, "bbmap_map_batch_device: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    MHIP(hipSetDevice(c->cfg.device));
    c->batch = {n_reads, reads, bases, minus_delta, baseScores, keyinfo};
    c->tierStarted = false; c->tierReads = 0;
    if (c->tier) c->tier->ran = false;
    const int rc = map_records(c, stream, n_reads, reads, bases, minus_delta, baseScores, keyinfo, true);
    hipEvent_t e0 = c->ev[10], e1 = c->ev[11];
    // the tier's helper thread is joined before anything else can return: a joinable std::thread left behind would terminate the
    // process at the next batch's assignment
    if (c->tierStarted) {
        c->tierThread.join();
        if (rc == BBMAP_OK && c->tierRc != BBMAP_OK) return mfail(c->tierRc, c->tierErr);
    }
    MTRY(rc);
    if (c->tier) MHIP(hipEventRecord(e0, stream));
    if (!c->tier || c->stats.reads_overflowed == 0) return BBMAP_OK;
    if (c->stats.reads_overflowed > c->overAfterBegin || !c->tierStarted) {
        // lists that outgrew max_sites in rescue: one more tier pass, over every flagged read
        const long long units = c->cfg.paired ? n_reads / 2 : n_reads;
        const unsigned TB = 128;
        MHIP(hipMemsetAsync(c->d_counters + 16, 0, 4, stream));
        hipLaunchKernelGGL(bbmapper::collect_overflow_kernel, dim3((unsigned)((units + TB - 1) / TB)), dim3(TB), 0, stream, c->d_mcount, units, c->cfg.paired,
                           c->d_tierUnits, c->d_counters + 16);
        MHIP(hipGetLastError());
        MTRY(read_counters(c, stream));
        if (c->h_counters[16] > 0) MTRY(tier_pass(c, stream, c->h_counters[16]));
    }
    MTRY(tier_finish(c, stream));
    MHIP(hipEventRecord(e1, stream));
    MHIP(hipStreamSynchronize(stream));
    (void)hipEventElapsedTime(&c->stats.ms_overflow, e0, e1);      // what the tier added to the batch after the main pass
    c->stats.ms_total += c->stats.ms_overflow;
    return BBMAP_OK;
}

// The final alignment stage alone, over site lists the caller provides (see include/bbmap_amd.h).
extern "C" int bbmap_final_batch_device(bbmap_ctx *c, void *stream_, int64_t n_reads, const bbidx_read *reads, uint8_t *bases, int64_t minus_delta,
                                        const bbmap_msite *sites, const int32_t *nsites) {
    if (!c) return mfail(BBMAP_E_ARG, "bbmap_final_batch_device: null context");
    if (!c->S.finalStage) return mfail(BBMAP_E_ARG, "bbmap_final_batch_device: the context was created without the final stage");
    if (n_reads < 1 || n_reads > c->cfg.max_reads || (c->cfg.paired && (n_reads & 1))) return mfail(BBMAP_E_ARG, "bbmap_final_batch_device: bad read count");
    if (!reads || !bases || !sites || !nsites) return mfail(BBMAP_E_ARG, "bbmap_final_batch_device: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    MHIP(hipSetDevice(c->cfg.device));
    c->tierStarted = false; c->tierReads = 0;
    if (c->tier) c->tier->ran = false;
    memset(&c->stats, 0, sizeof c->stats);
    c->stats.reads = n_reads;
    MHIP(hipMemsetAsync(c->d_counters, 0, 64 * 4, stream));
    MHIP(hipMemsetAsync(c->d_slow, 0, sizeof(bbmapper::SlowState) * (size_t)n_reads, stream));      // fills are numbered from 0
    MHIP(hipMemcpyAsync(c->d_ms, sites, sizeof(bbmap_msite) * (size_t)n_reads * (size_t)c->cfg.max_sites, hipMemcpyDeviceToDevice, stream));
    MHIP(hipMemcpyAsync(c->d_mcount, nsites, 4 * (size_t)n_reads, hipMemcpyDeviceToDevice, stream));
    MHIP(hipEventRecord(c->ev[6], stream));
    bbmapper::Dev D;
    fill_dev(c, D, n_reads, reads, bases, minus_delta);
    long long finalRounds = 0, finalLocal = 0;
    MTRY(run_final_stage(c, stream, D, n_reads, bases, 0, 0, finalRounds, finalLocal));
    MHIP(hipEventRecord(c->ev[7], stream));
    MTRY(read_counters(c, stream));
    c->poolUsed = 4ll * c->h_counters[20];
    c->nJobs = c->h_counters[0]; c->nGapped = c->h_counters[1];
    c->finalFills = c->nJobs + c->nGapped;
    bbmap_stats &st = c->stats;
    st.fills = c->nJobs; st.gapped_fills = c->nGapped;
    (void)hipEventElapsedTime(&st.ms_final, c->ev[6], c->ev[7]);
    st.ms_total = st.ms_final;
    st.final_fills = c->finalFills; st.final_rounds = finalRounds; st.final_local = finalLocal;
    c->ran = true;
    return BBMAP_OK;
}

// The batch's site lists without the empty slots: counts[r] sites of read r (0 for a read without a list: no site, flagged, or
// mapped by the overflow tier) at packed[offsets[r] ...], offsets = exclusive prefix sums of counts (offsets[n] = their total).
extern "C" int bbmap_pack_sites_device(bbmap_ctx *c, void *stream_, int64_t n_reads, int32_t *counts, int64_t *offsets, bbmap_msite *packed,
                                       int64_t packed_cap) {
    if (!c || !counts || !offsets || !packed) return mfail(BBMAP_E_ARG, "bbmap_pack_sites_device: null argument");
    if (!c->ran || n_reads != c->stats.reads) return mfail(BBMAP_E_ARG, "bbmap_pack_sites_device: n_reads is not the last batch's");
    hipStream_t stream = (hipStream_t)stream_;
    MHIP(hipSetDevice(c->cfg.device));
    const long long n = n_reads;
    size_t need = 0;
    // (the scan's accumulator type follows its INPUT type: the counts go in as long long so that offsets beyond 2^31 records stay exact)
    auto wide = hipcub::TransformInputIterator<long long, bbmapper::ToLL, const int *>((const int *)counts, bbmapper::ToLL());
    MHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, wide, (long long *)offsets, (int)(n + 1), stream));
    if (need > c->packTmpBytes) {
        if (c->d_packTmp) { MHIP(hipStreamSynchronize(stream)); (void)hipFree(c->d_packTmp); c->d_packTmp = nullptr; c->packTmpBytes = 0; }
        MHIP(hipMalloc(&c->d_packTmp, need));
        c->packTmpBytes = need;
    }
    // counts has n + 1 entries for the scan (the last one a zero), so that offsets[n] is the total
    hipLaunchKernelGGL(bbmapper::pack_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c->d_mcount, n, counts);
    MHIP(hipGetLastError());
    MHIP(hipMemsetAsync(counts + n, 0, 4, stream));
    MHIP(hipcub::DeviceScan::ExclusiveSum(c->d_packTmp, need, wide, (long long *)offsets, (int)(n + 1), stream));
    const long long threads = n * c->cfg.max_sites * 8;
    hipLaunchKernelGGL(bbmapper::pack_sites_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, c->d_ms, c->d_mcount, (const long long *)offsets, n,
                       c->cfg.max_sites, (long long)packed_cap, packed);
    MHIP(hipGetLastError());
    return BBMAP_OK;
}

// Host-buffer form of the batch call, for a host that owns no device memory (the JNI glue, jni/hip_glue.c): uploads the batch,
// maps it, packs the site lists and copies them back.  Lists of reads the overflow tier mapped are appended behind the packed ones.
static int hio_grow(bbmap_ctx *c, int i, size_t need) {
    if (need <= c->hio.cap[i]) return BBMAP_OK;
    if (c->hio.p[i]) { (void)hipFree(c->hio.p[i]); c->hio.p[i] = nullptr; c->hio.cap[i] = 0; }
    need += need / 4 + 256;
    MHIP(hipMalloc(&c->hio.p[i], need));
    c->hio.cap[i] = need;
    return BBMAP_OK;
}
extern "C" int bbmap_map_batch(bbmap_ctx *c, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases, int64_t bases_bytes,
                               const int8_t *baseScores, const int32_t *keyinfo, int64_t keyinfo_ints, int32_t *nsites_out,
                               int64_t *offsets_out, bbmap_msite *sites_out, int64_t sites_cap, int64_t *total_out) {
    if (!c) return mfail(BBMAP_E_ARG, "bbmap_map_batch: null context");
    if (n_reads < 0 || n_reads > c->cfg.max_reads) return mfail(BBMAP_E_ARG, "bbmap_map_batch: more reads than the context was made for");
    if (total_out) *total_out = 0;
    if (n_reads == 0) return BBMAP_OK;
    if (!reads || !bases || !baseScores || !keyinfo || !nsites_out || !offsets_out || (sites_cap > 0 && !sites_out) || sites_cap < 0 ||
        bases_bytes < 0 || keyinfo_ints < 0)
        return mfail(BBMAP_E_ARG, "bbmap_map_batch: bad argument");
    for (int64_t r = 0; r < n_reads; r++) {
        const bbidx_read &rd = reads[r];
        if (rd.len < 0 || rd.bases_off < 0 || rd.bases_off + rd.len > bases_bytes)
            return mfail(BBMAP_E_ARG, "bbmap_map_batch: a read lies outside the bases buffer");
        if (rd.nkeys < 0 || rd.keys_off < 0 || rd.keys_off + 2LL * rd.nkeys > keyinfo_ints)
            return mfail(BBMAP_E_ARG, "bbmap_map_batch: a read's key offsets and scores lie outside keyinfo");
        // the probe kernels index the read with these offsets (LDS and global memory): every key inside its read, offsets ascending
        // (KeyRing.makeOffsets3 gives them so), no more keys than the profile's kernels take
        if (rd.nkeys > (c->cfg.reserved[3] == BBIDX_PROFILE_PACBIO ? BBIDX_PACBIO_MAX_KEYS : BBIDX_MAX_KEYS))
            return mfail(BBMAP_E_ARG, "bbmap_map_batch: a read has more keys than the index profile allows (BBIDX_MAX_KEYS / BBIDX_PACBIO_MAX_KEYS)");
        const int kk = c->index->dev.p.k;
        for (int q = 0; q < rd.nkeys; q++) {
            const int o = keyinfo[rd.keys_off + q];
            if (o < 0 || o + kk > rd.len || (q > 0 && o < keyinfo[rd.keys_off + q - 1]))
                return mfail(BBMAP_E_ARG, "bbmap_map_batch: a key offset lies outside its read, or the offsets are not ascending");
        }
    }
    MHIP(hipSetDevice(c->cfg.device));
    const size_t nb = (size_t)bases_bytes;
    MTRY(hio_grow(c, 0, (size_t)n_reads * sizeof(bbidx_read)));
    MTRY(hio_grow(c, 1, 2 * nb + 16));
    MTRY(hio_grow(c, 2, nb + 16));
    MTRY(hio_grow(c, 3, (size_t)keyinfo_ints * 4 + 16));
    MTRY(hio_grow(c, 4, (size_t)(n_reads + 1) * 4));
    MTRY(hio_grow(c, 5, (size_t)(n_reads + 1) * 8));
    MTRY(hio_grow(c, 6, (size_t)(sites_cap > 0 ? sites_cap : 1) * sizeof(bbmap_msite)));
    // A stream of this context's own, non-blocking: several mapping threads, each with its own bbmap_ctx on one shared index (BBMap's
    // thread model), then overlap on the GPU instead of queueing behind one another on the legacy default stream.
    if (!c->hostStream) MHIP(hipStreamCreateWithFlags(&c->hostStream, hipStreamNonBlocking));
    hipStream_t hs = c->hostStream;
    MHIP(hipMemcpyAsync(c->hio.p[0], reads, (size_t)n_reads * sizeof(bbidx_read), hipMemcpyHostToDevice, hs));
    MHIP(hipMemcpyAsync(c->hio.p[1], bases, nb, hipMemcpyHostToDevice, hs));
    MHIP(hipMemcpyAsync(c->hio.p[2], baseScores, nb, hipMemcpyHostToDevice, hs));
    MHIP(hipMemcpyAsync(c->hio.p[3], keyinfo, (size_t)keyinfo_ints * 4, hipMemcpyHostToDevice, hs));
    MTRY(bbmap_map_batch_device(c, hs, n_reads, (const bbidx_read *)c->hio.p[0], (uint8_t *)c->hio.p[1], (int64_t)nb,
                                (const int8_t *)c->hio.p[2], (const int32_t *)c->hio.p[3]));
    MTRY(bbmap_pack_sites_device(c, hs, n_reads, (int32_t *)c->hio.p[4], (int64_t *)c->hio.p[5], (bbmap_msite *)c->hio.p[6], sites_cap));
    MHIP(hipMemcpyAsync(nsites_out, c->d_mcount, (size_t)n_reads * 4, hipMemcpyDeviceToHost, hs));      // counts, or the flags (-1, -2, -3)
    MHIP(hipMemcpyAsync(offsets_out, c->hio.p[5], (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, hs));
    MHIP(hipStreamSynchronize(hs));
    long long total = offsets_out[n_reads];
    const long long have = total < sites_cap ? total : sites_cap;
    if (have > 0) { MHIP(hipMemcpyAsync(sites_out, c->hio.p[6], (size_t)have * sizeof(bbmap_msite), hipMemcpyDeviceToHost, hs)); MHIP(hipStreamSynchronize(hs)); }
    bbmap_overflow_output ov;
    MTRY(bbmap_get_overflow_output(c, &ov));
    if (ov.n_reads > 0) {
        std::vector<int32_t> ids((size_t)ov.n_reads), tn((size_t)ov.n_reads);
        MHIP(hipMemcpy(ids.data(), ov.read_ids, (size_t)ov.n_reads * 4, hipMemcpyDeviceToHost));
        MHIP(hipMemcpy(tn.data(), ov.out.nsites, (size_t)ov.n_reads * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < ov.n_reads; i++) {
            const int32_t r = ids[(size_t)i];
            if (r < 0 || r >= n_reads || nsites_out[r] != BBMAP_NSITES_IN_TIER) continue;
            nsites_out[r] = tn[(size_t)i];
            offsets_out[r] = total;
            if (tn[(size_t)i] <= 0) continue;
            if (total + tn[(size_t)i] <= sites_cap)
                MHIP(hipMemcpy(sites_out + total, ov.out.sites + i * (int64_t)ov.out.cap, (size_t)tn[(size_t)i] * sizeof(bbmap_msite), hipMemcpyDeviceToHost));
            total += tn[(size_t)i];
        }
    }
    if (total_out) *total_out = total;
    return BBMAP_OK;
}

extern "C" int bbmap_get_overflow_output(bbmap_ctx *c, bbmap_overflow_output *o) {
    if (!c || !o) return mfail(BBMAP_E_ARG, "bbmap_get_overflow_output: null argument");
    if (!c->ran) return mfail(BBMAP_E_ARG, "bbmap_get_overflow_output: no batch has been mapped yet");
    memset(o, 0, sizeof *o);
    if (!c->tier || c->tierReads == 0 || !c->tier->ran) return BBMAP_OK;
    o->n_reads = c->tierReads; o->read_ids = c->d_tierReadIds;
    return bbmap_get_output(c->tier, &o->out);
}

extern "C" int bbmap_get_output(bbmap_ctx *c, bbmap_output *o) {
    if (!c || !o) return mfail(BBMAP_E_ARG, "bbmap_get_output: null argument");
    if (!c->ran) return mfail(BBMAP_E_ARG, "bbmap_get_output: no batch has been mapped yet");
    memset(o, 0, sizeof *o);
    o->sites = c->d_ms; o->nsites = c->d_mcount; o->cap = c->cfg.max_sites;
    o->match_stride = c->matchStride; o->gmatch_stride = c->gmatchStride;
    o->n_jobs = c->nJobs; o->n_gapped_jobs = c->nGapped;
    o->jobs = c->d_jobs; o->results = c->d_results; o->jobinfo = c->d_jinfo; o->match = c->d_match;
    o->gjobs = c->d_gjobs; o->gresults = c->d_gresults; o->gjobinfo = c->d_ginfo; o->gmatch = c->d_gmatch; o->ggaps = c->d_ggaps;
    if (c->S.finalStage) { o->final = c->d_final; o->final_match = c->d_pool; o->final_match_bytes = c->poolUsed; o->n_final_fills = c->finalFills; }
    return BBMAP_OK;
}

extern "C" int bbmap_set_average_pair_dist(bbmap_ctx *c, int32_t v) {
    if (!c) return mfail(BBMAP_E_ARG, "bbmap_set_average_pair_dist: null context");
    if (v < 0) return mfail(BBMAP_E_ARG, "bbmap_set_average_pair_dist: negative distance");
    c->S.averagePairDist = v; c->cfg.averagePairDist = v;
    if (c->tier) { c->tier->S.averagePairDist = v; c->tier->cfg.averagePairDist = v; }
    return BBMAP_OK;
}

// The last batch's final records on the host, overflow tier included; match strings packed in read order.
extern "C" int bbmap_get_final(bbmap_ctx *c, int64_t n_reads, bbmap_final *out, uint8_t *match_out, int64_t match_cap, int64_t *match_bytes) {
    if (!c || !out) return mfail(BBMAP_E_ARG, "bbmap_get_final: null argument");
    if (!c->S.finalStage) return mfail(BBMAP_E_ARG, "bbmap_get_final: the context runs without the final stage (bbmap_config.finalStage)");
    if (!c->ran || n_reads != c->stats.reads) return mfail(BBMAP_E_ARG, "bbmap_get_final: n_reads is not the last batch's");
    if (match_cap < 0 || (match_cap > 0 && !match_out)) return mfail(BBMAP_E_ARG, "bbmap_get_final: bad match buffer");
    MHIP(hipSetDevice(c->cfg.device));
    MHIP(hipMemcpy(out, c->d_final, (size_t)n_reads * sizeof(bbmap_final), hipMemcpyDeviceToHost));
    std::vector<uint8_t> pool, tpool;
    if (match_out) {
        pool.resize((size_t)c->poolUsed + 4);
        if (c->poolUsed > 0) MHIP(hipMemcpy(pool.data(), c->d_pool, (size_t)c->poolUsed, hipMemcpyDeviceToHost));
    }
    std::vector<uint8_t> fromTier((size_t)n_reads, 0);
    if (c->tier && c->tierReads > 0 && c->tier->ran) {
        bbmap_ctx *t = c->tier;
        std::vector<int32_t> ids((size_t)c->tierReads);
        std::vector<bbmap_final> tf((size_t)c->tierReads);
        MHIP(hipMemcpy(ids.data(), c->d_tierReadIds, (size_t)c->tierReads * 4, hipMemcpyDeviceToHost));
        MHIP(hipMemcpy(tf.data(), t->d_final, (size_t)c->tierReads * sizeof(bbmap_final), hipMemcpyDeviceToHost));
        if (match_out) {
            tpool.resize((size_t)t->poolUsed + 4);
            if (t->poolUsed > 0) MHIP(hipMemcpy(tpool.data(), t->d_pool, (size_t)t->poolUsed, hipMemcpyDeviceToHost));
        }
        for (long long i = 0; i < c->tierReads; i++) {
            const int32_t r = ids[(size_t)i];
            if (r < 0 || r >= n_reads || out[r].nsites != BBMAP_NSITES_IN_TIER) continue;
            out[r] = tf[(size_t)i]; fromTier[(size_t)r] = 1;
        }
    }
    int64_t used = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        bbmap_final &f = out[r];
        if (f.match_len <= 0) { f.match_off = 0; continue; }
        if (match_out) {
            const std::vector<uint8_t> &src = fromTier[(size_t)r] ? tpool : pool;
            if (f.match_off < 0 || f.match_off + f.match_len > (int64_t)src.size()) return mfail(BBMAP_E_HIP, "bbmap_get_final: a match string lies outside its pool (internal error)");
            if (used + f.match_len <= match_cap) memcpy(match_out + used, src.data() + f.match_off, (size_t)f.match_len);
        }
        f.match_off = used; used += f.match_len;
    }
    if (match_bytes) *match_bytes = used;
    return BBMAP_OK;
}

extern "C" int bbmap_last_stats(bbmap_ctx *c, bbmap_stats *out) {
    if (!c || !out) return mfail(BBMAP_E_ARG, "bbmap_last_stats: null argument");
    if (!c->ran) return mfail(BBMAP_E_ARG, "bbmap_last_stats: no batch has been mapped yet");
    *out = c->stats;
    return BBMAP_OK;
}

extern "C" int bbmap_copy_to_host(void *dst, const void *src_device, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src_device))) return mfail(BBMAP_E_ARG, "bbmap_copy_to_host: bad argument");
    if (bytes > 0) MHIP(hipMemcpy(dst, src_device, (size_t)bytes, hipMemcpyDeviceToHost));
    return BBMAP_OK;
}
