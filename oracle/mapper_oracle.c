/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (same rules as msa11ts_oracle.h).
 *
 * CPU restatement of the mapper control flow AROUND the two hot kernels, i.e. the part of
 * BBMapThread.processRead / processReadPair that decides which probe sites are aligned, with which window and
 * minScore, in which order, and which rescue searches run (paths relative to /root/reference):
 *   quickMap tail            current/align2/AbstractMapThread.java:736-751  (findAdvanced, removeOutOfBounds :2444-2476)
 *   processRead   (single)   current/align2/BBMapThread.java:389-490   (trimList, scoreNoIndels, sort, findTipDeletions,
 *                                                                        scoreSlow, mergeDuplicateSites, sort)
 *   processReadPair (paired) current/align2/BBMapThread.java:943-1098  (pairSiteScoresInitial :736-940, trimList :140-249,
 *                                                                        scoreNoIndels/scoreSlow per mate, rescue x2)
 *   scoreNoIndels            current/align2/AbstractMapThread.java:762-856
 *   findTipDeletions         current/align2/AbstractMapThread.java:1075-1141, 2178-2292
 *   scoreSlow                current/align2/BBMapThread.java:252-386
 *   rescue / slowRescue      current/align2/AbstractMapThread.java:1144-1306  (quickRescue: rescue_oracle.c)
 *   Tools.*                  current/align2/Tools.java:654-674 trimSiteList, :697-759 mergeDuplicateSites,
 *                            :934-960 removeLowQualitySitesPaired, :1113-1161 trimSitesBelowCutoff
 *   SiteScore                current/stream/SiteScore.java:55-73 compareTo, :379-395 PCOMP, :905-998 setters
 *   GapTools                 current/align2/GapTools.java:27-204
 * Default configuration of bbmap.sh (BBMap.setDefaults, current/align2/BBMap.java:45-65): QUICK_MATCH_STRINGS=false,
 * TRIM_LIST=true, RESCUE=true, TIP_SEARCH_DIST=100, PENALIZE_AMBIG=true, STRICT_MAX_INDEL=false, quality-less reads.
 * NOT restated (thread-global adaptive state of the Java mapper, different for every thread count):
 *   DYNAMIC_INSERT_LENGTH (AVERAGE_PAIR_DIST stays at its initial value), the "mating is not working" skip at the top of
 *   rescue(); scaffold boundaries inside a chromosome (removeOutOfBounds' isSingleScaffold test: one scaffold per chromosome).
 * Added product (not in the reference's default flow): the traceback string of every successful fill is recorded, as the
 * quickmatch=t branch would obtain it (BBMapThread.java:345), without fixXY / clipTipIndels; site state follows the default.
 *
 * PARITY STATUS: restatement only (Java-only code; no JVM here).  tests/test_golden_phix.py anchors the end result
 * to the truth coordinates the reference's own fixture carries (resources/sample1.fq.gz, sample2.fq.gz).
 */
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "index_oracle.h"
#include "mapper_oracle.h"
#include "msa11ts_oracle.h"

void orc_set_perfect(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int start, int stop, int32_t *out2);
void orc_quick_rescue(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int minIndex,
                      int loc, int searchDist, int searchRight, int idealStart, int maxAllowedMismatches,
                      int pointsMatch, int pointsMatch2, int useAffine, int baseHitScore, int32_t *out8);

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabsdif(int a, int b) { return a > b ? a - b : b - a; }

/* Shared.java:21-26 */
enum { GAPBUFFER2 = 128, GAPLEN = 128, MINGAP = 256 };
/* AbstractMapThread.java:2987-2993 */
enum { TIP_DELETION_MAX_TIPLEN = 8, OUTER_DIST_MULT = 14, OUTER_DIST_DIV = 32 };
enum { MIN_TRIM_SITES_TO_RETAIN_SINGLE = 3, MIN_TRIM_SITES_TO_RETAIN_PAIRED = 2 };   /* BBMapThread.java:62-63 */

#define LISTCAP 2048
/* AbstractMapThread.java:142: CLEARZONE1e = 2*POINTS_MATCH2 - POINTS_MATCH - POINTS_SUB + 1; quickRescue's points (:1205-1206) */
#ifdef ORC_PACBIO
enum { CLEARZONE1E = 2 * 100 - 90 + 137 + 1, QR_MATCH = 90, QR_MATCH2 = 100 };     /* MultiStateAligner9PacBio.java:2377-2380 */
#else
enum { CLEARZONE1E = 2 * 100 - 70 + 127 + 1, QR_MATCH = 70, QR_MATCH2 = 100 };     /* = 258 */
#endif

typedef struct { orc_msite s[LISTCAP]; int n; } slist;
typedef struct { uint8_t *p; int len; } mstr;       /* a match string of the final stage (final_stage.inc) */

typedef struct {
    const orc_index *ix;
    const orc_map_params *P;
    orc_msa *msa;
    uint8_t *tb;                   /* traceback scratch */
    int tbcap;
    /* job log (optional) */
    orc_mjob *log; int64_t logcap; volatile int64_t *nlog;
    uint8_t *match; int matchStride;
    int64_t dpJobs, cells, rescueScans;
    int seq[2];                    /* per-read fill sequence numbers of the pair being processed */
    int64_t readIdx[2];
    mstr *ms; int nms, mscap;      /* match strings of the pair being processed (final stage) */
    int64_t finalFills;
} mapper;

/* ------------------------------------------------------------------ SiteScore setters (SiteScore.java:905-998) */
static int overlap4(int a1, int b1, int a2, int b2) { return a2 <= b1 && b2 >= a1; }   /* Tools.overlap */
static int constrict(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }

/* GapTools.fixGaps2 (:127-175) */
static void fix_gaps2(orc_msite *ss, int minGap) {
    int ra[ORC_MAX_GAPS / 2], rb[ORC_MAX_GAPS / 2], alive[ORC_MAX_GAPS / 2];
    const int nr = ss->ngaps / 2;
    for (int i = 0; i < nr; i++) { ra[i] = ss->gaps[2 * i]; rb[i] = ss->gaps[2 * i + 1]; alive[i] = 1; }
    for (int i = 1; i < nr; i++) {
        if (alive[i - 1] && ra[i] - rb[i - 1] <= minGap) {
            ra[i] = imin(ra[i - 1], ra[i]); rb[i] = imax(rb[i - 1], rb[i]); alive[i - 1] = 0;
        }
    }
    int m = 0;
    for (int i = 0; i < nr; i++) if (alive[i]) { ss->gaps[2 * m] = ra[i]; ss->gaps[2 * m + 1] = rb[i]; m++; }
    ss->ngaps = (m < 2) ? 0 : 2 * m;
}

/* GapTools.fixGaps(a, b, gaps, minGap) (:27-72) on ss->gaps; ngaps = 0 stands for null */
static void fix_gaps(orc_msite *ss, int a, int b) {
    if (ss->ngaps == 0) return;
    int *g = ss->gaps; const int n = ss->ngaps;
    if (!overlap4(a, b, g[0], g[n - 1])) { ss->ngaps = 0; return; }
    int changed = 0;
    if (g[0] != a) { g[0] = a; changed++; }
    if (g[n - 1] != b) { g[n - 1] = b; changed++; }
    for (int i = 0; i < n; i++) { if (g[i] < a) { g[i] = a; changed++; } else if (g[i] > b) { g[i] = b; changed++; } }
    for (int i = 1; i < n; i++) if (g[i - 1] > g[i]) { g[i] = g[i - 1]; changed++; }
    if (changed == 0) return;
    g[0] = a; g[n - 1] = b;
    int remove = 0;
    for (int i = 0; i < n; i += 2) { g[i] = constrict(g[i], a, b); g[i + 1] = constrict(g[i + 1], a, b); if (g[i] == g[i + 1]) remove++; }
    if (remove == 0) return;
    fix_gaps2(ss, MINGAP);
}

static int check_gaps(const orc_msite *ss) {           /* SiteScore.CHECKGAPS :952-959 */
    if (ss->ngaps == 0) return 1;
    if (ss->ngaps & 1) return 0;
    for (int i = 1; i < ss->ngaps; i++) if (ss->gaps[i - 1] > ss->gaps[i]) return 0;
    return ss->gaps[0] == ss->start && ss->gaps[ss->ngaps - 1] == ss->stop;
}
static void set_limits(orc_msite *ss, int a, int b) {
    ss->start = a; ss->stop = b;
    if (ss->ngaps) { ss->gaps[0] = a; ss->gaps[ss->ngaps - 1] = b; if (!check_gaps(ss)) fix_gaps(ss, ss->start, ss->stop); }
}
static void set_start(orc_msite *ss, int a) {
    ss->start = a;
    if (ss->ngaps) { ss->gaps[0] = a; if (ss->gaps[0] > ss->gaps[1]) fix_gaps(ss, ss->start, ss->stop); }
}
static void set_stop(orc_msite *ss, int b) {
    ss->stop = b;
    if (ss->ngaps) { ss->gaps[ss->ngaps - 1] = b; fix_gaps(ss, ss->start, ss->stop); }
}
static void set_slow_score(orc_msite *ss, int x) {
    if (x <= 0) { ss->pairedScore = ss->slowScore = x; }
    else if (ss->pairedScore <= 0) { ss->slowScore = x; }
    else { if (ss->slowScore > 0) ss->pairedScore = x + (ss->pairedScore - ss->slowScore); else ss->pairedScore = x + 1; }
    ss->slowScore = x;
}

/* GapTools.calcGrefLen (:80-92) */
static int calc_gref_len(const orc_msite *ss) {
    int total = ss->stop - ss->start + 1;
    for (int i = 2; i < ss->ngaps; i += 2) {
        const int gap = ss->gaps[i] - ss->gaps[i - 1] - GAPBUFFER2;
        const int syms = imax(0, gap / GAPLEN);
        total -= syms * (GAPLEN - 1);
    }
    return total;
}

static void site_set_perfect(orc_msite *ss, const uint8_t *bases, int L, const orc_index *ix) {
    int32_t ps[2];
    orc_set_perfect(bases, L, ix->chromArr[ss->chrom], ix->chromArrLen[ss->chrom], ss->start, ss->stop, ps);
    ss->perfect = ps[0]; ss->semiperfect = ps[1];
}

/* ------------------------------------------------------------------ comparators and list tools */
static int cmp_score(const orc_msite *a, const orc_msite *b) {      /* SiteScore.compareTo :55-73 */
    int x = b->score - a->score; if (x) return x;
    x = b->slowScore - a->slowScore; if (x) return x;
    x = b->pairedScore - a->pairedScore; if (x) return x;
    x = b->quickScore - a->quickScore; if (x) return x;
    x = a->chrom - b->chrom; if (x) return x;
    return a->start - b->start;
}
static int cmp_pos(const orc_msite *a, const orc_msite *b) {        /* PositionComparator :379-395 */
    if (a->chrom != b->chrom) return a->chrom - b->chrom;
    if (a->start != b->start) return a->start - b->start;
    if (a->stop != b->stop) return a->stop - b->stop;
    if (a->strand != b->strand) return a->strand - b->strand;
    if (a->score != b->score) return b->score - a->score;
    if (a->slowScore != b->slowScore) return b->slowScore - a->slowScore;
    if (a->quickScore != b->quickScore) return b->quickScore - a->quickScore;
    if (a->perfect != b->perfect) return a->perfect ? -1 : 1;
    if (a->rescued != b->rescued) return a->rescued ? 1 : -1;
    return 0;
}
/* Collections.sort is a stable merge sort: any stable sort gives the same order */
static void sort_list(slist *l, int (*cmp)(const orc_msite *, const orc_msite *)) {
    for (int i = 1; i < l->n; i++) {
        orc_msite t = l->s[i];
        int j = i - 1;
        while (j >= 0 && cmp(&l->s[j], &t) > 0) { l->s[j + 1] = l->s[j]; j--; }
        l->s[j + 1] = t;
    }
}
static void condense(slist *l, const uint8_t *dead) {
    int m = 0;
    for (int i = 0; i < l->n; i++) if (!dead[i]) { if (m != i) l->s[m] = l->s[i]; m++; }
    l->n = m;
}

/* Tools.trimSitesBelowCutoff (Tools.java:1113-1161) */
static void trim_below_cutoff(slist *l, int cutoff, int retainPaired, int retainSemiperfect, int minRetain, int maxRetain) {
    if (l->n <= minRetain) return;
    while (l->n > maxRetain) l->n--;
    uint8_t dead[LISTCAP]; memset(dead, 0, (size_t)l->n);
    int removed = 0;
    const int maxToRemove = l->n - minRetain;
    for (int i = l->n - 1; i >= 0; i--) {
        const orc_msite *ss = &l->s[i];
        if (!retainSemiperfect || !ss->semiperfect) {
            if (ss->score < cutoff && (!retainPaired || ss->pairedScore <= 0)) {
                dead[i] = 1; removed++;
                if (removed >= maxToRemove) break;
            }
        }
    }
    if (removed > 0) condense(l, dead);
}
/* Tools.trimSiteList (Tools.java:654-674) */
static int trim_site_list(slist *l, float fractionOfMax, int retainPaired, int retainSemiperfect, int minRetain, int maxRetain) {
    if (l->n == 0) return -999999;
    if (l->n == 1) return l->s[0].score;
    int maxScore = -999999;
    if (minRetain > 1 && minRetain < l->n) maxScore = l->s[0].score;
    else for (int i = 0; i < l->n; i++) maxScore = imax(maxScore, l->s[i].score);
    const int cutoff = (int)((float)maxScore * fractionOfMax);
    trim_below_cutoff(l, cutoff, retainPaired, retainSemiperfect, minRetain, maxRetain);
    return maxScore;
}
/* BBMapThread.trimList, USE_AFFINE_SCORE branch (BBMapThread.java:140-197) */
static int trim_list(slist *l, int retainPaired, int maxScore, int specialCasePerfect, int minRetain, int maxRetain) {
    if (l->n == 0) return -99999;
    if (l->n == 1) return l->s[0].score;
    const int highest = trim_site_list(l, .6f, retainPaired, 1, minRetain, maxRetain);
    if (highest == maxScore && specialCasePerfect) {
        trim_site_list(l, .94f, retainPaired, 1, minRetain, maxRetain);
        if (l->n > 8) trim_site_list(l, .99f, retainPaired, 1, minRetain, maxRetain);
        return highest;
    }
    const int mstr2 = (minRetain <= 1 ? 1 : minRetain + 1);
    if (l->n > 4) trim_site_list(l, .65f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 8) trim_site_list(l, .7f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 12) trim_site_list(l, .75f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 16) trim_site_list(l, .8f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 20) trim_site_list(l, .85f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 24) trim_site_list(l, .9f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 32) trim_site_list(l, .95f, retainPaired, 1, minRetain, maxRetain);
    if (l->n > 40) trim_site_list(l, .97f, retainPaired, 1, mstr2, maxRetain);
    if (l->n > 48) trim_site_list(l, .99f, retainPaired, 1, mstr2, maxRetain);
    return highest;
}

static int positional_match(const orc_msite *a, const orc_msite *b, int testGaps) {    /* SiteScore.java:353-365 */
    if (a->chrom != b->chrom || a->strand != b->strand || a->start != b->start || a->stop != b->stop) return 0;
    if (!testGaps || (a->ngaps == 0 && b->ngaps == 0)) return 1;
    if ((a->ngaps == 0) != (b->ngaps == 0)) return 0;
    if (a->ngaps != b->ngaps) return 0;
    for (int i = 0; i < a->ngaps; i++) if (a->gaps[i] != b->gaps[i]) return 0;
    return 1;
}
static int max3(int a, int b, int c) { return imax(a, imax(b, c)); }
/* Tools.mergeDuplicateSites(list, doAssertions, mergeDifferentGaps=true) (Tools.java:697-759) */
static void merge_duplicate_sites(slist *l) {
    if (l->n < 2) return;
    sort_list(l, cmp_pos);
    uint8_t dead[LISTCAP]; memset(dead, 0, (size_t)l->n);
    int removed = 0, ai = 0;
    for (int i = 1; i < l->n; i++) {
        orc_msite *a = &l->s[ai], *b = &l->s[i];
        const int exact = positional_match(a, b, 1);
        if (exact || positional_match(a, b, 0)) {
            const orc_msite *better = a;
            if (!exact) {
                if (a->score != b->score) better = (a->score > b->score ? a : b);
                else if (a->slowScore != b->slowScore) better = (a->slowScore > b->slowScore ? a : b);
                else if (a->pairedScore != b->pairedScore) better = (a->pairedScore > b->pairedScore ? a : b);
            }
            int g[ORC_MAX_GAPS], ng = better->ngaps;
            memcpy(g, better->gaps, sizeof g);
            set_slow_score(a, imax(a->slowScore, b->slowScore));
            a->pairedScore = (a->pairedScore <= a->slowScore && b->pairedScore <= a->slowScore) ? 0 : max3(0, a->pairedScore, b->pairedScore);
            a->score = imax(a->score, b->score);
            a->perfect = (a->perfect || b->perfect);
            a->semiperfect = (a->semiperfect || b->semiperfect);
            if (!exact) { a->ngaps = ng; memcpy(a->gaps, g, sizeof g); }
            removed++; dead[i] = 1;
        } else ai = i;
    }
    if (removed) condense(l, dead);
}

/* Tools.removeLowQualitySitesPaired (Tools.java:934-960) */
static void remove_low_quality_paired(slist *l, int maxSwScore, float multSingle, float multPaired) {
    if (l->n == 0) return;
    const int thresh = (int)((float)maxSwScore * multSingle), threshPaired = (int)((float)maxSwScore * multPaired);
    if (l->s[0].score < threshPaired) { l->n = 0; return; }
    uint8_t dead[LISTCAP]; memset(dead, 0, (size_t)l->n);
    for (int i = l->n - 1; i >= 0; i--) {
        const orc_msite *ss = &l->s[i];
        if (ss->pairedScore > 0) { if (ss->slowScore < threshPaired) dead[i] = 1; }
        else if (ss->slowScore < thresh) dead[i] = 1;
    }
    condense(l, dead);
}

/* ------------------------------------------------------------------ quickMap tail */
static void complement_into(uint8_t *dst, const uint8_t *src, int L) {
    for (int i = 0; i < L; i++) {
        uint8_t b = src[L - 1 - i], c;
        switch (b) { case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break; case 'N': c = 'N'; break; default: c = 0xFF; }
        dst[i] = c;
    }
}

/* what quickMap hands to the index for one read: AbstractMapThread.java:659-736 (offsets, key scores and base scores come from the
 * read's qualities there; here they are inputs).  bs == NULL: all-zero base scores (a read without qualities, QualityTools.java:164-181) */
typedef struct { const uint8_t *bp; int L; const int8_t *bs; const int32_t *offsets, *keyScores; int nkeys; } readin;

static void quick_map(mapper *M, const readin *in, const uint8_t *bm, slist *out) {
    const uint8_t *bp = in->bp; const int L = in->L;
    out->n = 0;
    if (L < M->ix->p.k || in->nkeys < 1) return;                          /* AbstractMapThread.java:646, :701 */
    static __thread orc_site raw[LISTCAP];
    int8_t *zero = NULL;
    const int8_t *bs = in->bs;
    if (!bs) { zero = (int8_t *)calloc((size_t)L + 1, 1); bs = zero; }
    const int ns = orc_index_find(M->ix, bp, bm, L, bs, in->keyScores, in->offsets, in->nkeys, raw, LISTCAP, NULL);
    free(zero);
    const int expLimit = (M->P->alignColumns * 17) / 20 - (2 * (M->P->slowAlignPadding + 10));     /* EXPECTED_LEN_LIMIT :92 */
    for (int i = 0; i < ns; i++) {
        orc_msite ss; memset(&ss, 0, sizeof ss);
        ss.chrom = raw[i].chrom; ss.strand = raw[i].strand; ss.start = raw[i].start; ss.stop = raw[i].stop; ss.hits = raw[i].hits;
        ss.quickScore = ss.score = raw[i].score; ss.perfect = raw[i].perfect; ss.semiperfect = raw[i].semiperfect;
        ss.ngaps = raw[i].ngaps; memcpy(ss.gaps, raw[i].gaps, sizeof ss.gaps);
        ss.match_job = -1;
        /* removeOutOfBounds (AbstractMapThread.java:2444-2476); cha.maxIndex = last array index */
        const int mx = M->ix->chromArrLen[ss.chrom] - 1;
        if (ss.start < 0 || ss.stop > mx) continue;
        if (calc_gref_len(&ss) >= expLimit) { set_stop(&ss, ss.start + imin(L + 40, expLimit)); if (ss.ngaps) fix_gaps(&ss, ss.start, ss.stop); }
        out->s[out->n++] = ss;
    }
}

/* ------------------------------------------------------------------ scoreNoIndels (AbstractMapThread.java:762-856) */
static int score_no_indels_list(mapper *M, slist *l, const uint8_t *bp, const uint8_t *bm, int L, int maxSw, int maxImp) {
    int near = 0, force = 0;
    for (int j = 0; j < l->n; j++) {
        orc_msite *ss = &l->s[j];
        const int oldScore = ss->score;
        const uint8_t *bases = ss->strand ? bm : bp;
        if (ss->perfect) { near++; set_slow_score(ss, maxSw); ss->score = maxSw; ss->ngaps = 0; }
        else {
            const uint8_t *c = M->ix->chromArr[ss->chrom]; const int cl = M->ix->chromArrLen[ss->chrom];
            int sw = orc_score_no_indels(bases, L, c, cl, NULL, ss->start);
            if (sw < oldScore && oldScore >= maxImp && ss->stop - ss->start + 1 != L) {
                const int sw2 = orc_score_no_indels(bases, L, c, cl, NULL, ss->stop - L + 1);
                if (sw2 >= maxImp) { sw = sw2; set_start(ss, ss->stop - L + 1); site_set_perfect(ss, bases, L, M->ix); }
            }
            set_slow_score(ss, sw); ss->score = sw;
            if (sw >= maxImp) {
                near++;
                set_stop(ss, ss->start + L - 1); ss->ngaps = 0;
                if (sw >= maxSw) ss->perfect = ss->semiperfect = 1;
                else site_set_perfect(ss, bases, L, M->ix);
            } else if (oldScore >= maxImp) force = 1;
        }
    }
    return force ? -near : near;
}

/* ------------------------------------------------------------------ findTipDeletions (AbstractMapThread.java:2178-2292, 1107-1141) */
static int tip_right(const orc_index *ix, const uint8_t *bases, int L, int chrom, int originalStop, int searchDist, int tiplen) {
    const uint8_t *ref = ix->chromArr[chrom]; const int reflen = ix->chromArrLen[chrom], minIndex = 0;
    if (originalStop < minIndex + tiplen - 1) return 0;
    int minMismatches, bestStart = originalStop;
    const int tipCoord = L - 1;
    int lastMismatch = 0, originalMismatches = 0, contig = 0;
    for (int i = 0; i < tiplen && contig < 5; i++) {
        if (bases[tipCoord - i] != ref[originalStop - i]) { originalMismatches++; lastMismatch = i; contig = 0; } else contig++;
    }
    if (originalMismatches < 3) return 0;
    minMismatches = originalMismatches;
    tiplen = lastMismatch + 1;
    if (tiplen < 4) return 0;
    searchDist = imin(searchDist, 30 * originalMismatches);
    const int lastIndexToStart = imin(reflen - 1, originalStop + searchDist);
    for (int start = originalStop + 1; start <= lastIndexToStart && minMismatches > 0; start++) {
        int mismatches = 0;
        for (int j = 0; j < tiplen && mismatches < minMismatches; j++) if (bases[tipCoord - j] != ref[start - j]) mismatches++;
        if (mismatches < minMismatches) { bestStart = start; minMismatches = mismatches; }
    }
    if (minMismatches > 2 || originalMismatches - minMismatches < 2) return 0;
    return bestStart - originalStop;
}
static int tip_left(const orc_index *ix, const uint8_t *bases, int chrom, int originalStart, int searchDist, int tiplen) {
    const uint8_t *ref = ix->chromArr[chrom]; const int reflen = ix->chromArrLen[chrom], minIndex = 0;
    if (originalStart + tiplen >= reflen) return 0;
    if (minIndex >= originalStart) return 0;
    int minMismatches, bestStart = originalStart;
    int lastMismatch = 0, originalMismatches = 0, contig = 0;
    for (int i = 0; i < tiplen && contig < 5; i++) {
        if (bases[i] != ref[originalStart + i]) { originalMismatches++; lastMismatch = i; contig = 0; } else contig++;
    }
    if (originalMismatches < 3) return 0;
    minMismatches = originalMismatches;
    tiplen = lastMismatch + 1;
    if (tiplen < 4) return 0;
    searchDist = imin(searchDist, 16 + 16 * originalMismatches + 8 * tiplen);
    const int lastIndexToStart = imax(minIndex, originalStart - searchDist);
    for (int start = originalStart - 1; start >= lastIndexToStart && minMismatches > 0; start--) {
        int mismatches = 0;
        for (int j = 0; j < tiplen && mismatches < minMismatches; j++) if (bases[j] != ref[start + j]) mismatches++;
        if (mismatches < minMismatches) { bestStart = start; minMismatches = mismatches; }
    }
    if (minMismatches > 2 || originalMismatches - minMismatches < 2) return 0;
    return originalStart - bestStart;
}
static int find_tip_deletions_site(mapper *M, orc_msite *ss, const uint8_t *bases, int L, int maxImp, int lookRight, int lookLeft) {
    if (ss->slowScore >= maxImp) return 0;
    if (L <= 2 * TIP_DELETION_MAX_TIPLEN) return 0;
    const orc_map_params *P = M->P;
    int maxSearch = P->tipSearchDist;
    maxSearch = imin(maxSearch, P->alignColumns - (P->slowRescuePadding + 8 + imax(L, ss->stop - ss->start)));
    if (maxSearch < 1) return 0;
    int changed = 0;
    if (lookRight) {
        const int x = tip_right(M->ix, bases, L, ss->chrom, ss->stop, maxSearch, TIP_DELETION_MAX_TIPLEN);
        if (x > 0) {
            set_stop(ss, ss->stop + x); changed = 1;
            maxSearch = imin(maxSearch, P->alignColumns - (P->slowRescuePadding + 8 + imax(L, ss->stop - ss->start)));
            if (maxSearch < 1) return changed;
        }
    }
    if (lookLeft) {
        const int y = tip_left(M->ix, bases, ss->chrom, ss->start, maxSearch, TIP_DELETION_MAX_TIPLEN);
        if (y > 0) { set_start(ss, ss->start - y); changed = 1; }
    }
    return changed;
}
/* findTipDeletions(Read ...) (:1075-1105); reads carry no qualities: findRight = findLeft = true */
static void find_tip_deletions_list(mapper *M, slist *l, const uint8_t *bp, const uint8_t *bm, int L, int maxSw, int maxImp) {
    for (int i = 0; i < l->n; i++) {
        orc_msite *ss = &l->s[i];
        const uint8_t *bases = ss->strand ? bm : bp;
        if (!ss->semiperfect && ss->slowScore < maxImp) {
            if (find_tip_deletions_site(M, ss, bases, L, maxImp, 1, 1)) {
                ss->match_job = -1;
                set_slow_score(ss, orc_score_no_indels(bases, L, M->ix->chromArr[ss->chrom], M->ix->chromArrLen[ss->chrom], NULL, ss->start));
                if (ss->slowScore == maxSw) { set_stop(ss, ss->start + L - 1); ss->perfect = ss->semiperfect = 1; }
                else { ss->perfect = 0; site_set_perfect(ss, bases, L, M->ix); }
            }
        }
    }
}

/* ------------------------------------------------------------------ one MSA.fillAndScoreLimited call (+ the recorded traceback) */
static int fill_and_score(mapper *M, int which, int kind, const uint8_t *bases, int L, const orc_msite *ss, int pad, int minscore,
                          int32_t *sc, int *jobOut) {
    const uint8_t *c = M->ix->chromArr[ss->chrom]; const int cl = M->ix->chromArrLen[ss->chrom];
    int32_t mx[4];
    const int a0 = ss->start - pad, b0 = ss->stop + pad;
    const int64_t it0 = M->msa->iterationsLimited + M->msa->iterationsUnlimited;
    int gcopy[ORC_MAX_GAPS]; memcpy(gcopy, ss->gaps, sizeof gcopy);
    const int n = orc_fill_and_score_limited(M->msa, bases, L, c, cl, a0, b0, minscore, ss->ngaps ? gcopy : NULL, ss->ngaps, sc, mx);
    const int64_t it = M->msa->iterationsLimited + M->msa->iterationsUnlimited - it0;
    M->dpJobs++; M->cells += it;
    int mlen = 0;
    if (n) {
        const int a = a0 < 0 ? 0 : a0;
        int b = b0 > cl - 1 ? cl - 1 : b0;
        if (ss->ngaps == 0 && b - a >= M->msa->maxColumns) b = imin(cl - 1, a + M->msa->maxColumns - 1);     /* MSA.java:118-121 */
        mlen = orc_traceback(M->msa, bases, c, a, b, mx[0], mx[1], mx[2], ss->ngaps ? 1 : 0, M->tb, M->tbcap);
        if (mlen < 0) mlen = 0;
    }
    *jobOut = -1;
    if (M->log) {
        const int64_t k = __sync_fetch_and_add(M->nlog, 1);
        if (k < M->logcap) {
            orc_mjob *j = &M->log[k]; memset(j, 0, sizeof *j);
            j->read = (int32_t)M->readIdx[which]; j->seq = M->seq[which]; j->kind = kind; j->strand = ss->strand; j->chrom = ss->chrom;
            j->refStartLoc = a0; j->refEndLoc = b0; j->minScore = minscore; j->ngaps = ss->ngaps;
            j->score_len = n; for (int i = 0; i < n; i++) j->score[i] = sc[i];
            j->iterations = it; j->match_len = mlen;
            if (M->match && mlen > 0 && mlen <= M->matchStride) memcpy(M->match + k * (int64_t)M->matchStride, M->tb, (size_t)mlen);
            *jobOut = (int)k;
        }
    }
    M->seq[which]++;
    return n;
}

/* ------------------------------------------------------------------ scoreSlow (BBMapThread.java:252-386) */
static void score_slow(mapper *M, int which, slist *l, const uint8_t *bp, const uint8_t *bm, int L, int maxSw, int maxImp, int paired) {
    const orc_map_params *P = M->P;
    const float R = P->minRatio;
    const float ratio = paired ? orc_ratio_pre_rescue(R) : R;
    const int CZ1e = CLEARZONE1E, CZ3 = P->clearzone3;
    int minMsaLimit = -CZ1e + (int)(ratio * (float)maxSw);
    const int expLimit = (P->alignColumns * 17) / 20 - (2 * (P->slowAlignPadding + 10));
    for (int i = 0; i < l->n; i++) {
        orc_msite *ss = &l->s[i];
        const uint8_t *bases = ss->strand ? bm : bp;
        if (ss->stop - ss->start != L - 1) { set_slow_score(ss, 0); ss->semiperfect = 0; ss->perfect = 0; }      /* :278-284 */
        const int swNoIndel = ss->slowScore;
        int32_t sc[8]; int n = 0, job = -1;
        if (swNoIndel < maxImp && !ss->semiperfect) {
            const int expectedLen = calc_gref_len(ss);
            if (expectedLen >= expLimit) set_stop(ss, ss->start + imin(L + 40, expLimit));
            int pad = P->slowAlignPadding;
            const int minscore = imax(swNoIndel, minMsaLimit);
            n = fill_and_score(M, which, 0, bases, L, ss, pad, minscore, sc, &job);
            if (n > 6 && (sc[3] + sc[4] + expectedLen < expLimit)) {                                             /* :312-335 */
                int32_t old[8]; memcpy(old, sc, sizeof old); const int oldn = n, oldjob = job;
                set_limits(ss, ss->start - sc[6], ss->stop + sc[7]);
                pad = P->slowAlignPadding + P->extraPadding;
                n = fill_and_score(M, which, 1, bases, L, ss, pad, minscore, sc, &job);
                if (n == 0 || sc[0] < old[0]) { memcpy(sc, old, sizeof old); n = oldn; job = oldjob; }
            }
        }
        if (n) { set_slow_score(ss, sc[0]); set_limits(ss, sc[1], sc[2]); ss->match_job = job; }
        ss->score = ss->slowScore;
        minMsaLimit = imax(minMsaLimit, ss->slowScore - CZ3);
        ss->perfect = (ss->slowScore == maxSw);
        if (ss->perfect) ss->semiperfect = 1;
        else if (!ss->semiperfect) site_set_perfect(ss, bases, L, M->ix);
    }
}

/* ------------------------------------------------------------------ pairSiteScoresInitial (BBMapThread.java:736-940) */
static void pair_site_scores_initial(mapper *M, slist *l1, slist *l2, int len1, int len2, int trim) {
    const orc_map_params *P = M->P;
    if (l1->n < 1 || l2->n < 1) return;
    sort_list(l1, cmp_pos); sort_list(l2, cmp_pos);
    for (int i = 0; i < l1->n; i++) l1->s[i].pairedScore = 0;
    for (int i = 0; i < l2->n; i++) l2->s[i].pairedScore = 0;
    int maxPairedScore1 = -1, maxPairedScore2 = -1;
    const int ilimit = l1->n - 1, jlimit = l2->n - 1;
    const int maxReadLen = imax(len1, len2);
    const int outerDistLimit = (maxReadLen * OUTER_DIST_MULT) / OUTER_DIST_DIV;
    const int innerDistLimit = P->maxPairDist;
    const int expectedFragLength = P->averagePairDist + len1 + len2;
    int numPerfectPairs = 0;
    for (int i = 0, j = 0; i <= ilimit && j <= jlimit; i++) {
        orc_msite *ss1 = &l1->s[i], *ss2 = &l2->s[j];
        while (j < jlimit && (ss2->chrom < ss1->chrom || (ss2->chrom == ss1->chrom && ss1->start - ss2->stop > innerDistLimit))) { j++; ss2 = &l2->s[j]; }
        for (int k = j; k <= jlimit; k++) {
            ss2 = &l2->s[k];
            if (ss2->chrom > ss1->chrom) break;
            if (ss2->start - ss1->stop > innerDistLimit) break;
            int innerdist, outerdist;
            if (ss1->strand != ss2->strand) {                     /* REQUIRE_CORRECT_STRANDS_PAIRS = true */
                if (ss1->strand == 0) { innerdist = ss2->start - ss1->stop; outerdist = ss2->stop - ss1->start; }
                else { innerdist = ss1->start - ss2->stop; outerdist = ss1->stop - ss2->start; }
            } else {
                if (ss1->start <= ss2->start) { innerdist = ss2->start - ss1->stop; outerdist = ss2->stop - ss1->start; }
                else { innerdist = ss1->start - ss2->stop; outerdist = ss1->stop - ss2->start; }
            }
            if (outerdist >= outerDistLimit && innerdist <= innerDistLimit) {
                const int strandOK = ((ss1->strand == ss2->strand) == 0);      /* SAME_STRAND_PAIRS = false */
                if (strandOK) {                                               /* || !REQUIRE_CORRECT_STRANDS_PAIRS (true) */
                    int paired1 = 0, paired2 = 0;
                    const int deviation = iabsdif(P->averagePairDist, innerdist);
                    const int ps1 = ss1->score + 1 + imax(1, ss2->score / 2 - ((deviation * ss2->score) / (32 * expectedFragLength + 100)));
                    const int ps2 = ss2->score + 1 + imax(1, ss1->score / 2 - ((deviation * ss1->score) / (32 * expectedFragLength + 100)));
                    if (ps1 > ss1->pairedScore) { paired1 = 1; ss1->pairedScore = imax(ss1->pairedScore, ps1); maxPairedScore1 = imax(ss1->score, maxPairedScore1); }
                    if (ps2 > ss2->pairedScore) { paired2 = 1; ss2->pairedScore = imax(ss2->pairedScore, ps2); maxPairedScore2 = imax(ss2->score, maxPairedScore2); }
                    if (paired1 && paired2 && outerdist >= maxReadLen && deviation <= expectedFragLength && ss1->perfect && ss2->perfect) numPerfectPairs++;
                }
            }
        }
    }
    for (int i = 0; i < l1->n; i++) if (l1->s[i].pairedScore > l1->s[i].score) l1->s[i].score = l1->s[i].pairedScore;
    for (int i = 0; i < l2->n; i++) if (l2->s[i].pairedScore > l2->s[i].score) l2->s[i].score = l2->s[i].pairedScore;
    if (trim) {
        if (numPerfectPairs > 0) {
            trim_below_cutoff(l1, (int)((float)maxPairedScore1 * .94f), 0, 1, 1, P->maxTrimSitesToRetain);
            trim_below_cutoff(l2, (int)((float)maxPairedScore2 * .94f), 0, 1, 1, P->maxTrimSitesToRetain);
        } else {
            if (l1->n > 4) trim_below_cutoff(l1, (int)((float)maxPairedScore1 * .9f), 1, 1, 1, P->maxTrimSitesToRetain);
            if (l2->n > 4) trim_below_cutoff(l2, (int)((float)maxPairedScore2 * .9f), 1, 1, 1, P->maxTrimSitesToRetain);
        }
    }
}

/* ------------------------------------------------------------------ slowRescue (AbstractMapThread.java:1246-1306) */
static void slow_rescue(mapper *M, int which, const uint8_t *bases, int L, orc_msite *ss, int maxScore, int maxImp, int tipR, int tipL) {
    const orc_map_params *P = M->P;
    const uint8_t *c = M->ix->chromArr[ss->chrom]; const int cl = M->ix->chromArrLen[ss->chrom];
    int sw = orc_score_no_indels(bases, L, c, cl, NULL, ss->start);
    const int oldStart = ss->start;
    if (sw < maxImp && M->ix->p.maxIndel > 0) {
        set_slow_score(ss, sw);
        if (tipR || tipL) {
            if (find_tip_deletions_site(M, ss, bases, L, maxImp, tipR, tipL)) { ss->match_job = -1; sw = orc_score_no_indels(bases, L, c, cl, NULL, ss->start); }
        }
        const int minMsaLimit = -CLEARZONE1E + (int)(orc_ratio_paired(P->minRatio) * (float)maxScore);
        const int minscore = imax(sw, minMsaLimit);
        int32_t sc[8]; int job = -1;
        const int n = fill_and_score(M, which, 2, bases, L, ss, P->slowRescuePadding, minscore, sc, &job);
        if (n) { set_slow_score(ss, sc[0]); ss->score = ss->slowScore; set_start(ss, sc[1]); set_stop(ss, sc[2]); ss->match_job = job; }
        else { set_slow_score(ss, sw); ss->score = ss->slowScore; set_start(ss, oldStart); set_stop(ss, ss->start + L - 1); }
    } else { set_slow_score(ss, sw); ss->score = ss->slowScore; set_stop(ss, ss->start + L - 1); }
    ss->pairedScore = ss->score + 1;
    ss->perfect = (ss->slowScore == maxScore);
    if (ss->perfect) ss->semiperfect = 1; else site_set_perfect(ss, bases, L, M->ix);
}

/* rescue(anchor, loose, basesP, basesM, searchDist) (AbstractMapThread.java:1144-1243); basesP/basesM are the loose read's */
static void rescue(mapper *M, int whichLoose, slist *anchor, slist *loose, int anchorLen, const uint8_t *bp, const uint8_t *bm, int L, int searchDist) {
    const orc_map_params *P = M->P;
    if (searchDist > P->maxRescueDist) return;
    if (anchor->n == 0) return;
    const int maxLooseSw = orc_max_quality(L), maxAnchorSw = orc_max_quality(anchorLen), maxImp = orc_max_imperfect_score(L);
    const int bestLoose = loose->n == 0 ? 0 : loose->s[0].slowScore;
    const int bestAnchor = anchor->s[0].slowScore;
    if (bestLoose == maxLooseSw && bestAnchor == maxAnchorSw && anchor->s[0].pairedScore > 0) return;
    const int rescueScoreLimit = (int)(0.95f * (float)bestAnchor);
    const int retainScoreLimit = imax((int)(0.68f * (float)bestLoose), (int)(0.4f * (float)maxLooseSw));
    const int retainScoreLimit2 = imax((int)(0.95f * (float)bestLoose), (int)(0.55f * (float)maxLooseSw));
    const int maxMismatches = (bestLoose > maxImp) ? 5 : imin(P->maxRescueMismatches, (int)(0.60f * (float)L - 1.0f));
    const int findTip = (P->tipSearchDist > 0) && bestLoose < maxImp;
    const int nAnchor = anchor->n;                                  /* the loop runs over the anchor's list; loose grows */
    for (int i = 0; i < nAnchor; i++) {
        orc_msite *ssa = &anchor->s[i];
        if (ssa->slowScore < rescueScoreLimit) break;
        if (ssa->pairedScore == 0 && !ssa->rescued) {
            const int searchIntoAnchor = ssa->stop - ssa->start - 1 + (anchorLen * 11 / 16);
            int loc, idealStart; const uint8_t *bases;
            const int strand = ssa->strand ^ 1;                     /* SAME_STRAND_PAIRS = false */
            const int searchRight = (strand == 1);
            if (ssa->strand == 0) { bases = bm; loc = ssa->stop - searchIntoAnchor; idealStart = ssa->stop + P->averagePairDist; }
            else { bases = bp; loc = ssa->start + searchIntoAnchor; idealStart = ssa->start - P->averagePairDist; }
            int32_t q[8];
            const uint8_t *c = M->ix->chromArr[ssa->chrom]; const int cl = M->ix->chromArrLen[ssa->chrom];
            orc_quick_rescue(bases, L, c, cl, 0, loc, searchDist + searchIntoAnchor, searchRight, idealStart, maxMismatches, QR_MATCH, QR_MATCH2, 1, 100, q);
            M->rescueScans++;
            if (q[0]) {
                orc_msite ss; memset(&ss, 0, sizeof ss);
                ss.chrom = ssa->chrom; ss.strand = strand; ss.start = q[1]; ss.stop = q[2]; ss.hits = 0;
                ss.quickScore = ss.score = q[3]; ss.perfect = q[5]; ss.semiperfect = q[6]; ss.rescued = 1; ss.match_job = -1;
                const int mxI = cl - 1;
                if (ss.start >= 0 && ss.stop <= mxI) {              /* isInBounds */
                    const int mismatches = q[4];
                    set_slow_score(&ss, 0);
                    if (mismatches <= maxMismatches) {
                        slow_rescue(M, whichLoose, bases, L, &ss, maxLooseSw, maxImp, findTip, findTip);
                        if (ss.score > retainScoreLimit && ss.start >= 0 && ss.stop <= mxI) {
                            if (ss.score > retainScoreLimit2) {
                                ss.pairedScore = imax(ss.pairedScore, ss.slowScore + ssa->slowScore / 4);
                                ssa->pairedScore = imax(ssa->pairedScore, ssa->slowScore + ss.slowScore / 4);
                            }
                            if (loose->n < LISTCAP) loose->s[loose->n++] = ss;
                        }
                    }
                }
            }
        }
    }
}

#include "final_stage.inc"

/* ------------------------------------------------------------------ processRead (BBMapThread.java:389-490; with finalStage to :732) */
static void process_read(mapper *M, const readin *in, slist *l, orc_final *fin, uint8_t *fmatch, int fstride) {
    const uint8_t *bp = in->bp; const int L = in->L;
    uint8_t *bm = (uint8_t *)malloc((size_t)L + 1);
    complement_into(bm, bp, L);
    quick_map(M, in, bm, l);
    const int maxSw = orc_max_quality(L), maxImp = orc_max_imperfect_score(L);
    if (M->P->trimList && l->n > 1) { sort_list(l, cmp_score); trim_list(l, 0, maxSw, 1, MIN_TRIM_SITES_TO_RETAIN_SINGLE, M->P->maxTrimSitesToRetain); }
    if (l->n > 0) {
        const int near = score_no_indels_list(M, l, bp, bm, L, maxSw, maxImp);
        sort_list(l, cmp_score);
        if (near < 1 && M->P->tipSearchDist > 0) find_tip_deletions_list(M, l, bp, bm, L, maxSw, maxImp);
        if (near < 1) score_slow(M, 0, l, bp, bm, L, maxSw, maxImp, 0);
    }
    if (l->n > 0) { merge_duplicate_sites(l); sort_list(l, cmp_score); }
    if (M->P->finalStage) {
        rstate r; r_init(&r, bp, bm, L, M->readIdx[0]);
        final_single(M, &r, l);
        store_final(M, &r, l, fin, fmatch, fstride);
        m_reset(M);
    }
    free(bm);
}

/* ------------------------------------------------------------------ processReadPair (BBMapThread.java:943-1098) */
static void process_pair_tail(mapper *M, const uint8_t *bp1, const uint8_t *bm1, const uint8_t *bp2, const uint8_t *bm2, int L, slist *l1, slist *l2);
static void process_pair(mapper *M, const readin *in1, const readin *in2, slist *l1, slist *l2, orc_final *fin1, orc_final *fin2,
                         uint8_t *fmatch1, uint8_t *fmatch2, int fstride) {
    const orc_map_params *P = M->P;
    const uint8_t *bp1 = in1->bp, *bp2 = in2->bp;
    const int L = in1->L;                                   /* both mates have one length in every caller of this restatement */
    uint8_t *bm1 = (uint8_t *)malloc((size_t)L + 1), *bm2 = (uint8_t *)malloc((size_t)L + 1);
    complement_into(bm1, bp1, L); complement_into(bm2, bp2, L);
    quick_map(M, in1, bm1, l1);
    quick_map(M, in2, bm2, l2);
    const int maxSw = orc_max_quality(L), maxImp = orc_max_imperfect_score(L);
    pair_site_scores_initial(M, l1, l2, L, L, P->trimList);
    if (P->trimList) {
        if (l1->n > MIN_TRIM_SITES_TO_RETAIN_PAIRED) sort_list(l1, cmp_score);
        if (l2->n > MIN_TRIM_SITES_TO_RETAIN_PAIRED) sort_list(l2, cmp_score);
        trim_list(l1, 1, maxSw, 0, MIN_TRIM_SITES_TO_RETAIN_PAIRED, P->maxTrimSitesToRetain);
        trim_list(l2, 1, maxSw, 0, MIN_TRIM_SITES_TO_RETAIN_PAIRED, P->maxTrimSitesToRetain);
    }
    for (int i = 0; i < l1->n; i++) l1->s[i].score = l1->s[i].quickScore;
    for (int i = 0; i < l2->n; i++) l2->s[i].score = l2->s[i].quickScore;
    slist *ls[2] = {l1, l2}; const uint8_t *bps[2] = {bp1, bp2}, *bms[2] = {bm1, bm2};
    for (int w = 0; w < 2; w++) {
        slist *l = ls[w];
        if (l->n > 0) {
            const int near = score_no_indels_list(M, l, bps[w], bms[w], L, maxSw, maxImp);
            sort_list(l, cmp_score);
            if (near < 1 && P->tipSearchDist > 0) find_tip_deletions_list(M, l, bps[w], bms[w], L, maxSw, maxImp);
            score_slow(M, w, l, bps[w], bms[w], L, maxSw, maxImp, 1);
            merge_duplicate_sites(l);
        }
    }
    process_pair_tail(M, bp1, bm1, bp2, bm2, L, l1, l2);
    if (P->finalStage) {
        rstate r1, r2; r_init(&r1, bp1, bm1, L, M->readIdx[0]); r_init(&r2, bp2, bm2, L, M->readIdx[1]);
        final_pair(M, &r1, &r2, l1, l2);
        store_final(M, &r1, l1, fin1, fmatch1, fstride); store_final(M, &r2, l2, fin2, fmatch2, fstride);
        m_reset(M);
    }
    free(bm1); free(bm2);
}
static void process_pair_tail(mapper *M, const uint8_t *bp1, const uint8_t *bm1, const uint8_t *bp2, const uint8_t *bm2, int L, slist *l1, slist *l2) {
    const orc_map_params *P = M->P;
    const int maxSw = orc_max_quality(L);
    if (P->doRescue) {
        int unpaired1 = 0, unpaired2 = 0;
        for (int i = 0; i < l1->n; i++) if (l1->s[i].pairedScore == 0) unpaired1++;
        for (int i = 0; i < l2->n; i++) if (l2->s[i].pairedScore == 0) unpaired2++;
        const float pre = orc_ratio_pre_rescue(P->minRatio);
        const int searchDist = imin(P->maxPairDist, 2 * P->averagePairDist + 100);
        if (unpaired1 > 0 && l1->n > 0) {
            sort_list(l1, cmp_score);
            remove_low_quality_paired(l1, maxSw, pre, pre);
            rescue(M, 1, l1, l2, L, bp2, bm2, L, searchDist);
            merge_duplicate_sites(l2);
        }
        if (unpaired2 > 0 && l2->n > 0) {
            sort_list(l2, cmp_score);
            remove_low_quality_paired(l2, maxSw, pre, pre);
            rescue(M, 0, l2, l1, L, bp1, bm1, L, searchDist);
            merge_duplicate_sites(l1);
        }
    }
}

/* ------------------------------------------------------------------ driver */
float orc_ratio_paired(float R) { const float a = R * .80f, b = 1.0f - ((1.0f - R) * 1.4f); return a > b ? a : b; }       /* AbstractMapThread.java:106 */
float orc_ratio_pre_rescue(float R) { const float a = R * .60f, b = 1.0f - ((1.0f - R) * 1.8f); return a > b ? a : b; }   /* :107 */

void orc_map_default_params(orc_map_params *P) {
    memset(P, 0, sizeof *P);
    P->maxPairDist = 32000; P->averagePairDist = 100; P->maxRescueDist = 1200; P->maxRescueMismatches = 32;
    P->maxTrimSitesToRetain = 800; P->trimList = 1; P->doRescue = 1; P->clearzone3 = 800; P->extraPadding = 10;
#ifdef ORC_PACBIO   /* the final stage is restated for BBMapThread only (BBMapThreadPacBio's tail differs: clearzone rule, applyClearzone3 call) */
    P->finalStage = 0;
#else
    P->finalStage = 1;
#endif
#ifdef ORC_PACBIO   /* BBMapPacBio.setDefaults (BBMapPacBio.java:47-69), BBMapThreadPacBio.java:27-28, BBIndexPacBio.java:2462 */
    P->minRatio = 0.46f; P->slowAlignPadding = 8; P->slowRescuePadding = 16; P->tipSearchDist = 15;
    P->alignColumns = 7600; P->msaMaxRows = 6020; P->msaMaxColumns = 7600;
#else               /* BBMap.setDefaults (BBMap.java:45-65), BBMapThread.java:27-28 */
    P->minRatio = 0.56f; P->slowAlignPadding = 4; P->slowRescuePadding = 8; P->tipSearchDist = 100;
    P->alignColumns = 3000; P->msaMaxRows = 601; P->msaMaxColumns = 3000;
#endif
}

typedef struct {
    const orc_index *ix; const orc_map_params *P;
    const orc_read *recs; const uint8_t *bases; const int8_t *baseScores; const int32_t *keyinfo;
    int64_t n; int paired;         /* n = reads (single-ended) or pairs; pair p = records 2p, 2p+1 */
    int cap; orc_msite *sites; int32_t *nsites;
    orc_mjob *log; int64_t logcap; volatile int64_t *nlog; uint8_t *match; int matchStride;
    volatile int64_t *next;
    int64_t dpJobs, cells, rescueScans, mapped, finalFills;
    orc_final *fin; uint8_t *fmatch; int fstride;
} drv_arg;

static void store_list(const slist *l, orc_msite *out, int32_t *nout, int cap) {
    if (!out) return;
    if (l->n > cap) { *nout = -1; return; }
    *nout = l->n;
    memcpy(out, l->s, sizeof(orc_msite) * (size_t)l->n);
    for (int i = 0; i < l->n; i++) out[i].reserved[0] = out[i].reserved[1] = 0;     /* the final stage's arena ids mean nothing outside */
}

static readin read_in(const drv_arg *w, int64_t r) {
    const orc_read *rec = &w->recs[r];
    readin in;
    in.bp = w->bases + rec->bases_off; in.L = rec->len; in.bs = w->baseScores ? w->baseScores + rec->bases_off : NULL;
    in.offsets = w->keyinfo + rec->keys_off; in.keyScores = in.offsets + rec->nkeys; in.nkeys = rec->nkeys;
    return in;
}

static void *drv_worker(void *p) {
    drv_arg *w = (drv_arg *)p;
    mapper M; memset(&M, 0, sizeof M);
    M.ix = w->ix; M.P = w->P; M.msa = orc_msa_new(w->P->msaMaxRows, w->P->msaMaxColumns);
    M.tbcap = w->P->msaMaxRows + w->P->msaMaxColumns + 64 + 128 * 64; M.tb = (uint8_t *)malloc((size_t)M.tbcap);
    M.log = w->log; M.logcap = w->logcap; M.nlog = w->nlog; M.match = w->match; M.matchStride = w->matchStride;
    slist *l1 = (slist *)malloc(sizeof(slist)), *l2 = (slist *)malloc(sizeof(slist));
    const int64_t chunk = w->recs[0].len > 1000 ? 1 : 64;
    for (;;) {
        const int64_t i0 = __sync_fetch_and_add(w->next, chunk);
        if (i0 >= w->n) break;
        const int64_t hi = i0 + chunk < w->n ? i0 + chunk : w->n;
        for (int64_t r = i0; r < hi; r++) {
            M.seq[0] = M.seq[1] = 0;
            if (w->paired) {
                M.readIdx[0] = 2 * r; M.readIdx[1] = 2 * r + 1;
                const readin a = read_in(w, 2 * r), b = read_in(w, 2 * r + 1);
                process_pair(&M, &a, &b, l1, l2, w->fin ? &w->fin[2 * r] : NULL, w->fin ? &w->fin[2 * r + 1] : NULL,
                             w->fmatch ? w->fmatch + 2 * r * (int64_t)w->fstride : NULL, w->fmatch ? w->fmatch + (2 * r + 1) * (int64_t)w->fstride : NULL, w->fstride);
                store_list(l1, w->sites ? w->sites + 2 * r * w->cap : NULL, w->nsites ? &w->nsites[2 * r] : NULL, w->cap);
                store_list(l2, w->sites ? w->sites + (2 * r + 1) * w->cap : NULL, w->nsites ? &w->nsites[2 * r + 1] : NULL, w->cap);
                w->mapped += (l1->n > 0) + (l2->n > 0);
            } else {
                M.readIdx[0] = r;
                const readin a = read_in(w, r);
                process_read(&M, &a, l1, w->fin ? &w->fin[r] : NULL, w->fmatch ? w->fmatch + r * (int64_t)w->fstride : NULL, w->fstride);
                store_list(l1, w->sites ? w->sites + r * w->cap : NULL, w->nsites ? &w->nsites[r] : NULL, w->cap);
                w->mapped += (l1->n > 0);
            }
        }
    }
    w->dpJobs = M.dpJobs; w->cells = M.cells; w->rescueScans = M.rescueScans; w->finalFills = M.finalFills;
    free(l1); free(l2); free(M.tb); free(M.ms); orc_msa_free(M.msa);
    return NULL;
}

/* Maps n_reads read records (paired: records 2p and 2p+1 are mates of equal length).  A record addresses its bases (and, when
 * baseScores != NULL, its base scores) at bases_off and its keys at keyinfo[keys_off]: offsets[nkeys] then keyScores[nkeys] -- the
 * layout of bbidx_read (include/bbmap_amd.h).  sites: n_reads x cap records (may be NULL: timing only).  log (optional): one record
 * per fillAndScoreLimited call; match: logcap x matchStride bytes.  stats4 = {DP calls, visited cells, quickRescue scans, reads
 * with at least one site}.  Returns elapsed seconds (-1: bad argument). */
double orc_map_reads(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired,
                     const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo, int cap,
                     orc_msite *sites, int32_t *nsites,
                     orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4) {
    return orc_map_reads_final(ix, P, recs, n_reads, paired, bases, baseScores, keyinfo, cap, sites, nsites, log, logcap, nlog, match, matchStride,
                               threads, stats4, NULL, NULL, 0);
}
/* The same with the final stage's per-read records (P->finalStage): fin[n_reads], fmatch = n_reads x fstride bytes (the read's match string,
 * when it fits; fin[r].match_len is its length either way).  stats4[0] counts the final stage's fills too. */
double orc_map_reads_final(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired,
                           const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo, int cap,
                           orc_msite *sites, int32_t *nsites,
                           orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4,
                           orc_final *fin, uint8_t *fmatch, int fstride) {
    if (threads < 1) threads = 1;
    if (n_reads < 1 || (paired && (n_reads & 1))) return -1.0;
    for (int64_t r = 0; r < n_reads; r++) if (recs[r].len > P->msaMaxRows - 1) return -1.0;      /* maxReadLength() = ALIGN_ROWS - 1 */
    if (paired) for (int64_t r = 0; r < n_reads; r += 2) if (recs[r].len != recs[r + 1].len) return -1.0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    drv_arg *wa = (drv_arg *)calloc((size_t)threads, sizeof(drv_arg));
    volatile int64_t next = 0, nl = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        drv_arg *a = &wa[t];
        a->ix = ix; a->P = P; a->recs = recs; a->bases = bases; a->baseScores = baseScores; a->keyinfo = keyinfo;
        a->n = paired ? n_reads / 2 : n_reads; a->paired = paired;
        a->cap = cap; a->sites = sites; a->nsites = nsites;
        a->log = log; a->logcap = logcap; a->nlog = &nl; a->match = match; a->matchStride = matchStride; a->next = &next;
        a->fin = fin; a->fmatch = fmatch; a->fstride = fstride;
        pthread_create(&th[t], NULL, drv_worker, a);
    }
    int64_t s[4] = {0, 0, 0, 0};
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); s[0] += wa[t].dpJobs; s[1] += wa[t].cells; s[2] += wa[t].rescueScans; s[3] += wa[t].mapped; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (nlog) *nlog = nl;
    if (stats4) memcpy(stats4, s, sizeof s);
    free(th); free(wa);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The uniform special case (every read L bases, one set of offsets / key scores for all, no base scores): n reads (reads2 == NULL)
 * or n pairs.  Site lists come back per mate as before. */
double orc_map_batch(const orc_index *ix, const orc_map_params *P, const uint8_t *reads1, const uint8_t *reads2, int64_t n, int L,
                     const int32_t *offsets, const int32_t *keyScores, int nkeys, int cap,
                     orc_msite *sites1, int32_t *nsites1, orc_msite *sites2, int32_t *nsites2,
                     orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4) {
    const int paired = reads2 != NULL;
    const int64_t nr = paired ? 2 * n : n;
    orc_read *recs = (orc_read *)malloc(sizeof(orc_read) * (size_t)nr);
    uint8_t *bases = (uint8_t *)malloc((size_t)nr * (size_t)L + 1);
    int32_t *keyinfo = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)nkeys + 4);
    memcpy(keyinfo, offsets, sizeof(int32_t) * (size_t)nkeys); memcpy(keyinfo + nkeys, keyScores, sizeof(int32_t) * (size_t)nkeys);
    for (int64_t r = 0; r < nr; r++) {
        const uint8_t *src = paired ? ((r & 1) ? reads2 + (r / 2) * L : reads1 + (r / 2) * L) : reads1 + r * L;
        memcpy(bases + r * L, src, (size_t)L);
        recs[r].bases_off = r * L; recs[r].keys_off = 0; recs[r].len = L; recs[r].nkeys = nkeys;
    }
    orc_msite *sites = NULL; int32_t *nsites = NULL;
    if (sites1) { sites = (orc_msite *)malloc(sizeof(orc_msite) * (size_t)nr * (size_t)cap); nsites = (int32_t *)malloc(sizeof(int32_t) * (size_t)nr); }
    const double t = orc_map_reads(ix, P, recs, nr, paired, bases, NULL, keyinfo, cap, sites, nsites, log, logcap, nlog, match, matchStride, threads, stats4);
    if (sites) {
        for (int64_t r = 0; r < nr; r++) {
            orc_msite *dst = paired ? ((r & 1) ? sites2 : sites1) + (r / 2) * cap : sites1 + r * cap;
            int32_t *nd = paired ? ((r & 1) ? nsites2 : nsites1) + (r / 2) : nsites1 + r;
            *nd = nsites[r];
            if (nsites[r] > 0) memcpy(dst, sites + r * cap, sizeof(orc_msite) * (size_t)nsites[r]);
        }
    }
    free(sites); free(nsites); free(recs); free(bases); free(keyinfo);
    return t;
}

/* The final alignment stage ALONE over given site lists (the counterpart of bbmap_final_batch_device): sites = n_reads x cap records with
 * nsites[r] in use, replaced by the lists after the stage; fills are numbered from 0 per read.  Single-threaded. */
int orc_final_reads(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired, const uint8_t *bases,
                    int cap, orc_msite *sites, int32_t *nsites, orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride,
                    orc_final *fin, uint8_t *fmatch, int fstride) {
    if (n_reads < 1 || (paired && (n_reads & 1)) || !sites || !nsites || !fin) return -1;
    mapper M; memset(&M, 0, sizeof M);
    volatile int64_t nl = 0;
    M.ix = ix; M.P = P; M.msa = orc_msa_new(P->msaMaxRows, P->msaMaxColumns);
    M.tbcap = P->msaMaxRows + P->msaMaxColumns + 64 + 128 * 64; M.tb = (uint8_t *)malloc((size_t)M.tbcap);
    M.log = log; M.logcap = logcap; M.nlog = &nl; M.match = match; M.matchStride = matchStride;
    slist *ls[2] = {(slist *)malloc(sizeof(slist)), (slist *)malloc(sizeof(slist))};
    const int per = paired ? 2 : 1;
    for (int64_t u = 0; u < n_reads / per; u++) {
        rstate rs[2]; uint8_t *bm[2] = {NULL, NULL};
        M.seq[0] = M.seq[1] = 0;
        for (int w = 0; w < per; w++) {
            const int64_t r = per * u + w;
            const int L = recs[r].len;
            M.readIdx[w] = r;
            bm[w] = (uint8_t *)malloc((size_t)L + 1);
            complement_into(bm[w], bases + recs[r].bases_off, L);
            r_init(&rs[w], bases + recs[r].bases_off, bm[w], L, r);
            ls[w]->n = nsites[r] > 0 ? nsites[r] : 0;
            memcpy(ls[w]->s, sites + r * cap, sizeof(orc_msite) * (size_t)ls[w]->n);
        }
        if (paired) final_pair(&M, &rs[0], &rs[1], ls[0], ls[1]); else final_single(&M, &rs[0], ls[0]);
        for (int w = 0; w < per; w++) {
            const int64_t r = per * u + w;
            store_final(&M, &rs[w], ls[w], &fin[r], fmatch ? fmatch + r * (int64_t)fstride : NULL, fstride);
            store_list(ls[w], sites + r * cap, &nsites[r], cap);
            free(bm[w]);
        }
        m_reset(&M);
    }
    if (nlog) *nlog = nl;
    free(ls[0]); free(ls[1]); free(M.tb); free(M.ms); orc_msa_free(M.msa);
    return 0;
}
