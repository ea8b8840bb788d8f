/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  See index_oracle.h for scope and parity status
 * ("restatement only": the reference path is pure Java and cannot run here).
 *
 * Follows, function by function (paths relative to /root/reference/current):
 *   align2/IndexMaker4.java:303-421      index construction (count, prefix sum, fill)
 *   align2/BBIndex.java:101-191          analyzeIndex (COUNTS, clumpy keys, length histogram)
 *   align2/Tools.java:1797-1850          makeLengthHistogram3/4
 *   align2/BBIndex.java:394-639          findAdvanced / find
 *   align2/BBIndex.java:266-350          trimExcessHitListsByGreedy  + align2/Solver.java:46-151
 *   align2/BBIndex.java:642-741          prescanAllBlocks
 *   align2/BBIndex.java:2294-2450        findMaxQscore2
 *   align2/BBIndex.java:1219-1706        slowWalk3
 *   align2/BBIndex.java:2490-2511,2882-3035  quickScore, scoreZ2, scoreLeft/Right, maxQuickScore
 *   align2/AbstractIndex.java:52-78      scoreY
 *   align2/BBIndex.java:2558-2833        extendScore  (affine score: MultiStateAligner11tsJNI.java:871-1027)
 *   align2/BBIndex.java:2837-2878        makeGapArray
 *   align2/BBIndex.java:3267-3294        calcApproxHitsCutoff
 *   stream/SiteScore.java:239-292        setPerfect
 * Not restated: GapTools.fixGaps (only reached when a site that already carries a gap array is widened by a
 * later overlapping site; the gap array's end points are moved, its interior is left as is).
 */
#include "index_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "msa11ts_oracle.h"

#define KEYBUF 2048                 /* BBIndexPacBio.HEAP_LENGTH 2047 (BBIndexPacBio.java:2395); BBIndex keeps 256 (BBIndex.java:3102) */
#define BASE_HIT_SCORE 100
#define Y_SCORE_MULT 10
#define MINGAP 256
#ifdef ORC_PACBIO
/* align2.BBIndexPacBio: the same class with other tunables (a diff of BBIndexPacBio.java against BBIndex.java shows constants, the
 * thresholds of find()'s "too few hits" retries and array sizes; BBIndex's camelWalk3 is dead code, USE_CAMELWALK=false).
 * BBIndexPacBio.java:2475 (Z_SCORE_MULT), :2538 (SMALL_GENOME_LIST), :2527, :2523, :2590-2593, :76-77 (INDEL_PENALTY), :2545-2546 */
#define Z_SCORE_MULT 25
#define SMALL_GENOME_LIST 80
#define MIN_HIT_LISTS_TO_RETAIN 12
static const float HIT_FRACTION_TO_RETAIN = 0.97f;
static const float MIN_SCORE_MULT = 0.02f;
static const float MIN_QSCORE_MULT = 0.005f, MIN_QSCORE_MULT2 = 0.005f;
static const float DYNAMIC_SCORE_THRESH = 0.64f, DYNAMIC_QSCORE_THRESH = 0.6f, DYNAMIC_QSCORE_THRESH_PERFECT = 0.8f;
#define INDEL_PENALTY_OF(bkhs) (((bkhs) / 8) - 1)
#define INDEL_PENALTY_MULT 25
#define RELAX1 20
#define RELAX2 18
#define RELAX3 16
#define RELAX4 14
#define CLUMPY_MIN_LENGTH_INDEX 2800
#define CLUMPY_FRACTION 0.8f
#define DEF_MAX_INDEL 100
#define DEF_MAX_INDEL2 800
#define DEF_MAX_HITS_REDUCTION2 3
#define DEF_MAXIMUM_MAX_HITS_REDUCTION 6
#define DEF_HIT_REDUCTION_DIV 4
#define MAX_HITS_REDUCTION_PERFECT 2     /* BBIndexPacBio.java:2557 */
#else
#define Z_SCORE_MULT 20
#define SMALL_GENOME_LIST 20
#define MIN_HIT_LISTS_TO_RETAIN 6
static const float HIT_FRACTION_TO_RETAIN = 0.85f;
static const float MIN_SCORE_MULT = 0.15f;          /* USE_AFFINE_SCORE */
static const float MIN_QSCORE_MULT = 0.025f, MIN_QSCORE_MULT2 = 0.1f;
static const float DYNAMIC_SCORE_THRESH = 0.84f, DYNAMIC_QSCORE_THRESH = 0.6f, DYNAMIC_QSCORE_THRESH_PERFECT = 0.8f;
#define INDEL_PENALTY_OF(bkhs) (((bkhs) / 2) - 1)   /* BBIndex.java:75-76 */
#define INDEL_PENALTY_MULT 20
#define RELAX1 4                    /* BBIndex.java:424-436 */
#define RELAX2 3
#define RELAX3 3
#define RELAX4 2
#define CLUMPY_MIN_LENGTH_INDEX 2000
#define CLUMPY_FRACTION 0.75f
#define DEF_MAX_INDEL 16000
#define DEF_MAX_INDEL2 32000
#define DEF_MAX_HITS_REDUCTION2 2
#define DEF_MAXIMUM_MAX_HITS_REDUCTION 3
#define DEF_HIT_REDUCTION_DIV 5
#define MAX_HITS_REDUCTION_PERFECT 0     /* BBIndex.java:3262 */
#endif
#define PRESCAN_QSCORE_THRESH (DYNAMIC_QSCORE_THRESH * .95f)

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int absdif(int a, int b) { return a > b ? a - b : b - a; }

static inline int base_num(uint8_t b) {          /* AminoAcid.baseToNumber */
    switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; default: return -1; }
}
static int rc_key(int kmer, int k) {              /* AminoAcid.reverseComplementBinaryFast */
    int out = 0;
    for (int i = 0; i < k; i++) { out = (out << 2) | ((~kmer) & 3); kmer >>= 2; }
    return out;
}

/* ------------------------------------------------------------------------------------ build */
static int key_at(const uint8_t *arr, int a, int k) {    /* ChromosomeArray.getNumber(a,b) */
    int out = 0;
    for (int i = a; i < a + k; i++) { const int x = base_num(arr[i]); if (x < 0) return -1; out = (out << 2) | x; }
    return out;
}

orc_index *orc_index_build(int k, int chromBits, int nchroms, const uint8_t **chromArr, const int32_t *chromArrLen,
                           float fractionToExclude) {
    orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
    const int keyspace = 1 << (2 * k);
    const int cpb = 1 << chromBits, shift = 31 - chromBits, lowMask = cpb - 1;
    ix->p.k = k; ix->p.chromBits = chromBits; ix->p.minChrom = 1; ix->p.maxChrom = nchroms;
    ix->nchroms = nchroms;
    ix->chromArr = (const uint8_t **)calloc((size_t)nchroms + 1, sizeof(uint8_t *));
    ix->chromArrLen = (int32_t *)calloc((size_t)nchroms + 1, sizeof(int32_t));
    ix->chromLengths = (int32_t *)calloc((size_t)nchroms + 1, sizeof(int32_t));
    for (int c = 1; c <= nchroms; c++) { ix->chromArr[c] = chromArr[c]; ix->chromArrLen[c] = chromArrLen[c]; ix->chromLengths[c] = chromArrLen[c]; }
    ix->nblocks = (nchroms >> chromBits) + 1;            /* blocks are index[baseChrom]; chrom 0 does not exist */
    ix->starts = (int32_t **)calloc((size_t)ix->nblocks, sizeof(int32_t *));
    ix->sites = (int32_t **)calloc((size_t)ix->nblocks, sizeof(int32_t *));
    ix->numSites = (int64_t *)calloc((size_t)ix->nblocks, sizeof(int64_t));
    const int banmask = ~((-1) << (2 * k - 4));
    for (int b = 0; b < ix->nblocks; b++) {
        int32_t *starts = (int32_t *)calloc((size_t)keyspace + 1, sizeof(int32_t));
        int32_t *sizes = (int32_t *)calloc((size_t)keyspace, sizeof(int32_t));
        for (int pass = 0; pass < 2; pass++) {
            for (int chrom = imax(1, b * cpb); chrom <= imin(nchroms, b * cpb + cpb - 1); chrom++) {
                const uint8_t *arr = chromArr[chrom];
                const int maxIndex = chromArrLen[chrom] - 1;
                const int max = maxIndex - k + 1;
                for (int a = 0; a < max; a++) {
                    const uint8_t f = arr[a];
                    if (f != 'A' && f != 'C' && f != 'G' && f != 'T') continue;   /* array[a]==idb for one of the 4 threads */
                    const int key = key_at(arr, a, k);
                    if (key >= 0 && (key >> 4) != (key & banmask)) {
                        if (pass == 0) sizes[key]++;
                        else { ix->sites[b][sizes[key]] = ((chrom & lowMask) << shift) | a; sizes[key]++; }
                    }
                }
            }
            if (pass == 0) {
                int64_t sum = 0;
                for (int key = 0; key < keyspace; key++) { starts[key] = (int32_t)sum; sum += sizes[key]; sizes[key] = starts[key]; }
                starts[keyspace] = (int32_t)sum;
                ix->numSites[b] = sum;
                ix->sites[b] = (int32_t *)calloc((size_t)(sum > 0 ? sum : 1), sizeof(int32_t));
            }
        }
        free(sizes);
        ix->starts[b] = starts;
    }

    /* analyzeIndex */
    int32_t *COUNTS = (int32_t *)calloc((size_t)keyspace, sizeof(int32_t));
    int64_t *clumpsOf = (int64_t *)calloc((size_t)keyspace, sizeof(int64_t));   /* keyed by min(key, rc) */
    for (int b = 0; b < ix->nblocks; b++) {
        const int32_t *starts = ix->starts[b], *sites = ix->sites[b];
        for (int key = 0; key < keyspace; key++) {
            const int s1 = starts[key], e1 = starts[key + 1];
            int64_t v = (int64_t)COUNTS[key] + (e1 - s1);
            COUNTS[key] = (int32_t)(v > INT_MAX ? INT_MAX : v);
            int64_t clumps = 0;
            for (int i = s1 + 1; i < e1; i++) { const int dif = sites[i] - sites[i - 1]; if (dif > 0 && dif <= 5) clumps++; }
            if (clumps > 0) { const int r = rc_key(key, k); clumpsOf[key < r ? key : r] += clumps; }
        }
    }
    for (int key = 0; key < keyspace; key++) {
        const int r = rc_key(key, k);
        if (key < r) { int64_t x = (int64_t)COUNTS[key] + COUNTS[r]; if (x > INT_MAX) x = INT_MAX; COUNTS[key] = COUNTS[r] = (int32_t)x; }
    }
    for (int key = 0; key < keyspace; key++) {
        if (clumpsOf[key] > 0) {
            const int64_t clumps = clumpsOf[key], len = COUNTS[key];
            if (len > CLUMPY_MIN_LENGTH_INDEX && (float)clumps > CLUMPY_FRACTION * (float)len) { COUNTS[key] = 0; COUNTS[rc_key(key, k)] = 0; }
        }
    }
    free(clumpsOf);
    ix->counts = COUNTS;
    {   /* Tools.makeLengthHistogram3 -> 4 (the int product counts[ptr]*ptr wraps like the Java) */
        int max = 0;
        for (int i = 0; i < keyspace; i++) if (COUNTS[i] > max) max = COUNTS[i];
        int32_t *cnt = (int32_t *)calloc((size_t)max + 1, sizeof(int32_t));
        int64_t total = 0;
        for (int i = 0; i < keyspace; i++) { cnt[COUNTS[i]]++; total += COUNTS[i]; }
        if (total <= 0) { total = 0; for (int i = 1; i <= max; i++) total += (int64_t)i * cnt[i]; }
        int64_t sum = 0; int ptr = 0;
        for (int i = 0; i < 1000; i++) {
            const int64_t nextLimit = ((total * i) + 500) / 1000;
            while (ptr < max + 1 && sum < nextLimit) { sum += (int32_t)((uint32_t)cnt[ptr] * (uint32_t)ptr); ptr++; }
            ix->lengthHistogram[i] = imax(0, ptr - 1);
        }
        ix->lengthHistogram[1000] = max;
        free(cnt);
    }
    orc_index_params *p = &ix->p;
    p->maxIndel = DEF_MAX_INDEL; p->maxIndel2 = DEF_MAX_INDEL2; p->minApproxHitsToKeep = 1; p->kfilter = 0;
    p->quitAfterTwoPerfects = 1; p->prescanQscore = 1; p->trimByGreedy = 1; p->slow = 0;
    p->maxHitsReduction2 = DEF_MAX_HITS_REDUCTION2; p->maximumMaxHitsReduction = DEF_MAXIMUM_MAX_HITS_REDUCTION; p->hitReductionDiv = DEF_HIT_REDUCTION_DIV;
    const double f = fractionToExclude;                   /* BBIndex.setFractionToExclude: double arithmetic on a float */
    p->maxAverageListToSearch = (int)(1000 * (1 - 2.3 * f));
    p->maxAverageListToSearch2 = (int)(1000 * (1 - 1.4 * f));
    p->maxShortestListToSearch = (int)(1000 * (1 - 2.8 * f));
    {
        const int i1 = (int)((1 - fractionToExclude) * 1000.0f);
        const int i2 = (int)((1 - fractionToExclude * 0.25f) * 1000.0f);
        p->maxUsableLength = imax(2 * SMALL_GENOME_LIST, ix->lengthHistogram[i1]);
        p->maxUsableLength2 = imax(6 * SMALL_GENOME_LIST, ix->lengthHistogram[i2]);
        const float q = (-50 * 4000.0f) / (float)imax(2 * SMALL_GENOME_LIST, ix->lengthHistogram[p->maxAverageListToSearch]);
        p->pointsPerSite = (int)floor((double)q);
        if (p->pointsPerSite == 0) p->pointsPerSite = -1;
    }
    return ix;
}

void orc_index_free(orc_index *ix) {
    if (!ix) return;
    for (int b = 0; b < ix->nblocks; b++) { free(ix->starts[b]); free(ix->sites[b]); }
    free(ix->starts); free(ix->sites); free(ix->numSites); free(ix->counts);
    free((void *)ix->chromArr); free(ix->chromArrLen); free(ix->chromLengths);
    free(ix);
}

/* An index whose arrays the caller already holds (e.g. exported from a device build that the tests have shown equal to
 * orc_index_build array by array): nothing is copied, orc_index_free_view frees only the shell. */
orc_index *orc_index_from_arrays(const orc_index_params *p, int nblocks, int32_t **starts, int32_t **sites, const int64_t *numSites,
                                 int32_t *counts, const int32_t *lengthHistogram, int nchroms, const uint8_t **chromArr,
                                 const int32_t *chromArrLen) {
    orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
    if (!ix) return NULL;
    ix->p = *p; ix->nblocks = nblocks; ix->nchroms = nchroms;
    ix->starts = (int32_t **)calloc((size_t)nblocks, sizeof(int32_t *));
    ix->sites = (int32_t **)calloc((size_t)nblocks, sizeof(int32_t *));
    ix->numSites = (int64_t *)calloc((size_t)nblocks, sizeof(int64_t));
    for (int b = 0; b < nblocks; b++) { ix->starts[b] = starts[b]; ix->sites[b] = sites[b]; ix->numSites[b] = numSites[b]; }
    ix->counts = counts;
    memcpy(ix->lengthHistogram, lengthHistogram, sizeof ix->lengthHistogram);
    ix->chromArr = (const uint8_t **)calloc((size_t)nchroms + 1, sizeof(uint8_t *));
    ix->chromArrLen = (int32_t *)calloc((size_t)nchroms + 1, sizeof(int32_t));
    ix->chromLengths = (int32_t *)calloc((size_t)nchroms + 1, sizeof(int32_t));
    for (int c = 1; c <= nchroms; c++) { ix->chromArr[c] = chromArr[c]; ix->chromArrLen[c] = chromArrLen[c]; ix->chromLengths[c] = chromArrLen[c]; }
    return ix;
}
void orc_index_free_view(orc_index *ix) {
    if (!ix) return;
    free(ix->starts); free(ix->sites); free(ix->numSites);
    free((void *)ix->chromArr); free(ix->chromArrLen); free(ix->chromLengths);
    free(ix);
}

/* ------------------------------------------------------------------------------------ offsets */
int orc_make_offsets(int readlen, int blocksize, float density, int minKeysDesired, int32_t *out, int cap) {
    if (readlen < blocksize) return 0;
    const int slots = readlen - blocksize + 1;
    int desired = (int)ceil((double)((readlen * density) / blocksize));     /* float product and quotient, then Math.ceil */
    desired = imax(minKeysDesired, desired);
    desired = imin(slots, desired);
    int maxKeys = desired;
    if (slots == 1 || maxKeys == 1) { out[0] = slots / 2; return 1; }
    if (slots == 2 || maxKeys == 2) { out[0] = 0; out[1] = slots - 1; return 2; }
    if (slots == 3 || maxKeys == 3) { out[0] = 0; out[1] = slots / 2; out[2] = slots - 1; return 3; }
    const int midslots = slots - 2;
    maxKeys = imin(maxKeys, slots);
    const int middles = imin(maxKeys - 2, midslots);
    if (middles + 2 > cap) return -1;
    float fspacing = midslots / (float)(middles + 1.0f);
    if (fspacing < 1.0f) fspacing = 1.0f;
    out[0] = 0; out[middles + 1] = slots - 1;
    for (int i = 1; i <= middles; i++) out[i] = (int)floorf(fspacing * i + 0.5f);   /* Math.round(float) */
    if (middles > 2) { out[1] = (int)fspacing; out[middles] = (int)ceil((double)(fspacing * middles)); }
    return middles + 2;
}

/* ------------------------------------------------------------------------------------ site codec */
typedef struct {
    const orc_index *ix;
    int shift, siteMask, lowMask, highMask, cpb;
} codec;
static inline int to_number(const codec *c, int site, int chrom) { return ((chrom & c->lowMask) << c->shift) | site; }
static inline int number_to_chrom(const codec *c, int number, int baseChrom) { return (int)((uint32_t)number >> c->shift) + (baseChrom & c->highMask); }
static inline int number_to_site(const codec *c, int number) { return number & c->siteMask; }
static inline int base_chrom(const codec *c, int chrom) { return imax(0, chrom & c->highMask); }

static inline int key_count(const orc_index *ix, int key) { return ix->counts[key]; }
static inline int block_len(const orc_index *ix, int b, int key) {
    const int x = ix->starts[b][key + 1] - ix->starts[b][key];
    if (x == 0) return 0;
    return ix->sites[b][ix->starts[b][key]] != -1 ? x : 0;
}

static int calc_approx_hits_cutoff(const orc_index_params *p, int keys, int hits, int currentCutoff, int perfect) {
    const int mahtk = p->minApproxHitsToKeep;
    const int reduction = imin(imax(hits / p->hitReductionDiv, p->maxHitsReduction2), imax(p->maximumMaxHitsReduction, keys / 8));
    int r = hits - reduction;
    r = imax(mahtk, imax(currentCutoff, r));
    if (perfect) r = imax(r, keys - MAX_HITS_REDUCTION_PERFECT);
    return r;
}

/* ------------------------------------------------------------------------------------ key-level scores */
typedef struct {
    const orc_index *ix;
    const orc_index_params *p;
    codec c;
    int keylen, baseKeyHitScore, indelPenalty, indelPenaltyMult, maxPenalty, scoreZ1Key;
    int64_t *stats;
} walker;

static int max_score_z(const walker *w, const int *offsets, int n) {
    int score = 0, a0 = -1, b0 = -1;
    for (int i = 0; i < n; i++) { const int a = offsets[i]; if (b0 < a) { score += b0 - a0; a0 = a; } b0 = a + w->keylen; }
    score += b0 - a0;
    return score * Z_SCORE_MULT;
}
static int max_quick_score(const walker *w, const int *offsets, const int *keyScores, int n) {
    int x = 0;
    for (int i = 0; i < n; i++) x += keyScores[i];
    const int y = Y_SCORE_MULT * (offsets[n - 1] - offsets[0]);
    x += max_score_z(w, offsets, n);
    return x + y;
}
static int score_z2(const walker *w, const int *locs, int centerIndex, const int *offsets, int numApproxHits, int numHits) {
    if (numApproxHits == 1) return w->scoreZ1Key;
    const int center = locs[centerIndex];
    const int maxLoc = center + w->p->maxIndel2, minLoc = imax(0, center - w->p->maxIndel);
    int score = 0, a0 = -1, b0 = -1;
    for (int i = 0; i < numHits; i++) {
        const int loc = locs[i];
        if (loc >= minLoc && loc <= maxLoc) { const int a = offsets[i]; if (b0 < a) { score += b0 - a0; a0 = a; } b0 = a + w->keylen; }
    }
    score += b0 - a0;
    return score * Z_SCORE_MULT;
}
static int score_side(const walker *w, const int *locs, const int *keyScores, int centerIndex, int numHits, int dir) {
    int score = 0, prev, loc = locs[centerIndex];
    for (int i = centerIndex + dir; i >= 0 && i < numHits; i += dir) {
        if (locs[i] >= 0) {
            prev = loc; loc = locs[i];
            const int offset = absdif(loc, prev);
            if (offset <= w->p->maxIndel) {
                score += keyScores[i];
                if (offset != 0) score -= imin(w->indelPenalty + w->indelPenaltyMult * offset, w->maxPenalty);
            } else loc = prev;
        }
    }
    return score;
}
static int score_y(const int *locs, int centerIndex, const int *offsets, int n) {
    const int center = locs[centerIndex];
    int rightIndex = -1;
    for (int i = n - 1; rightIndex < centerIndex; i--) if (locs[i] == center) rightIndex = i;
    return offsets[rightIndex] - offsets[centerIndex];
}
static int quick_score(const walker *w, const int *locs, const int *keyScores, int centerIndex, const int *offsets,
                       int numApproxHits, int numHits) {
    if (numApproxHits == 1) return keyScores[centerIndex];
    const int x = keyScores[centerIndex] + score_side(w, locs, keyScores, centerIndex, numHits, -1)
                + score_side(w, locs, keyScores, centerIndex, numHits, +1) - centerIndex;
    return x + Y_SCORE_MULT * score_y(locs, centerIndex, offsets, numHits);
}

/* ------------------------------------------------------------------------------------ list walking state */
typedef struct {
    int n;                    /* lists (after shrink) */
    int row[KEYBUF], stop[KEYBUF], value[KEYBUF], offs[KEYBUF], kscore[KEYBUF];
    int live[KEYBUF];         /* still in the heap */
    int nlive;
    int hp[KEYBUF], hn;       /* the heap: indices of the live lists */
    const int32_t *sites;
} lists;

/* QuadHeap (align2/QuadHeap.java) ordered by Quad.compareTo = (site, column) (Quad.java:19-22): a binary heap of list
 * indices.  Only its minimum is ever observed, so any correct heap gives the reference's pop sequence. */
static inline int heap_less(const lists *L, int a, int b) { return L->value[a] < L->value[b] || (L->value[a] == L->value[b] && a < b); }
static void heap_sift_down(lists *L, int i) {
    const int n = L->hn; const int x = L->hp[i];
    for (;;) {
        int c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && heap_less(L, L->hp[c + 1], L->hp[c])) c++;
        if (!heap_less(L, L->hp[c], x)) break;
        L->hp[i] = L->hp[c]; i = c;
    }
    L->hp[i] = x;
}
static void heap_build(lists *L) {
    L->hn = 0;
    for (int i = 0; i < L->n; i++) if (L->live[i]) L->hp[L->hn++] = i;
    for (int i = L->hn / 2 - 1; i >= 0; i--) heap_sift_down(L, i);
}
/* heap.peek(): smallest (site, column) among live lists */
static inline int lists_peek(const lists *L) { return L->hn > 0 ? L->hp[0] : -1; }
static inline void heap_fix_top(lists *L) { heap_sift_down(L, 0); }                 /* the top's value grew */
static inline void heap_pop(lists *L) { L->hp[0] = L->hp[--L->hn]; if (L->hn > 0) heap_sift_down(L, 0); }

static inline int adjust_site(const walker *w, int a, int offset, int baseChrom) {
    if ((a & w->c.siteMask) >= offset) return a - offset;
    const int ch = number_to_chrom(&w->c, a, baseChrom), st = number_to_site(&w->c, a);
    return to_number(&w->c, imax(st - offset, 0), ch);
}
static void lists_init(const walker *w, lists *L, int block, const int *starts, const int *stops, const int *offsets,
                       const int *keyScores, int n, int baseChrom) {
    L->n = 0; L->sites = w->ix->sites[block];
    for (int i = 0; i < n; i++) {
        if (starts[i] < 0) continue;                       /* shrink(): drop keys with no list in this block */
        const int j = L->n++;
        L->row[j] = starts[i]; L->stop[j] = stops[i]; L->offs[j] = offsets[i]; L->kscore[j] = keyScores[i];
        L->value[j] = adjust_site(w, L->sites[starts[i]], offsets[i], baseChrom);
        L->live[j] = 1;
    }
    L->nlive = L->n;
    heap_build(L);
}
/* ------------------------------------------------------------------------------------ findMaxQscore2 */
static void find_max_qscore2(const walker *w, lists *L, int baseChrom, int prevMaxHits, int earlyExit, int perfectOnly,
                             int *outQ, int *outHits) {
    const orc_index_params *p = w->p;
    const int numHits = L->n;
    const int maxQuickScore = max_quick_score(w, L->offs, L->kscore, numHits);
    int topQscore = -999999999, maxHits = 0, approxHitsCutoff, indelCutoff;
    if (perfectOnly) { approxHitsCutoff = numHits; indelCutoff = 0; }
    else { approxHitsCutoff = imax(prevMaxHits, imin(p->minApproxHitsToKeep, numHits - 1)); indelCutoff = p->maxIndel2; }
    while (L->nlive > 0) {
        const int centerIndex = lists_peek(L);
        const int site = L->value[centerIndex];
        int approxHits = 0;
        {
            const int minsite = site - imin(p->maxIndel, indelCutoff), maxsite = site + p->maxIndel2;
            for (int column = 0, chances = numHits - approxHitsCutoff; column < numHits && chances >= 0; column++) {
                const int x = L->value[column];
                if (x >= minsite && x <= maxsite) approxHits++; else chances--;
            }
        }
        if (approxHits >= approxHitsCutoff) {
            int qscore = quick_score(w, L->value, L->kscore, centerIndex, L->offs, approxHits, numHits);
            qscore += score_z2(w, L->value, centerIndex, L->offs, approxHits, numHits);
            if (qscore > topQscore) {
                maxHits = imax(approxHits, maxHits);
                approxHitsCutoff = imax(approxHitsCutoff, approxHits - 1);
                topQscore = qscore;
                if (qscore >= maxQuickScore && earlyExit) { *outQ = topQscore; *outHits = maxHits; return; }
            }
        }
        for (;;) {
            const int col = lists_peek(L);
            if (col < 0 || L->value[col] != site) break;
            if (w->stats) w->stats[0]++;
            const int row = L->row[col] + 1;
            if (row < L->stop[col]) { L->row[col] = row; L->value[col] = adjust_site(w, L->sites[row], L->offs[col], baseChrom); heap_fix_top(L); }
            else {
                L->live[col] = 0; L->nlive--; heap_pop(L);
                if (earlyExit && (perfectOnly || L->nlive < approxHitsCutoff)) { *outQ = topQscore; *outHits = maxHits; return; }
            }
            if (L->nlive == 0) break;
        }
    }
    *outQ = topQscore; *outHits = maxHits;
}

/* ------------------------------------------------------------------------------------ extendScore */
static int extend_score(const walker *w, const uint8_t *bases, int blen, const int8_t *baseScores, const int *offsets,
                        const int *values, int chrom, int centerIndex, int *locArray, int numHits) {
    const orc_index_params *p = w->p;
    const int centerVal = values[centerIndex], centerLoc = number_to_site(&w->c, centerVal);
    const int minVal = centerVal - p->maxIndel, maxVal = centerVal + p->maxIndel2;
    const uint8_t *ref = w->ix->chromArr[chrom];
    const int reflen = w->ix->chromArrLen[chrom];
    const int K = w->keylen;
    if (w->stats) w->stats[2]++;
    for (int i = 0; i < blen; i++) locArray[i] = -1;
    for (int i = 0, keynum = 0; i < numHits; i++) {                     /* reverse fill, :2601-2643 */
        const int value = values[i];
        if (value >= minVal && value <= maxVal) {
            const int refbase = number_to_site(&w->c, value);
            keynum++;
            const int callbase = offsets[i];
            int misses = 0;
            for (int cloc = callbase + K - 1, rloc = refbase + cloc; cloc >= 0 && rloc >= 0 && rloc < reflen; cloc--, rloc--) {
                const int old = locArray[cloc];
                if (old == refbase) break;
                if (misses > 0 && old >= 0) break;
                if (w->stats) w->stats[3]++;
                if (bases[cloc] == ref[rloc]) { if (old < 0 || refbase == centerLoc) locArray[cloc] = refbase; }
                else { misses++; if (old >= 0 || keynum > 1) break; }
            }
        }
    }
    for (int i = 0; i < numHits; i++) {                                 /* forward fill, :2651-2679 */
        const int value = values[i];
        if (value >= minVal && value <= maxVal) {
            const int refbase = number_to_site(&w->c, value);
            const int callbase = offsets[i];
            int misses = 0;
            for (int cloc = callbase + K, rloc = refbase + cloc; cloc < blen && rloc < reflen; cloc++, rloc++) {
                const int old = locArray[cloc];
                if (old == refbase) break;
                if (misses > 0 && old >= 0) break;
                if (w->stats) w->stats[3]++;
                if (bases[cloc] == ref[rloc]) { if (old < 0 || refbase == centerLoc) locArray[cloc] = refbase; }
                else { misses++; if (old >= 0) break; }
            }
        }
    }
    for (int i = 0; i < blen; i++) if (bases[i] == 'N') locArray[i] = -2;
    return orc_calc_affine_score(locArray, blen, baseScores, p->kfilter);
}

/* makeGapArray, BBIndex.java:2837-2878 (destroys locArray) */
static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static int make_gap_array(int *locArray, int n, int minLoc, int minGap, int *out, int cap) {
    int gaps = 0, doSort = 0;
    if (locArray[0] < 0) locArray[0] = minLoc;
    for (int i = 1; i < n; i++) {
        if (locArray[i] < 0) locArray[i] = locArray[i - 1] + 1; else locArray[i] += i;
        if (locArray[i] < locArray[i - 1]) doSort = 1;
    }
    if (doSort) qsort(locArray, (size_t)n, sizeof(int), cmp_int);
    for (int i = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) gaps++;
    if (gaps < 1) return 0;
    const int len = 2 + gaps * 2;
    if (len > cap) return -1;
    out[0] = locArray[0]; out[len - 1] = locArray[n - 1];
    for (int i = 1, j = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) { out[j] = locArray[i - 1]; out[j + 1] = locArray[i]; j += 2; }
    return len;
}

/* SiteScore.setPerfect, stream/SiteScore.java:239-292 */
static void set_perfect(const orc_index *ix, orc_site *ss, const uint8_t *bases, int blen) {
    if (blen != ss->stop - ss->start + 1) { ss->perfect = 0; ss->semiperfect = 0; return; }
    const uint8_t *ref = ix->chromArr[ss->chrom];
    const int reflen = ix->chromArrLen[ss->chrom];
    int perfect = 1, semiperfect = 1;
    int refloc = ss->start, readloc = 0, N = 0;
    const int max = imin(ss->stop, reflen - 1), nlimit = blen / 2;
    if (ss->start < 0) { N -= ss->start; readloc -= ss->start; refloc -= ss->start; perfect = 0; }
    if (ss->stop >= reflen) { N += (ss->stop - reflen + 1); perfect = 0; }
    if (N > nlimit) { ss->perfect = ss->semiperfect = 0; return; }
    for (; refloc <= max; refloc++, readloc++) {
        const uint8_t c = bases[readloc], r = ref[refloc];
        if (c != r || c == 'N') {
            perfect = 0;
            if (c == 'N') semiperfect = 0;
            if (r != 'N' || (N = N + 1) > nlimit) { ss->perfect = perfect; ss->semiperfect = 0; return; }
        }
    }
    semiperfect = (semiperfect && (N <= nlimit));
    perfect = (perfect && semiperfect && (N == 0));
    ss->perfect = perfect; ss->semiperfect = semiperfect;
}
static inline int overlap(int a1, int b1, int a2, int b2) { return a2 <= b1 && b2 >= a1; }

/* ------------------------------------------------------------------------------------ slowWalk3 */
typedef struct { orc_site *v; int n, cap, overflow; } site_list;

static void slow_walk3(const walker *w, int block, const int *starts, const int *stops, const uint8_t *bases, int blen,
                       const int8_t *baseScores, const int *keyScores, const int *offsets, int numKeys,
                       int baseChrom_, int strand, site_list *ssl, int *bestScores, int allBasesCovered,
                       int maxScore, int fullyDefined) {
    const orc_index_params *p = w->p;
    const int maxQuickScore = max_quick_score(w, offsets, keyScores, numKeys);
    lists L;
    const int baseChrom = base_chrom(&w->c, baseChrom_);
    lists_init(w, &L, block, starts, stops, offsets, keyScores, numKeys, baseChrom);
    const int numHits = L.n;
    const int filter_by_qscore = (numKeys >= 5);
    const int minScore = (int)(MIN_SCORE_MULT * maxScore);
    const int minQuickScore = (int)(MIN_QSCORE_MULT * maxQuickScore);
    int *locArray = (int *)malloc(sizeof(int) * (size_t)blen);

    int currentTopScore = bestScores[0];
    int cutoff = imax(minScore, (int)(currentTopScore * DYNAMIC_SCORE_THRESH));
    int qcutoff = imax(bestScores[2], minQuickScore);
    int bestqscore = bestScores[3];
    int maxHits = bestScores[1];
    int perfectsFound = bestScores[5];
    int approxHitsCutoff = calc_approx_hits_cutoff(p, numKeys, maxHits, p->minApproxHitsToKeep, currentTopScore >= maxScore);
    if (approxHitsCutoff > numHits) { free(locArray); return; }
    const int shortCircuit = (allBasesCovered && numKeys == numHits && filter_by_qscore);
    if (currentTopScore >= maxScore) qcutoff = imax(qcutoff, (int)(maxQuickScore * DYNAMIC_QSCORE_THRESH_PERFECT));

    int prevIdx = -1;                       /* prevSS as an index into ssl */
    int finished = 0;
    while (L.nlive > 0 && !finished) {
        const int centerIndex = lists_peek(&L);
        const int site = L.value[centerIndex];
        int maxNearbySite = site, approxHits = 0;
        {
            const int minsite = site - p->maxIndel, maxsite = site + p->maxIndel2;
            for (int column = 0, chances = numHits - approxHitsCutoff; column < numHits && chances >= 0; column++) {
                const int x = L.value[column];
                if (x >= minsite && x <= maxsite) { if (x > maxNearbySite) maxNearbySite = x; approxHits++; } else chances--;
            }
        }
        if (approxHits >= approxHitsCutoff) {
            int score;
            int qscore = filter_by_qscore ? quick_score(w, L.value, L.kscore, centerIndex, L.offs, approxHits, numHits) : qcutoff;
            qscore += score_z2(w, L.value, centerIndex, L.offs, approxHits, numHits);
            int mapStart = site, mapStop = maxNearbySite;
            int locArrayValid = 0;
            if (qscore < qcutoff) score = -1;
            else {
                const int chrom = number_to_chrom(&w->c, site, baseChrom);
                if (shortCircuit && qscore == maxQuickScore) score = maxScore;
                else {
                    score = extend_score(w, bases, blen, baseScores, L.offs, L.value, chrom, centerIndex, locArray, numHits);
                    locArrayValid = 1;
                    int mn = INT_MAX, mx = INT_MIN;
                    for (int i = 0; i < blen; i++) { const int x = locArray[i]; if (x > -1) { if (x < mn) mn = x; if (x > mx) mx = x; } }
                    if (mn < 0 || mx < 0) score = -99999;
                    mapStart = to_number(&w->c, mn, chrom);
                    mapStop = to_number(&w->c, mx, chrom);
                }
                if (score == maxScore) {
                    qcutoff = imax(qcutoff, (int)(maxQuickScore * DYNAMIC_QSCORE_THRESH_PERFECT));
                    approxHitsCutoff = calc_approx_hits_cutoff(p, numKeys, maxHits, p->minApproxHitsToKeep, 1);
                }
                if (score >= cutoff) { qcutoff = imax(qcutoff, (int)(qscore * DYNAMIC_QSCORE_THRESH)); bestqscore = imax(qscore, bestqscore); }
            }
            if (score >= cutoff) {
                if (score > currentTopScore) {
                    maxHits = imax(approxHits, maxHits);
                    approxHitsCutoff = calc_approx_hits_cutoff(p, numKeys, maxHits, approxHitsCutoff, currentTopScore >= maxScore);
                    cutoff = imax(cutoff, (int)(score * DYNAMIC_SCORE_THRESH));
                    if (score >= maxScore) cutoff = imax(cutoff, (int)(score * 0.95f));
                    currentTopScore = score;
                }
                const int chrom = number_to_chrom(&w->c, mapStart, baseChrom);
                const int site2 = number_to_site(&w->c, mapStart);
                const int site3 = number_to_site(&w->c, mapStop) + blen - 1;
                int gapArr[16]; int ngaps = 0;
                if (site3 - site2 >= MINGAP + blen && locArrayValid) {
                    ngaps = make_gap_array(locArray, blen, site2, MINGAP, gapArr, 16);
                    if (ngaps < 0) ngaps = 0;
                    if (ngaps > 0) { gapArr[0] = imin(gapArr[0], site2); gapArr[ngaps - 1] = imax(gapArr[ngaps - 1], site3); }
                }
                orc_site ss; int haveSS = 0;
                const int perfect1 = (score == maxScore && fullyDefined);
                const int inbounds = (site2 >= 0 && site3 < w->ix->chromLengths[chrom]);
                orc_site *prevSS = prevIdx >= 0 ? &ssl->v[prevIdx] : NULL;
                if (inbounds && ngaps == 0 && prevSS && prevSS->chrom == chrom && prevSS->strand == strand &&
                    overlap(prevSS->start, prevSS->stop, site2, site3)) {
                    const int betterScore = imax(score, prevSS->score);
                    const int minStart = imin(prevSS->start, site2), maxStop = imax(prevSS->stop, site3);
                    const int perfect2 = (prevSS->score == maxScore && fullyDefined);
                    const int shortEnough = (maxStop - minStart < 2 * blen);
                    if (prevSS->start == site2 && prevSS->stop == site3) {
                        prevSS->score = betterScore;
                        prevSS->perfect = (prevSS->perfect || perfect1 || perfect2);
                        if (prevSS->perfect) prevSS->semiperfect = 1;
                    } else if (shortEnough && prevSS->start == site2 && !prevSS->semiperfect) {
                        if (perfect2) { }
                        else if (perfect1) {
                            prevSS->stop = site3; if (prevSS->ngaps) prevSS->gaps[prevSS->ngaps - 1] = site3;
                            if (!prevSS->perfect) perfectsFound++;
                            prevSS->perfect = prevSS->semiperfect = 1;
                        } else {
                            prevSS->stop = maxStop; if (prevSS->ngaps) prevSS->gaps[prevSS->ngaps - 1] = maxStop;
                            set_perfect(w->ix, prevSS, bases, blen);
                        }
                        prevSS->score = betterScore;
                    } else if (shortEnough && prevSS->stop == site3 && !prevSS->semiperfect) {
                        if (perfect2) { }
                        else if (perfect1) {
                            prevSS->start = site2; if (prevSS->ngaps) prevSS->gaps[0] = site2;
                            if (!prevSS->perfect) perfectsFound++;
                            prevSS->perfect = prevSS->semiperfect = 1;
                        } else {
                            prevSS->start = minStart; if (prevSS->ngaps) prevSS->gaps[0] = minStart;
                            set_perfect(w->ix, prevSS, bases, blen);
                        }
                        prevSS->score = betterScore;
                    } else {                                             /* SUBSUME_OVERLAPPING_SITES is false: class 5 */
                        memset(&ss, 0, sizeof ss);
                        ss.chrom = chrom; ss.strand = strand; ss.start = site2; ss.stop = site3; ss.hits = approxHits; ss.score = score;
                        ss.perfect = ss.semiperfect = perfect1;
                        if (!perfect1) set_perfect(w->ix, &ss, bases, blen);
                        haveSS = 1;
                    }
                } else if (inbounds) {
                    memset(&ss, 0, sizeof ss);
                    ss.chrom = chrom; ss.strand = strand; ss.start = site2; ss.stop = site3; ss.hits = approxHits; ss.score = score;
                    ss.perfect = ss.semiperfect = perfect1;
                    if (!perfect1) set_perfect(w->ix, &ss, bases, blen);
                    ss.ngaps = ngaps; for (int g = 0; g < ngaps; g++) ss.gaps[g] = gapArr[g];
                    haveSS = 1;
                }
                if (haveSS) {
                    if (ssl->n >= ssl->cap) { ssl->overflow = 1; finished = 1; }
                    else {
                        ssl->v[ssl->n] = ss;
                        const int idx = ssl->n++;
                        if (ss.perfect) {
                            const orc_site *pv = prevIdx >= 0 ? &ssl->v[prevIdx] : NULL;
                            if (!pv || !pv->perfect || !(pv->chrom == ss.chrom && pv->strand == ss.strand && overlap(ss.start, ss.stop, pv->start, pv->stop))) {
                                perfectsFound++;
                                if (p->quitAfterTwoPerfects && perfectsFound >= 2) { prevIdx = idx; break; }
                            }
                        }
                        prevIdx = idx;
                    }
                }
            }
        }
        /* advance every list whose head equals `site` (:1642-1691) */
        for (;;) {
            const int col = lists_peek(&L);
            if (col < 0 || L.value[col] != site) break;
            if (w->stats) w->stats[1]++;
            const int row = L.row[col] + 1;
            if (row < L.stop[col]) { L.row[col] = row; L.value[col] = adjust_site(w, L.sites[row], L.offs[col], baseChrom); heap_fix_top(&L); }
            else {
                L.live[col] = 0; L.nlive--; heap_pop(&L);
                if (L.nlive < approxHitsCutoff) { finished = 1; break; }
            }
            if (L.nlive == 0) break;
        }
    }
    bestScores[0] = imax(bestScores[0], currentTopScore);
    bestScores[1] = imax(bestScores[1], maxHits);
    bestScores[2] = imax(bestScores[2], qcutoff);
    bestScores[3] = imax(bestScores[3], bestqscore);
    bestScores[4] = maxQuickScore;
    bestScores[5] = perfectsFound;
    free(locArray);
}

/* ------------------------------------------------------------------------------------ Solver (greedy trim) */
static long long value_of_element(const int *offsets, int noffsets, const int *lengths, float keyWeight, int chunk,
                                  const int *lists_, int numlists, int index, long long pointsPerSite) {
    const long long POINTS_PER_LIST = 30000, POINTS_PER_BASE1 = 6000, BONUS_END = 40000, WIDTH = 5500, SPACING = -30;
    if (numlists < 1) return 0;
    const int prospect = lists_[index];
    if (lengths[prospect] == 0) return -999999;
    long long valuep = POINTS_PER_LIST + (POINTS_PER_LIST * 2 / numlists) + ((POINTS_PER_LIST * 10) / lengths[prospect]);
    const long long valuem = pointsPerSite * lengths[prospect];
    if (prospect == 0 || prospect == noffsets - 1) valuep += BONUS_END;
    if (numlists == 1) { valuep += (WIDTH + POINTS_PER_BASE1) * chunk; return ((long long)((float)valuep * keyWeight)) + valuem; }
    const int first = lists_[0], last = lists_[numlists - 1];
    const int offL = (prospect == first ? -1 : offsets[lists_[index - 1]]);
    const int offP = offsets[prospect];
    const int offR = (prospect == last ? offsets[noffsets - 1] + 1 : offsets[lists_[index + 1]]);
    const int oldL = offP - offL, oldR = offR - offP, newS = offR - offL;
    valuep += (long long)((oldL * oldL + oldR * oldR) - (newS * newS)) * SPACING;
    int uniquelyCovered;
    if (prospect == first) uniquelyCovered = offR - offP;
    else if (prospect == last) uniquelyCovered = offP - offL;
    else { const int b = offR - (offL + chunk); uniquelyCovered = b > 0 ? b : 0; }
    if (prospect == first || prospect == last) valuep += (POINTS_PER_BASE1 + WIDTH) * uniquelyCovered;
    else valuep += POINTS_PER_BASE1 * uniquelyCovered;
    return ((long long)((float)valuep * keyWeight)) + valuem;
}

static int trim_by_greedy(const orc_index *ix, const int *offsets, const int *keyScores, int n, int maxHitLists, int *keys,
                          int baseKeyHitScore) {
    const orc_index_params *p = &ix->p;
    float keyWeights[KEYBUF];
    const float inv = 1.0f / baseKeyHitScore;
    for (int i = 0; i < n; i++) keyWeights[i] = keyScores[i] * inv;
    const int limit = imax(SMALL_GENOME_LIST, ix->lengthHistogram[p->maxAverageListToSearch]) * n;
    const int limit2 = imax(SMALL_GENOME_LIST, ix->lengthHistogram[p->maxAverageListToSearch2]);
    const int limit3 = imax(SMALL_GENOME_LIST, ix->lengthHistogram[p->maxShortestListToSearch]);
    int sum = 0, initialHitCount = 0, shortest = INT_MAX - 1, shortest2 = INT_MAX;
    int lengths[KEYBUF];
    for (int i = 0; i < n; i++) {
        const int x = key_count(ix, keys[i]);
        lengths[i] = x; sum += x; initialHitCount += (x == 0 ? 0 : 1);
        if (x > 0 && x < shortest2) { shortest2 = x; if (shortest2 < shortest) { shortest2 = shortest; shortest = x; } }
    }
    if (initialHitCount < p->minApproxHitsToKeep) return initialHitCount;
    if (shortest > limit3 && !p->slow) { for (int i = 0; i < n; i++) keys[i] = -1; return 0; }
    int hitsCount = initialHitCount;
    const long long EARLY_TERMINATION_SCORE = -50LL * 2000;
    while (hitsCount >= p->minApproxHitsToKeep && (sum > limit || sum / initialHitCount > limit2 || hitsCount > maxHitLists)) {
        int lists_[KEYBUF];
        for (int i = 0, j = 0; j < hitsCount; i++) if (lengths[i] > 0) lists_[j++] = i;
        long long min = LLONG_MAX; int worstIndex = -1; long long worstValue64 = 0; int early = 0;
        for (int i = 0; i < hitsCount; i++) {
            const long long value = value_of_element(offsets, n, lengths, keyWeights[i], p->k, lists_, hitsCount, i, p->pointsPerSite);
            if (value < min) {
                if (min < EARLY_TERMINATION_SCORE && i != 0) { worstIndex = i; worstValue64 = value; early = 1; break; }
                min = value; worstIndex = i;
            }
        }
        if (!early) worstValue64 = min;
        const int worstValue = worstValue64 < INT_MIN ? INT_MIN : (worstValue64 > INT_MAX ? INT_MAX : (int)worstValue64);
        const int worst = lists_[worstIndex];
        sum -= lengths[worst];
        if (worstValue > 0 || lengths[worst] < SMALL_GENOME_LIST) return hitsCount;
        hitsCount--; lengths[worst] = 0; keys[worst] = -1;
    }
    return hitsCount;
}

/* ------------------------------------------------------------------------------------ find */
static int count_hits(const orc_index *ix, int *keys, int n, int maxLen) {
    int numHits = 0;
    for (int i = 0; i < n; i++) {
        const int key = keys[i];
        if (key >= 0) { const int len = key_count(ix, key); if (len > 0 && len < maxLen) numHits++; else keys[i] = -1; }
    }
    return numHits;
}
static int shrink2(int *offsets, int *keys, int *keyScores, int n) {
    int j = 0;
    for (int i = 0; i < n; i++) if (keys[i] >= 0) { offsets[j] = offsets[i]; keys[j] = keys[i]; keyScores[j] = keyScores[i]; j++; }
    return j;
}
static int get_hits(const orc_index *ix, const int *keys, int n, int block, int *starts, int *stops) {
    int numHits = 0;
    for (int i = 0; i < n; i++) {
        const int key = keys[i];
        starts[i] = -1; stops[i] = -1;
        if (key >= 0) {
            const int len = key_count(ix, key);
            if (len > 0) {
                const int len2 = block_len(ix, block, key);
                if (len2 > 0) { starts[i] = ix->starts[block][key]; stops[i] = starts[i] + len2; numHits++; }
            }
        }
    }
    return numHits;
}

int orc_index_find(const orc_index *ix, const uint8_t *basesP, const uint8_t *basesM, int blen,
                   const int8_t *baseScoresP, const int32_t *keyScoresIn, const int32_t *offsetsIn, int nkeysIn,
                   orc_site *out, int cap, int64_t *stats) {
    const orc_index_params *p = &ix->p;
    if (nkeysIn > KEYBUF || nkeysIn < 1) return 0;
    walker w; memset(&w, 0, sizeof w);
    w.ix = ix; w.p = p; w.stats = stats;
    w.c.ix = ix; w.c.shift = 31 - p->chromBits; w.c.siteMask = (int)(0xFFFFFFFFu >> (p->chromBits + 1));
    w.c.cpb = 1 << p->chromBits; w.c.lowMask = w.c.cpb - 1; w.c.highMask = ~w.c.lowMask;
    w.keylen = p->k; w.baseKeyHitScore = BASE_HIT_SCORE * p->k;
    w.indelPenalty = INDEL_PENALTY_OF(w.baseKeyHitScore); w.indelPenaltyMult = INDEL_PENALTY_MULT;
    w.maxPenalty = w.baseKeyHitScore - (1 + w.baseKeyHitScore / 8);
    w.scoreZ1Key = Z_SCORE_MULT * p->k;

    int keysOriginal[KEYBUF], keysP[KEYBUF], offsetsP[KEYBUF], keyScoresP[KEYBUF];
    int n = nkeysIn;
    for (int i = 0; i < n; i++) {                                      /* KeyRing.makeKeys */
        int key = 0;
        for (int q = offsetsIn[i]; q < offsetsIn[i] + p->k; q++) { const int x = base_num(basesP[q]); if (x < 0) { key = -1; break; } key = (key << 2) | x; }
        keysOriginal[i] = keysP[i] = key; offsetsP[i] = offsetsIn[i]; keyScoresP[i] = keyScoresIn[i];
    }
    const int maxLen = p->maxUsableLength;
    int numHits = count_hits(ix, keysP, n, maxLen);
    if (numHits > 0) {
        const int trigger = (3 * n) / 4;
        if (numHits < RELAX1 && numHits < trigger) { memcpy(keysP, keysOriginal, sizeof(int) * (size_t)n); numHits = count_hits(ix, keysP, n, (maxLen * 3) / 2); }
        if (numHits < RELAX2 && numHits < trigger) { memcpy(keysP, keysOriginal, sizeof(int) * (size_t)n); numHits = count_hits(ix, keysP, n, maxLen * 2); }
        if (numHits < RELAX3 && numHits < trigger) { memcpy(keysP, keysOriginal, sizeof(int) * (size_t)n); numHits = count_hits(ix, keysP, n, maxLen * 3); }
        if (numHits < RELAX4 && numHits < trigger) { memcpy(keysP, keysOriginal, sizeof(int) * (size_t)n); numHits = count_hits(ix, keysP, n, maxLen * 5); }
    }
    const int nOriginal = n;
    if (numHits < n) n = shrink2(offsetsP, keysP, keyScoresP, n);
    if (p->trimByGreedy) {
        const int maxLists = imax((int)(HIT_FRACTION_TO_RETAIN * n), MIN_HIT_LISTS_TO_RETAIN);
        numHits = trim_by_greedy(ix, offsetsP, keyScoresP, n, maxLists, keysP, w.baseKeyHitScore);
    }
    if (numHits < p->minApproxHitsToKeep) return 0;
    if (numHits < n) n = shrink2(offsetsP, keysP, keyScoresP, n);

    int offsetsM[KEYBUF], keysM[KEYBUF], keyScoresM[KEYBUF];
    for (int i = 0; i < n; i++) {
        offsetsM[i] = blen - (offsetsP[n - 1 - i] + p->k);                 /* KeyRing.reverseOffsets */
        keysM[i] = rc_key(keysP[n - 1 - i], p->k);                       /* KeyRing.reverseComplementKeys */
        keyScoresM[i] = keyScoresP[n - 1 - i];
    }
    int8_t *baseScoresM = (int8_t *)malloc((size_t)blen);
    for (int i = 0; i < blen; i++) baseScoresM[i] = baseScoresP[blen - 1 - i];
    const int maxQuickScore = max_quick_score(&w, offsetsP, keyScoresP, n);

    int bestScores[6] = {0, 0, 0, 0, 0, 0};
    const int prescan_qscore = (p->prescanQscore && numHits >= 5);
    int hitsCutoff = 0;
    int qscoreCutoff = (int)(MIN_QSCORE_MULT * maxQuickScore);
    int allBasesCovered = 1;
    if (offsetsP[0] != 0) allBasesCovered = 0;
    else if (offsetsP[n - 1] != (blen - p->k)) allBasesCovered = 0;
    else for (int i = 1; i < n; i++) if (offsetsP[i] > offsetsP[i - 1] + p->k) { allBasesCovered = 0; break; }
    const int pretend = (allBasesCovered || n >= nOriginal - 4 ||
                         (n >= 9 && (offsetsP[n - 1] - offsetsP[0] + p->k) > imax(40, (int)(blen * .75f))));

    const int cpb = w.c.cpb;
    int ncycles = 0;
    for (int chrom = p->minChrom; chrom <= p->maxChrom; chrom = ((chrom & w.c.highMask) + cpb)) ncycles += 2;
    int *precounts = NULL, *prescores = NULL;
    site_list ssl = {out, 0, cap, 0};
    int result = 0;
    if (prescan_qscore) {                                                /* prescanAllBlocks */
        precounts = (int *)malloc(sizeof(int) * (size_t)ncycles);
        prescores = (int *)malloc(sizeof(int) * (size_t)ncycles);
        for (int i = 0; i < ncycles; i++) { precounts[i] = n; prescores[i] = maxQuickScore; }
        int bestqscore = 0, maxHits = 0, minHitsToScore = p->minApproxHitsToKeep, cycle = 0, earlyOut = 0;
        for (int chrom = p->minChrom; chrom <= p->maxChrom && !earlyOut; chrom = ((chrom & w.c.highMask) + cpb)) {
            const int baseChrom = base_chrom(&w.c, chrom);
            const int block = baseChrom >> p->chromBits;
            for (int pmi = 0; pmi < 2 && !earlyOut; pmi++, cycle++) {
                const int *keys = pmi ? keysM : keysP, *kscores = pmi ? keyScoresM : keyScoresP, *offs = pmi ? offsetsM : offsetsP;
                int starts[KEYBUF], stops[KEYBUF];
                const int nh = get_hits(ix, keys, n, block, starts, stops);
                if (nh < minHitsToScore) { prescores[cycle] = -9999; precounts[cycle] = 0; }
                else {
                    lists L;
                    lists_init(&w, &L, block, starts, stops, offs, kscores, n, baseChrom);
                    int tq, th;
                    find_max_qscore2(&w, &L, baseChrom, minHitsToScore, 1, bestqscore >= maxQuickScore && pretend, &tq, &th);
                    prescores[cycle] = tq; precounts[cycle] = th;
                    bestqscore = imax(tq, bestqscore); maxHits = imax(maxHits, th);
                    if (bestqscore >= maxQuickScore && pretend) { minHitsToScore = imax(minHitsToScore, maxHits); earlyOut = 1; }
                }
            }
        }
        bestScores[1] = imax(bestScores[1], maxHits);
        bestScores[3] = imax(bestScores[3], bestqscore);
        if (bestScores[1] < p->minApproxHitsToKeep) goto done;
        if ((float)bestScores[3] < maxQuickScore * MIN_QSCORE_MULT2) goto done;
        if (bestScores[3] >= maxQuickScore && pretend) {
            hitsCutoff = calc_approx_hits_cutoff(p, n, bestScores[1], p->minApproxHitsToKeep, 1);
            qscoreCutoff = imax(qscoreCutoff, (int)(bestScores[3] * DYNAMIC_QSCORE_THRESH_PERFECT));
        } else {
            hitsCutoff = calc_approx_hits_cutoff(p, n, bestScores[1], p->minApproxHitsToKeep, 0);
            qscoreCutoff = imax(qscoreCutoff, (int)(bestScores[3] * PRESCAN_QSCORE_THRESH));
        }
    }
    {
        int sumBS = 0;
        for (int i = 0; i < blen; i++) sumBS += baseScoresP[i];
        const int maxScore = orc_max_quality(blen) + sumBS;                     /* msa.maxQuality(baseScores) */
        int fullyDefined = 1;
        for (int i = 0; i < blen; i++) if (base_num(basesP[i]) < 0 || basesP[i] >= 128) { fullyDefined = 0; break; }
        int cycle = 0;
        for (int chrom = p->minChrom; chrom <= p->maxChrom; chrom = ((chrom & w.c.highMask) + cpb)) {
            const int block = base_chrom(&w.c, chrom) >> p->chromBits;
            for (int strand = 0; strand < 2; strand++, cycle++) {
                if (!precounts || precounts[cycle] >= hitsCutoff || prescores[cycle] >= qscoreCutoff) {
                    const int *keys = strand ? keysM : keysP, *kscores = strand ? keyScoresM : keyScoresP, *offs = strand ? offsetsM : offsetsP;
                    int starts[KEYBUF], stops[KEYBUF];
                    const int nh = get_hits(ix, keys, n, block, starts, stops);
                    if (nh >= p->minApproxHitsToKeep)
                        slow_walk3(&w, block, starts, stops, strand ? basesM : basesP, blen, strand ? baseScoresM : baseScoresP,
                                   kscores, offs, n, chrom, strand, &ssl, bestScores, allBasesCovered, maxScore, fullyDefined);
                }
                if (p->quitAfterTwoPerfects && bestScores[5] >= 2) goto walked;
            }
        }
    }
walked:
    result = ssl.overflow ? -1 : ssl.n;
done:
    free(baseScoresM); free(precounts); free(prescores);
    return result;
}
