/*
 * ORACLE (test infrastructure only): CPU restatement of the paired-read rescue scan
 *   AbstractMapThread.quickRescue   current/align2/AbstractMapThread.java:2300-2391
 *   + SiteScore.setPerfect          current/stream/SiteScore.java:239-292 (as the probe oracle states it)
 * Parity: pinned by restatement only (Java-only code, no runnable reference here, no fixtures in the reference).
 */
#include <limits.h>
#include <stdint.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabsdif(int a, int b) { return a > b ? a - b : b - a; }

static void set_perfect(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int start, int stop,
                        int *perfectOut, int *semiOut) {
    *perfectOut = 0; *semiOut = 0;
    if (blen != stop - start + 1) return;
    int perfect = 1, semiperfect = 1;
    int refloc = start, readloc = 0, N = 0;
    const int mx = imin(stop, reflen - 1), nlimit = blen / 2;
    if (start < 0) { N -= start; readloc -= start; refloc -= start; perfect = 0; }
    if (stop >= reflen) { N += (stop - reflen + 1); perfect = 0; }
    if (N > nlimit) return;
    for (; refloc <= mx; refloc++, readloc++) {
        const int c = bases[readloc], r = ref[refloc];
        if (c != r || c == 'N') {
            perfect = 0;
            if (c == 'N') semiperfect = 0;
            if (r != 'N' || (N = N + 1) > nlimit) return;
        }
    }
    semiperfect = semiperfect && (N <= nlimit);
    perfect = perfect && semiperfect && (N == 0);
    *perfectOut = perfect; *semiOut = semiperfect;
}

/* SiteScore.setPerfect for callers outside this file (the pipeline emulations): out2 = {perfect, semiperfect} */
void orc_set_perfect(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int start, int stop, int32_t *out2) {
    int p = 0, sp = 0;
    set_perfect(bases, blen, ref, reflen, start, stop, &p, &sp);
    out2[0] = p; out2[1] = sp;
}

/* out8 = {found, start, stop, score, mismatches (ss.slowScore), perfect, semiperfect, maxContigMatches} */
void orc_quick_rescue(const uint8_t *bases, int blen, const uint8_t *ref, int reflen, int minIndex,
                      int loc, int searchDist, int searchRight, int idealStart, int maxAllowedMismatches,
                      int pointsMatch, int pointsMatch2, int useAffine, int baseHitScore, int32_t *out8) {
    for (int i = 0; i < 8; i++) out8[i] = 0;
    if (!bases || blen < 10) return;
    int lowerBound, upperBound;
    if (searchRight) { lowerBound = imax(minIndex, loc); upperBound = imin(reflen - blen, loc + searchDist); }
    else { lowerBound = imax(minIndex, loc - searchDist); upperBound = imin(reflen - blen, loc); }
    int minMismatches = maxAllowedMismatches + 1;
    int maxContigMatches = 0, bestScore = 0, bestStart = -1, bestAbsdif = INT_MAX;
    for (int start = searchRight ? lowerBound : upperBound; searchRight ? start <= upperBound : start >= lowerBound;
         start += searchRight ? 1 : -1) {
        int mismatches = 0, contig = 0, currentContig = 0;
        for (int j = 0; j < blen && mismatches <= minMismatches; j++) {
            const uint8_t c = bases[j], r = ref[start + j];
            if (c != r || c == 'N') { mismatches++; contig = imax(contig, currentContig); currentContig = 0; }
            else currentContig++;
        }
        const int score = (blen - mismatches) + contig;
        const int absdif = iabsdif(start, idealStart);
        if (mismatches <= minMismatches && (score > bestScore || (score == bestScore && absdif < bestAbsdif))) {
            bestStart = start; minMismatches = mismatches; maxContigMatches = contig; bestScore = score; bestAbsdif = absdif;
            if (mismatches == 0) {
                if (searchRight) upperBound = imin(upperBound, idealStart + absdif);
                else lowerBound = imax(lowerBound, idealStart - absdif);
            }
        }
    }
    if (bestStart < 0) return;
    const int scoreOut = useAffine ? pointsMatch + pointsMatch2 * (blen - 1 - minMismatches)
                                   : maxContigMatches + baseHitScore * (blen - minMismatches);
    out8[0] = 1; out8[1] = bestStart; out8[2] = bestStart + blen - 1; out8[3] = scoreOut; out8[4] = minMismatches;
    set_perfect(bases, blen, ref, reflen, bestStart, bestStart + blen - 1, &out8[5], &out8[6]);
    out8[7] = maxContigMatches;
}
