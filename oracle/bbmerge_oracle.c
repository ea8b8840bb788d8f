/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Function-by-function CPU restatement of BBMerge's overlap natives
 * (jni/BBMergeOverlapper.c: mateByOverlap :24-125, findBestRatio :127-179, findBestRatio_WithQualities :182-228,
 * mateByOverlapRatio_WithQualities :230-319, mateByOverlapRatio :321-402), the checker of jni/bbmerge_overlap.cpp.
 * PARITY STATUS: restatement only ("parity unpinned"): the reference file includes <jni.h>, which this image lacks, and the
 * reference holds no vectors for these functions.
 */
#include <stdint.h>
#include <stddef.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static float fmin2(float a, float b) { return a < b ? a : b; }
static float fmax2(float a, float b) { return a > b ? a : b; }
static int mid(int x, int y, int z) { return x < y ? (x < z ? imin(y, z) : x) : (y < z ? imin(x, z) : y); }

static const float probCorrect[71] = {
    0.000f, 0.251f, 0.369f, 0.499f, 0.602f, 0.684f, 0.749f, 0.800f, 0.842f, 0.874f, 0.900f, 0.921f, 0.937f, 0.950f, 0.960f, 0.968f,
    0.975f, 0.980f, 0.984f, 0.987f, 0.990f, 0.992f, 0.994f, 0.995f, 0.996f, 0.997f, 0.997f, 0.998f, 0.998f, 0.999f, 0.999f, 0.999f,
    0.999f, 0.999f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

int32_t orc_bbmerge_mate_by_overlap(const int8_t *abases, int alen, const int8_t *bbases, int blen, const int8_t *aqual, const int8_t *bqual,
                                    float *aprob, float *bprob, int32_t *rvector, int minOverlap0, int minOverlap, int minInsert0, int margin,
                                    int maxMismatches0, int maxMismatches, int minq) {
    minOverlap0 = imin(imax(1, minOverlap0), minOverlap);
    margin = imax(margin, 0);
    int bestOverlap = -1, bestGood = -1, bestBad = maxMismatches0, ambig = 0;
    const int maxOverlap = alen + blen - imax(minOverlap, minInsert0);
    if (aqual != NULL && bqual != NULL) {
        for (int i = 0; i < alen; i++) aprob[i] = probCorrect[aqual[i]];
        for (int i = 0; i < blen; i++) bprob[i] = probCorrect[bqual[i]];
    } else {
        for (int i = 0; i < alen; i++) aprob[i] = 0.98f;
        for (int i = 0; i < blen; i++) bprob[i] = 0.98f;
    }
    const float minprob = probCorrect[mid(1, minq, 41)];
    for (int overlap = imax(minOverlap0, 0); overlap < maxOverlap; overlap++) {
        int good = 0, bad = 0;
        int istart = (overlap <= alen ? 0 : overlap - alen);
        int jstart = (overlap <= alen ? alen - overlap : 0);
        {
            const int iters = imin(overlap - istart, imin(blen - istart, alen - jstart));
            const int imx = istart + iters;
            const int badlim = bestBad + margin;
            for (int i = istart, j = jstart; i < imx && bad <= badlim; i++, j++) {
                const int8_t ca1 = abases[j], cb1 = bbases[i];
                const float pc = aprob[j] * bprob[j];
                if (pc <= minprob) { /* nothing */ } else if (ca1 == cb1) { good++; } else { bad++; }
            }
        }
        if (bad * 2 < good) {
            if (good > minOverlap) {
                if (bad <= bestBad) {
                    if (bad < bestBad || (bad == bestBad && good > bestGood)) {
                        if (bestBad - bad < margin) ambig = 1;
                        bestOverlap = overlap; bestBad = bad; bestGood = good;
                    } else if (bad == bestBad) {
                        ambig = 1;
                    }
                    if (ambig && bestBad < margin) { rvector[2] = bestBad; rvector[4] = (ambig ? 1 : 0); return -1; }
                }
            } else if (bad < margin) {
                ambig = 1;
                rvector[2] = bestBad; rvector[4] = (ambig ? 1 : 0);
                return -1;
            }
        }
    }
    if (!ambig && bestBad > maxMismatches - margin) bestOverlap = -1;
    rvector[2] = bestBad;
    rvector[4] = (ambig ? 1 : 0);
    return (bestOverlap < 0 ? -1 : alen + blen - bestOverlap);
}

static float find_best_ratio(const int8_t *abases, int alen, const int8_t *bbases, int blen, int minOverlap0, int minOverlap, int minInsert,
                             float maxRatio, float offset, float gIncr, float bIncr) {
    float bestRatio = maxRatio + 0.0001f;
    const float halfmax = maxRatio * 0.5f;
    const int8_t N = 'N';
    const int largestInsertToTest = (alen + blen - minOverlap), smallestInsertToTest = minInsert;
    for (int insert = largestInsertToTest; insert >= smallestInsertToTest; insert--) {
        const int istart = (insert <= blen ? 0 : insert - blen);
        const int jstart = (insert >= blen ? 0 : blen - insert);
        const int overlapLength = imin(alen - istart, imin(blen - jstart, insert));
        const float badlimit = bestRatio * overlapLength;
        float good = 0, bad = 0;
        const int imx = istart + overlapLength;
        for (int i = istart, j = jstart; i < imx && bad <= badlimit; i++, j++) {
            const int8_t ca = abases[i], cb = bbases[j];
            if (ca == cb) { if (ca != N) good += gIncr; } else { bad += bIncr; }
        }
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) return 100.0f;
            float ratio = (bad + offset) / overlapLength;
            if (ratio < bestRatio) { bestRatio = ratio; if (good >= minOverlap && ratio < halfmax) return bestRatio; }
        }
    }
    return bestRatio;
}

static float find_best_ratio_q(const int8_t *abases, int alen, const int8_t *bbases, int blen, const float *aprob, const float *bprob,
                               int minOverlap0, int minOverlap, int minInsert, float maxRatio, float offset) {
    float bestRatio = maxRatio + 0.0001f;
    const float halfmax = maxRatio * 0.5f;
    const int largestInsertToTest = (alen + blen - minOverlap), smallestInsertToTest = minInsert;
    for (int insert = largestInsertToTest; insert >= smallestInsertToTest; insert--) {
        const int istart = (insert <= blen ? 0 : insert - blen);
        const int jstart = (insert >= blen ? 0 : blen - insert);
        const int overlapLength = imin(alen - istart, imin(blen - jstart, insert));
        const float badlimit = bestRatio * overlapLength;
        float good = 0, bad = 0;
        const int imx = istart + overlapLength;
        for (int i = istart, j = jstart; i < imx && bad <= badlimit; i++, j++) {
            const int8_t ca = abases[i], cb = bbases[j];
            const float x = aprob[i] * bprob[j];
            if (ca == cb) good += x; else bad += x;
        }
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) return 100.0f;
            float ratio = (bad + offset) / overlapLength;
            if (ratio < bestRatio) { bestRatio = ratio; if (good >= minOverlap && ratio < halfmax) return bestRatio; }
        }
    }
    return bestRatio;
}

int32_t orc_bbmerge_mate_by_overlap_ratio_q(const int8_t *abases, int alen, const int8_t *bbases, int blen, const int8_t *aqual, const int8_t *bqual,
                                            float *aprob, float *bprob, int32_t *rvector, int minOverlap0, int minOverlap, int minInsert0,
                                            int minInsert, float maxRatio, float margin, float offset) {
    minOverlap = imax(4, imax(minOverlap0, minOverlap));
    minOverlap0 = mid(4, minOverlap0, minOverlap);
    const int minLength = imin(alen, blen);
    for (int i = 0; i < alen; i++) aprob[i] = probCorrect[aqual[i]];
    for (int i = 0; i < blen; i++) bprob[i] = probCorrect[bqual[i]];
    {
        float x = find_best_ratio_q(abases, alen, bbases, blen, aprob, bprob, minOverlap0, minOverlap, minInsert, maxRatio, offset);
        if (x > maxRatio) { rvector[2] = minLength; rvector[4] = 0; return -1; }
        maxRatio = fmin2(maxRatio, x);
    }
    const float altBadlimit = fmax2(maxRatio, 0.07f) * 2.0f * alen + 1;
    const float margin2 = (margin + offset) / minLength;
    int bestInsert = -1, ambig = 0;
    float bestBad = minLength, bestRatio = 1;
    const int largestInsertToTest = (alen + blen - minOverlap0), smallestInsertToTest = minInsert0;
    for (int insert = largestInsertToTest; insert >= smallestInsertToTest; insert--) {
        float good = 0, bad = 0;
        const int istart = (insert <= blen ? 0 : insert - blen);
        const int jstart = (insert >= blen ? 0 : blen - insert);
        const int overlapLength = imin(alen - istart, imin(blen - jstart, insert));
        const float badlimit = fmin2(altBadlimit, fmin2(bestRatio, maxRatio) * margin * overlapLength);
        const int imx = istart + overlapLength;
        for (int i = istart, j = jstart; i < imx && bad <= badlimit; i++, j++) {
            const int8_t ca = abases[i], cb = bbases[j];
            const float x = aprob[i] * bprob[j];
            if (ca == cb) good += x; else bad += x;
        }
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            float ratio = (bad + offset) / overlapLength;
            if (ratio < bestRatio * margin) {
                ambig = (ratio * margin >= bestRatio || good < minOverlap);
                if (ratio < bestRatio) { bestInsert = insert; bestBad = bad; bestRatio = ratio; }
                if (ambig && bestRatio < margin2) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            }
        }
    }
    if (!ambig && bestRatio > maxRatio) bestInsert = -1;
    rvector[2] = (int32_t)bestBad;
    rvector[4] = (ambig ? 1 : 0);
    return (bestInsert < 0 ? -1 : bestInsert);
}

int32_t orc_bbmerge_mate_by_overlap_ratio(const int8_t *abases, int alen, const int8_t *bbases, int blen, int32_t *rvector, int minOverlap0,
                                          int minOverlap, int minInsert0, int minInsert, float maxRatio, float margin, float offset, float gIncr,
                                          float bIncr) {
    minOverlap = imax(4, imax(minOverlap0, minOverlap));
    minOverlap0 = mid(4, minOverlap0, minOverlap);
    const int minLength = imin(alen, blen);
    {
        float x = find_best_ratio(abases, alen, bbases, blen, minOverlap0, minOverlap, minInsert, maxRatio, offset, gIncr, bIncr);
        if (x >= maxRatio) { rvector[2] = minLength; rvector[4] = 0; return -1; }
        maxRatio = fmin2(maxRatio, x);
    }
    const float altBadlimit = fmax2(maxRatio, 0.07f) * 2.0f * alen + 1;
    const float margin2 = (margin + offset) / minLength;
    const int8_t N = 'N';
    int bestInsert = -1, ambig = 0;
    float bestBad = minLength, bestRatio = 1;
    const int largestInsertToTest = (alen + blen - minOverlap0), smallestInsertToTest = minInsert0;
    for (int insert = largestInsertToTest; insert >= smallestInsertToTest; insert--) {
        const int istart = (insert <= blen ? 0 : insert - blen);
        const int jstart = (insert >= blen ? 0 : blen - insert);
        const int overlapLength = imin(alen - istart, imin(blen - jstart, insert));
        const float badlimit = fmin2(altBadlimit, fmin2(bestRatio, maxRatio) * margin * overlapLength);
        float good = 0, bad = 0;
        const int imx = istart + overlapLength;
        for (int i = istart, j = jstart; i < imx && bad <= badlimit; i++, j++) {
            const int8_t ca = abases[i], cb = bbases[j];
            if (ca == cb) { if (ca != N) good += gIncr; } else { bad += bIncr; }
        }
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            float ratio = (bad + offset) / overlapLength;
            if (ratio < bestRatio * margin) {
                ambig = (ratio * margin >= bestRatio || good < minOverlap);
                if (ratio < bestRatio) { bestInsert = insert; bestBad = bad; bestRatio = ratio; }
                if (ambig && bestRatio < margin2) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            }
        }
    }
    if (!ambig && bestRatio > maxRatio) bestInsert = -1;
    rvector[2] = (int32_t)bestBad;
    rvector[4] = (ambig ? 1 : 0);
    return (bestInsert < 0 ? -1 : bestInsert);
}
