// BBMerge's overlap natives, host code (see bbmerge_overlap.h).  One scan template serves the four "slide read b over read a and
// weigh agreeing / disagreeing bases" loops of jni/BBMergeOverlapper.c; the weights are the only thing that differs between the
// plain and the quality-aware variants.  Arithmetic is IEEE single precision in the reference's order (no contraction).
#include "bbmerge_overlap.h"

#include <algorithm>

namespace {

// QualityTools.PROB_CORRECT as the C file embeds it (jni/BBMergeOverlapper.c:45-49)
const float kProbCorrect[71] = {
    0.000f, 0.251f, 0.369f, 0.499f, 0.602f, 0.684f, 0.749f, 0.800f, 0.842f, 0.874f, 0.900f, 0.921f, 0.937f, 0.950f, 0.960f, 0.968f,
    0.975f, 0.980f, 0.984f, 0.987f, 0.990f, 0.992f, 0.994f, 0.995f, 0.996f, 0.997f, 0.997f, 0.998f, 0.998f, 0.999f, 0.999f, 0.999f,
    0.999f, 0.999f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

inline int mid3(int x, int y, int z) { return x < y ? (x < z ? std::min(y, z) : x) : (y < z ? std::min(x, z) : y); }

struct Overlap { int istart, jstart, len; };
// the overlap of the two reads for a given insert size (:139-141, :196-198, :274-277, :349-351)
inline Overlap overlap_for(int insert, int alen, int blen) {
    Overlap o;
    o.istart = insert <= blen ? 0 : insert - blen;
    o.jstart = insert >= blen ? 0 : blen - insert;
    o.len = std::min(alen - o.istart, std::min(blen - o.jstart, insert));
    return o;
}

struct PlainWeights {          // mateByOverlapRatio / findBestRatio: fixed increments, N never counts as agreement
    float gIncr, bIncr;
    inline void add(int8_t ca, int8_t cb, int, int, float &good, float &bad) const {
        if (ca == cb) { if (ca != 'N') good += gIncr; } else bad += bIncr;
    }
};
struct ProbWeights {           // the _WithQualities variants: each base pair weighs aprob[i] * bprob[j]
    const float *aprob, *bprob;
    inline void add(int8_t ca, int8_t cb, int i, int j, float &good, float &bad) const {
        const float x = aprob[i] * bprob[j];
        if (ca == cb) good += x; else bad += x;
    }
};

template <class W> inline void scan(const int8_t *a, const int8_t *b, const Overlap &o, float badlimit, const W &w, float &good, float &bad) {
    good = 0; bad = 0;
    const int imax = o.istart + o.len;
    for (int i = o.istart, j = o.jstart; i < imax && bad <= badlimit; i++, j++) w.add(a[i], b[j], i, j, good, bad);
}

// findBestRatio / findBestRatio_WithQualities
template <class W> float find_best_ratio(const int8_t *a, int alen, const int8_t *b, int blen, int minOverlap0, int minOverlap, int minInsert,
                                         float maxRatio, float offset, const W &w) {
    float bestRatio = maxRatio + 0.0001f;
    const float halfmax = maxRatio * 0.5f;
    for (int insert = alen + blen - minOverlap; insert >= minInsert; insert--) {
        const Overlap o = overlap_for(insert, alen, blen);
        const float badlimit = bestRatio * o.len;
        float good, bad;
        scan(a, b, o, badlimit, w, good, bad);
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) return 100.0f;
            const float ratio = (bad + offset) / o.len;
            if (ratio < bestRatio) {
                bestRatio = ratio;
                if (good >= minOverlap && ratio < halfmax) return bestRatio;
            }
        }
    }
    return bestRatio;
}

// the second pass of mateByOverlapRatio / mateByOverlapRatio_WithQualities
template <class W> int best_insert(const int8_t *a, int alen, const int8_t *b, int blen, int32_t *rvector, int minOverlap0, int minOverlap,
                                   int minInsert0, float maxRatio, float margin, float offset, const W &w) {
    const int minLength = std::min(alen, blen);
    const float altBadlimit = std::max(maxRatio, 0.07f) * 2.0f * alen + 1;
    const float margin2 = (margin + offset) / minLength;
    int bestInsert = -1;
    float bestBad = (float)minLength, bestRatio = 1;
    bool ambig = false;
    for (int insert = alen + blen - minOverlap0; insert >= minInsert0; insert--) {
        const Overlap o = overlap_for(insert, alen, blen);
        const float badlimit = std::min(altBadlimit, std::min(bestRatio, maxRatio) * margin * o.len);
        float good, bad;
        scan(a, b, o, badlimit, w, good, bad);
        if (bad <= badlimit) {
            if (bad == 0 && good > minOverlap0 && good < minOverlap) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            const float ratio = (bad + offset) / o.len;
            if (ratio < bestRatio * margin) {
                ambig = (ratio * margin >= bestRatio || good < minOverlap);
                if (ratio < bestRatio) { bestInsert = insert; bestBad = bad; bestRatio = ratio; }
                if (ambig && bestRatio < margin2) { rvector[2] = (int32_t)bestBad; rvector[4] = 1; return -1; }
            }
        }
    }
    if (!ambig && bestRatio > maxRatio) bestInsert = -1;
    rvector[2] = (int32_t)bestBad;
    rvector[4] = ambig ? 1 : 0;
    return bestInsert < 0 ? -1 : bestInsert;
}

}  // namespace

extern "C" int32_t bbmerge_mate_by_overlap(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, const int8_t *aqual, const int8_t *bqual,
                                           float *aprob, float *bprob, int32_t *rvector, int32_t minOverlap0, int32_t minOverlap, int32_t minInsert0,
                                           int32_t margin, int32_t maxMismatches0, int32_t maxMismatches, int32_t minq) {
    minOverlap0 = std::min(std::max(1, minOverlap0), minOverlap);
    margin = std::max(margin, 0);
    int bestOverlap = -1, bestGood = -1, bestBad = maxMismatches0;
    bool ambig = false;
    const int maxOverlap = alen + blen - std::max(minOverlap, minInsert0);
    if (aqual && bqual) {
        for (int i = 0; i < alen; i++) aprob[i] = kProbCorrect[aqual[i]];
        for (int i = 0; i < blen; i++) bprob[i] = kProbCorrect[bqual[i]];
    } else {
        for (int i = 0; i < alen; i++) aprob[i] = 0.98f;
        for (int i = 0; i < blen; i++) bprob[i] = 0.98f;
    }
    const float minprob = kProbCorrect[mid3(1, minq, 41)];
    for (int overlap = std::max(minOverlap0, 0); overlap < maxOverlap; overlap++) {
        int good = 0, bad = 0;
        const int istart = overlap <= alen ? 0 : overlap - alen;
        const int jstart = overlap <= alen ? alen - overlap : 0;
        const int iters = std::min(overlap - istart, std::min(blen - istart, alen - jstart));
        const int imax = istart + iters, badlim = bestBad + margin;
        for (int i = istart, j = jstart; i < imax && bad <= badlim; i++, j++) {
            const float pc = aprob[j] * bprob[j];                     // the reference indexes BOTH probability arrays with j (:70)
            if (pc <= minprob) { } else if (a[j] == b[i]) good++; else bad++;
        }
        if (bad * 2 < good) {
            if (good > minOverlap) {
                if (bad <= bestBad) {
                    if (bad < bestBad || (bad == bestBad && good > bestGood)) {
                        if (bestBad - bad < margin) ambig = true;
                        bestOverlap = overlap; bestBad = bad; bestGood = good;
                    } else if (bad == bestBad) ambig = true;
                    if (ambig && bestBad < margin) { rvector[2] = bestBad; rvector[4] = 1; return -1; }
                }
            } else if (bad < margin) { rvector[2] = bestBad; rvector[4] = 1; return -1; }
        }
    }
    if (!ambig && bestBad > maxMismatches - margin) bestOverlap = -1;
    rvector[2] = bestBad;
    rvector[4] = ambig ? 1 : 0;
    return bestOverlap < 0 ? -1 : alen + blen - bestOverlap;
}

extern "C" int32_t bbmerge_mate_by_overlap_ratio(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, int32_t *rvector, int32_t minOverlap0,
                                                 int32_t minOverlap, int32_t minInsert0, int32_t minInsert, float maxRatio, float margin, float offset,
                                                 float gIncr, float bIncr) {
    minOverlap = std::max(4, std::max(minOverlap0, minOverlap));
    minOverlap0 = mid3(4, minOverlap0, minOverlap);
    const PlainWeights w{gIncr, bIncr};
    const float x = find_best_ratio(a, alen, b, blen, minOverlap0, minOverlap, minInsert, maxRatio, offset, w);
    if (x >= maxRatio) { rvector[2] = std::min(alen, blen); rvector[4] = 0; return -1; }
    maxRatio = std::min(maxRatio, x);
    return best_insert(a, alen, b, blen, rvector, minOverlap0, minOverlap, minInsert0, maxRatio, margin, offset, w);
}

extern "C" int32_t bbmerge_mate_by_overlap_ratio_with_qualities(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, const int8_t *aqual,
                                                                const int8_t *bqual, float *aprob, float *bprob, int32_t *rvector, int32_t minOverlap0,
                                                                int32_t minOverlap, int32_t minInsert0, int32_t minInsert, float maxRatio, float margin,
                                                                float offset) {
    minOverlap = std::max(4, std::max(minOverlap0, minOverlap));
    minOverlap0 = mid3(4, minOverlap0, minOverlap);
    for (int i = 0; i < alen; i++) aprob[i] = kProbCorrect[aqual[i]];
    for (int i = 0; i < blen; i++) bprob[i] = kProbCorrect[bqual[i]];
    const ProbWeights w{aprob, bprob};
    const float x = find_best_ratio(a, alen, b, blen, minOverlap0, minOverlap, minInsert, maxRatio, offset, w);
    if (x > maxRatio) { rvector[2] = std::min(alen, blen); rvector[4] = 0; return -1; }
    maxRatio = std::min(maxRatio, x);
    return best_insert(a, alen, b, blen, rvector, minOverlap0, minOverlap, minInsert0, maxRatio, margin, offset, w);
}
