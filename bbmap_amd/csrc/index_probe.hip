// k-mer index probe (align2.BBIndex.findAdvanced) on gfx950.
//
// Layout in HBM: per block the CSR arrays of the reference (starts[4^k+1], sites[]; current/align2/Block.java:
// 162-165), COUNTS[4^k], the 1001-entry length histogram, and every chromosome's byte array; all uploaded once by
// bbidx_create and shared by every launch.
//
// Mapping: ONE READ PER LANE.  BBIndex.find is an order-dependent state machine per read (cutoffs, the
// bestScores[6] array and the previous-site subsumption are carried across strands and blocks),
// so the parallelism is across reads; a lane's working arrays (keys, list cursors, the per-base location array)
// live in its private (scratch) memory, which the hardware interleaves across lanes.  The reference's binary heap
// (QuadHeap) only ever exposes its minimum under the total order (site, column), so it is replaced by a scan
// over the list heads with the same tie-break.
//
// Functions follow, in this order, current/align2/BBIndex.java: calcApproxHitsCutoff :3267-3294, maxScoreZ
// :2948-2964, maxQuickScore :2473-2487, scoreZ2 :2882-2914, scoreLeft/Right :2967-3035, quickScore :2490-2511
// (+ AbstractIndex.scoreY, AbstractIndex.java:52-78), findMaxQscore2 :2294-2450, extendScore :2558-2833
// (+ MultiStateAligner11tsJNI.calcAffineScore :871-1027), makeGapArray :2837-2878, SiteScore.setPerfect
// (current/stream/SiteScore.java:239-292), slowWalk3 :1219-1706, trimExcessHitListsByGreedy :266-350
// (+ Solver.valueOfElement/findWorstGreedy, current/align2/Solver.java:46-151), find :403-639,
// prescanAllBlocks :642-741.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "bbmap_amd.h"
#include "index_common.h"

void bbmap_set_error(const char *msg);   // msa_host.hip

namespace bbidx {

// the read as one strand sees it: plus = bytes as given; minus = reverse complement, base scores reversed
struct Strand {
    const uint8_t *b; const int8_t *q; int len; bool minus;
    __device__ inline int base(int i) const { return minus ? complement_extended(b[len - 1 - i]) : b[i]; }
    __device__ inline int bscore(int i) const { return minus ? q[len - 1 - i] : q[i]; }
};

struct Walker {
    const DevIndex *ix;
    Codec c;
    int k, baseKeyHitScore, indelPenalty, indelPenaltyMult, maxPenalty, scoreZ1Key;
    // work counters of the current read: list entries consumed by the prescan / by the walk,
    // extendScore calls, reference bytes compared
    mutable unsigned cPrescan, cWalk, cExtend, cRefBytes;
};

struct Lists {
    int n, nlive;
    int row[KB], stop[KB], value[KB], offs[KB], kscore[KB];
    bool live[KB];
    const int *sites;
};

__device__ int maxScoreZ(const Walker &w, const int *offsets, int n) {
    int score = 0, a0 = -1, b0 = -1;
    for (int i = 0; i < n; i++) { const int a = offsets[i]; if (b0 < a) { score += b0 - a0; a0 = a; } b0 = a + w.k; }
    return (score + b0 - a0) * Z_MULT;
}
__device__ int maxQuickScore(const Walker &w, const int *offsets, const int *keyScores, int n) {
    int x = 0;
    for (int i = 0; i < n; i++) x += keyScores[i];
    return x + maxScoreZ(w, offsets, n) + Y_MULT * (offsets[n - 1] - offsets[0]);
}
__device__ int scoreZ2(const Walker &w, const int *locs, int centerIndex, const int *offsets, int numApproxHits, int numHits) {
    if (numApproxHits == 1) return w.scoreZ1Key;
    const int center = locs[centerIndex];
    const int maxLoc = center + w.ix->p.maxIndel2, minLoc = max(0, center - w.ix->p.maxIndel);
    int score = 0, a0 = -1, b0 = -1;
    for (int i = 0; i < numHits; i++) {
        const int loc = locs[i];
        if (loc >= minLoc && loc <= maxLoc) { const int a = offsets[i]; if (b0 < a) { score += b0 - a0; a0 = a; } b0 = a + w.k; }
    }
    return (score + b0 - a0) * Z_MULT;
}
__device__ int scoreSide(const Walker &w, const int *locs, const int *keyScores, int centerIndex, int numHits, int dir) {
    int score = 0, prev, loc = locs[centerIndex];
    for (int i = centerIndex + dir; i >= 0 && i < numHits; i += dir) {
        if (locs[i] >= 0) {
            prev = loc; loc = locs[i];
            const int offset = absdif(loc, prev);
            if (offset <= w.ix->p.maxIndel) {
                score += keyScores[i];
                if (offset != 0) score -= min(w.indelPenalty + w.indelPenaltyMult * offset, w.maxPenalty);
            } else loc = prev;
        }
    }
    return score;
}
__device__ int quickScore(const Walker &w, const int *locs, const int *keyScores, int centerIndex, const int *offsets,
                          int numApproxHits, int numHits) {
    if (numApproxHits == 1) return keyScores[centerIndex];
    const int x = keyScores[centerIndex] + scoreSide(w, locs, keyScores, centerIndex, numHits, -1)
                + scoreSide(w, locs, keyScores, centerIndex, numHits, +1) - centerIndex;
    const int center = locs[centerIndex];
    int rightIndex = -1;
    for (int i = numHits - 1; rightIndex < centerIndex; i--) if (locs[i] == center) rightIndex = i;
    return x + Y_MULT * (offsets[rightIndex] - offsets[centerIndex]);
}

__device__ inline int adjustSite(const Walker &w, int a, int offset, int baseChrom) {
    if ((a & w.c.siteMask) >= offset) return a - offset;
    const int ch = w.c.chromOf(a, baseChrom), st = w.c.siteOf(a);
    return w.c.toNumber(max(st - offset, 0), ch);
}
__device__ void listsInit(const Walker &w, Lists &L, int block, const int *starts, const int *stops, const int *offsets,
                          const int *keyScores, int n, int baseChrom) {
    L.n = 0; L.sites = w.ix->sites[block];
    for (int i = 0; i < n; i++) {
        if (starts[i] < 0) continue;
        const int j = L.n++;
        L.row[j] = starts[i]; L.stop[j] = stops[i]; L.offs[j] = offsets[i]; L.kscore[j] = keyScores[i];
        L.value[j] = adjustSite(w, L.sites[starts[i]], offsets[i], baseChrom);
        L.live[j] = true;
    }
    L.nlive = L.n;
}
__device__ inline int listsPeek(const Lists &L) {      // QuadHeap.peek(): minimum under (site, column)
    int best = -1;
    for (int i = 0; i < L.n; i++) if (L.live[i] && (best < 0 || L.value[i] < L.value[best])) best = i;
    return best;
}

__device__ void findMaxQscore2(const Walker &w, Lists &L, int baseChrom, int prevMaxHits, bool perfectOnly, int &outQ, int &outHits) {
    const bbidx_params &p = w.ix->p;
    const int numHits = L.n;
    const int mqs = maxQuickScore(w, L.offs, L.kscore, numHits);
    int topQscore = -999999999, maxHits = 0, approxHitsCutoff, indelCutoff;
    if (perfectOnly) { approxHitsCutoff = numHits; indelCutoff = 0; }
    else { approxHitsCutoff = max(prevMaxHits, min(p.minApproxHitsToKeep, numHits - 1)); indelCutoff = p.maxIndel2; }
    while (L.nlive > 0) {
        const int centerIndex = listsPeek(L);
        const int site = L.value[centerIndex];
        int approxHits = 0;
        {
            const int minsite = site - min(p.maxIndel, indelCutoff), maxsite = site + p.maxIndel2;
            for (int column = 0, chances = numHits - approxHitsCutoff; column < numHits && chances >= 0; column++) {
                const int x = L.value[column];
                if (x >= minsite && x <= maxsite) approxHits++; else chances--;
            }
        }
        if (approxHits >= approxHitsCutoff) {
            const int qscore = quickScore(w, L.value, L.kscore, centerIndex, L.offs, approxHits, numHits)
                             + scoreZ2(w, L.value, centerIndex, L.offs, approxHits, numHits);
            if (qscore > topQscore) {
                maxHits = max(approxHits, maxHits);
                approxHitsCutoff = max(approxHitsCutoff, approxHits - 1);
                topQscore = qscore;
                if (qscore >= mqs) { outQ = topQscore; outHits = maxHits; return; }
            }
        }
        for (;;) {
            const int col = listsPeek(L);
            if (col < 0 || L.value[col] != site) break;
            w.cPrescan++;
            const int row = L.row[col] + 1;
            if (row < L.stop[col]) { L.row[col] = row; L.value[col] = adjustSite(w, L.sites[row], L.offs[col], baseChrom); }
            else {
                L.live[col] = false; L.nlive--;
                if (perfectOnly || L.nlive < approxHitsCutoff) { outQ = topQscore; outHits = maxHits; return; }
            }
            if (L.nlive == 0) break;
        }
    }
    outQ = topQscore; outHits = maxHits;
}

// MultiStateAligner11tsJNI.calcAffineScore(locArray, baseScores, bases[, minContig]) in plain points
__device__ int calcAffineScore(const int *locArray, int n, const Strand &rd, int minContig) {
    int contig = 0, maxContig = 0, score = 0, lastLoc = -3, lastValue = -1, timeInMode = 0;
    for (int i = 0; i < n; i++) {
        const int loc = locArray[i];
        if (loc > 0) {
            if (loc == lastValue) { contig++; score += 100 + rd.bscore(i); }
            else if (loc == lastLoc || lastLoc < 0) { maxContig = max(maxContig, contig); contig = 1; score += 70 + rd.bscore(i); }
            else if (loc < lastLoc) {
                maxContig = max(maxContig, contig); contig = 0;
                score += 70 + rd.bscore(i) + calcDelScoreApprox(lastLoc - loc + 1);
                timeInMode = 1;
            } else {
                maxContig = max(maxContig, contig); contig = 0;
                score += 70 + rd.bscore(i) + insCum(min(loc - lastLoc, 5));
                timeInMode = 1;
            }
            lastLoc = loc;
        } else if (loc == -1) {
            if (lastValue < 0 && timeInMode > 0) { timeInMode++; score += subArr(timeInMode); }
            else { score += -127; timeInMode = 1; }
        } else { timeInMode = 0; }
        lastValue = loc;
    }
    if (minContig > 1 && max(contig, maxContig) < minContig) score = min(score, -50 * n);
    return score;
}

__device__ int extendScore(const Walker &w, const Strand &rd, const int *offsets, const int *values, int chrom, int centerIndex,
                           int *locArray, int numHits) {
    const bbidx_params &p = w.ix->p;
    const int blen = rd.len;
    const int centerVal = values[centerIndex], centerLoc = w.c.siteOf(centerVal);
    const int minVal = centerVal - p.maxIndel, maxVal = centerVal + p.maxIndel2;
    const uint8_t *ref = w.ix->chromArr[chrom];
    const int reflen = w.ix->chromArrLen[chrom];
    w.cExtend++;
    for (int i = 0; i < blen; i++) locArray[i] = -1;
    for (int i = 0, keynum = 0; i < numHits; i++) {
        const int value = values[i];
        if (value >= minVal && value <= maxVal) {
            const int refbase = w.c.siteOf(value);
            keynum++;
            int misses = 0;
            for (int cloc = offsets[i] + w.k - 1, rloc = refbase + cloc; cloc >= 0 && rloc >= 0 && rloc < reflen; cloc--, rloc--) {
                const int old = locArray[cloc];
                if (old == refbase) break;
                if (misses > 0 && old >= 0) break;
                w.cRefBytes++;
                if (rd.base(cloc) == ref[rloc]) { if (old < 0 || refbase == centerLoc) locArray[cloc] = refbase; }
                else { misses++; if (old >= 0 || keynum > 1) break; }
            }
        }
    }
    for (int i = 0; i < numHits; i++) {
        const int value = values[i];
        if (value >= minVal && value <= maxVal) {
            const int refbase = w.c.siteOf(value);
            int misses = 0;
            for (int cloc = offsets[i] + w.k, rloc = refbase + cloc; cloc < blen && rloc < reflen; cloc++, rloc++) {
                const int old = locArray[cloc];
                if (old == refbase) break;
                if (misses > 0 && old >= 0) break;
                w.cRefBytes++;
                if (rd.base(cloc) == ref[rloc]) { if (old < 0 || refbase == centerLoc) locArray[cloc] = refbase; }
                else { misses++; if (old >= 0) break; }
            }
        }
    }
    for (int i = 0; i < blen; i++) if (rd.base(i) == 'N') locArray[i] = -2;
    return calcAffineScore(locArray, blen, rd, p.kfilter);
}

__device__ int makeGapArray(int *locArray, int n, int minLoc, int minGap, int *out, int cap) {
    int gaps = 0; bool doSort = false;
    if (locArray[0] < 0) locArray[0] = minLoc;
    for (int i = 1; i < n; i++) {
        if (locArray[i] < 0) locArray[i] = locArray[i - 1] + 1; else locArray[i] += i;
        if (locArray[i] < locArray[i - 1]) doSort = true;
    }
    if (doSort) {                                        // Arrays.sort: insertion sort is enough for <=600 nearly sorted ints
        for (int i = 1; i < n; i++) { const int v = locArray[i]; int j = i - 1; while (j >= 0 && locArray[j] > v) { locArray[j + 1] = locArray[j]; j--; } locArray[j + 1] = v; }
    }
    for (int i = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) gaps++;
    if (gaps < 1) return 0;
    const int len = 2 + gaps * 2;
    if (len > cap) return -1;
    out[0] = locArray[0]; out[len - 1] = locArray[n - 1];
    for (int i = 1, j = 1; i < n; i++) if (locArray[i] - locArray[i - 1] > minGap) { out[j] = locArray[i - 1]; out[j + 1] = locArray[i]; j += 2; }
    return len;
}

__device__ void setPerfect(const DevIndex &ix, bbidx_site &ss, const Strand &rd) {
    const int blen = rd.len;
    if (blen != ss.stop - ss.start + 1) { ss.perfect = 0; ss.semiperfect = 0; return; }
    const uint8_t *ref = ix.chromArr[ss.chrom];
    const int reflen = ix.chromArrLen[ss.chrom];
    bool perfect = true, semiperfect = true;
    int refloc = ss.start, readloc = 0, N = 0;
    const int mx = min(ss.stop, reflen - 1), nlimit = blen / 2;
    if (ss.start < 0) { N -= ss.start; readloc -= ss.start; refloc -= ss.start; perfect = false; }
    if (ss.stop >= reflen) { N += (ss.stop - reflen + 1); perfect = false; }
    if (N > nlimit) { ss.perfect = ss.semiperfect = 0; return; }
    for (; refloc <= mx; refloc++, readloc++) {
        const int c = rd.base(readloc), r = ref[refloc];
        if (c != r || c == 'N') {
            perfect = false;
            if (c == 'N') semiperfect = false;
            if (r != 'N' || (N = N + 1) > nlimit) { ss.perfect = perfect; ss.semiperfect = 0; return; }
        }
    }
    semiperfect = semiperfect && (N <= nlimit);
    perfect = perfect && semiperfect && (N == 0);
    ss.perfect = perfect; ss.semiperfect = semiperfect;
}
__device__ inline bool overlap(int a1, int b1, int a2, int b2) { return a2 <= b1 && b2 >= a1; }

struct SiteList { bbidx_site *v; int n, cap; bool overflow; };

__device__ void slowWalk3(const Walker &w, Lists &L, int *locArray, int block, const int *starts, const int *stops,
                          const Strand &rd, const int *keyScores, const int *offsets, int numKeys, int baseChrom_, int strand,
                          SiteList &ssl, int *bestScores, bool allBasesCovered, int maxScore, bool fullyDefined) {
    const bbidx_params &p = w.ix->p;
    const int blen = rd.len;
    const int mqs = maxQuickScore(w, offsets, keyScores, numKeys);
    const int baseChrom = w.c.baseChrom(baseChrom_);
    listsInit(w, L, block, starts, stops, offsets, keyScores, numKeys, baseChrom);
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

    int prevIdx = -1;
    bool finished = false;
    while (L.nlive > 0 && !finished) {
        const int centerIndex = listsPeek(L);
        const int site = L.value[centerIndex];
        int maxNearbySite = site, approxHits = 0;
        {
            const int minsite = site - p.maxIndel, maxsite = site + p.maxIndel2;
            for (int column = 0, chances = numHits - approxHitsCutoff; column < numHits && chances >= 0; column++) {
                const int x = L.value[column];
                if (x >= minsite && x <= maxsite) { if (x > maxNearbySite) maxNearbySite = x; approxHits++; } else chances--;
            }
        }
        if (approxHits >= approxHitsCutoff) {
            int score;
            int qscore = filter_by_qscore ? quickScore(w, L.value, L.kscore, centerIndex, L.offs, approxHits, numHits) : qcutoff;
            qscore += scoreZ2(w, L.value, centerIndex, L.offs, approxHits, numHits);
            int mapStart = site, mapStop = maxNearbySite;
            bool locArrayValid = false;
            if (qscore < qcutoff) score = -1;
            else {
                const int chrom = w.c.chromOf(site, baseChrom);
                if (shortCircuit && qscore == mqs) score = maxScore;
                else {
                    score = extendScore(w, rd, L.offs, L.value, chrom, centerIndex, locArray, numHits);
                    locArrayValid = true;
                    int mn = INT_MAX, mx = INT_MIN;
                    for (int i = 0; i < blen; i++) { const int x = locArray[i]; if (x > -1) { if (x < mn) mn = x; if (x > mx) mx = x; } }
                    if (mn < 0 || mx < 0) score = -99999;
                    mapStart = w.c.toNumber(mn, chrom);
                    mapStop = w.c.toNumber(mx, chrom);
                }
                if (score == maxScore) {
                    qcutoff = max(qcutoff, (int)(mqs * DYN_QSCORE_PERFECT));
                    approxHitsCutoff = calcApproxHitsCutoff(p, numKeys, maxHits, p.minApproxHitsToKeep, true);
                }
                if (score >= cutoff) { qcutoff = max(qcutoff, (int)(qscore * DYN_QSCORE)); bestqscore = max(qscore, bestqscore); }
            }
            if (score >= cutoff) {
                if (score > currentTopScore) {
                    maxHits = max(approxHits, maxHits);
                    approxHitsCutoff = calcApproxHitsCutoff(p, numKeys, maxHits, approxHitsCutoff, currentTopScore >= maxScore);
                    cutoff = max(cutoff, (int)(score * DYN_SCORE));
                    if (score >= maxScore) cutoff = max(cutoff, (int)(score * 0.95f));
                    currentTopScore = score;
                }
                const int chrom = w.c.chromOf(mapStart, baseChrom);
                const int site2 = w.c.siteOf(mapStart);
                const int site3 = w.c.siteOf(mapStop) + blen - 1;
                int gapArr[BBIDX_MAX_GAPS]; int ngaps = 0;
                if (site3 - site2 >= MINGAP + blen && locArrayValid) {
                    ngaps = makeGapArray(locArray, blen, site2, MINGAP, gapArr, BBIDX_MAX_GAPS);
                    if (ngaps < 0) ngaps = 0;
                    if (ngaps > 0) { gapArr[0] = min(gapArr[0], site2); gapArr[ngaps - 1] = max(gapArr[ngaps - 1], site3); }
                }
                bbidx_site ss; bool haveSS = false;
                const bool perfect1 = (score == maxScore && fullyDefined);
                const bool inbounds = (site2 >= 0 && site3 < w.ix->chromLengths[chrom]);
                bbidx_site *prevSS = prevIdx >= 0 ? &ssl.v[prevIdx] : nullptr;
                auto newSite = [&](bool withGaps) {
                    ss.chrom = chrom; ss.strand = strand; ss.start = site2; ss.stop = site3; ss.hits = approxHits; ss.score = score;
                    ss.perfect = ss.semiperfect = perfect1 ? 1 : 0;
                    ss.ngaps = 0;
                    for (int g = 0; g < BBIDX_MAX_GAPS; g++) ss.gaps[g] = 0;
                    if (!perfect1) setPerfect(*w.ix, ss, rd);
                    if (withGaps) { ss.ngaps = ngaps; for (int g = 0; g < ngaps; g++) ss.gaps[g] = gapArr[g]; }
                    haveSS = true;
                };
                if (inbounds && ngaps == 0 && prevSS && prevSS->chrom == chrom && prevSS->strand == strand &&
                    overlap(prevSS->start, prevSS->stop, site2, site3)) {
                    const int betterScore = max(score, prevSS->score);
                    const int minStart = min(prevSS->start, site2), maxStop = max(prevSS->stop, site3);
                    const bool perfect2 = (prevSS->score == maxScore && fullyDefined);
                    const bool shortEnough = (maxStop - minStart < 2 * blen);
                    if (prevSS->start == site2 && prevSS->stop == site3) {
                        prevSS->score = betterScore;
                        prevSS->perfect = (prevSS->perfect || perfect1 || perfect2) ? 1 : 0;
                        if (prevSS->perfect) prevSS->semiperfect = 1;
                    } else if (shortEnough && prevSS->start == site2 && !prevSS->semiperfect) {
                        if (perfect2) { }
                        else if (perfect1) {
                            prevSS->stop = site3; if (prevSS->ngaps) prevSS->gaps[prevSS->ngaps - 1] = site3;
                            if (!prevSS->perfect) perfectsFound++;
                            prevSS->perfect = prevSS->semiperfect = 1;
                        } else {
                            prevSS->stop = maxStop; if (prevSS->ngaps) prevSS->gaps[prevSS->ngaps - 1] = maxStop;
                            setPerfect(*w.ix, *prevSS, rd);
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
                            setPerfect(*w.ix, *prevSS, rd);
                        }
                        prevSS->score = betterScore;
                    } else newSite(false);
                } else if (inbounds) newSite(true);
                if (haveSS) {
                    if (ssl.n >= ssl.cap) { ssl.overflow = true; finished = true; }
                    else {
                        ssl.v[ssl.n] = ss;
                        const int idx = ssl.n++;
                        if (ss.perfect) {
                            const bbidx_site *pv = prevIdx >= 0 ? &ssl.v[prevIdx] : nullptr;
                            if (!pv || !pv->perfect || !(pv->chrom == ss.chrom && pv->strand == ss.strand && overlap(ss.start, ss.stop, pv->start, pv->stop))) {
                                perfectsFound++;
                                if (p.quitAfterTwoPerfects && perfectsFound >= 2) { prevIdx = idx; break; }
                            }
                        }
                        prevIdx = idx;
                    }
                }
            }
        }
        for (;;) {
            const int col = listsPeek(L);
            if (col < 0 || L.value[col] != site) break;
            w.cWalk++;
            const int row = L.row[col] + 1;
            if (row < L.stop[col]) { L.row[col] = row; L.value[col] = adjustSite(w, L.sites[row], L.offs[col], baseChrom); }
            else {
                L.live[col] = false; L.nlive--;
                if (L.nlive < approxHitsCutoff) { finished = true; break; }
            }
            if (L.nlive == 0) break;
        }
    }
    bestScores[0] = max(bestScores[0], currentTopScore);
    bestScores[1] = max(bestScores[1], maxHits);
    bestScores[2] = max(bestScores[2], qcutoff);
    bestScores[3] = max(bestScores[3], bestqscore);
    bestScores[4] = mqs;
    bestScores[5] = perfectsFound;
}

__device__ long long valueOfElement(const int *offsets, int noffsets, const int *lengths, float keyWeight, int chunk,
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

__device__ int trimByGreedy(const DevIndex &ix, const int *offsets, const int *keyScores, int n, int maxHitLists, int *keys,
                            int baseKeyHitScore, int *lengths, int *lists) {
    const bbidx_params &p = ix.p;
    const float inv = __fdiv_rn(1.0f, (float)baseKeyHitScore);
    const int limit = max(SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch]) * n;
    const int limit2 = max(SMALL_LIST, ix.lengthHistogram[p.maxAverageListToSearch2]);
    const int limit3 = max(SMALL_LIST, ix.lengthHistogram[p.maxShortestListToSearch]);
    int sum = 0, initialHitCount = 0, shortest = INT_MAX - 1, shortest2 = INT_MAX;
    for (int i = 0; i < n; i++) {
        const int x = ix.counts[keys[i]];
        lengths[i] = x; sum += x; initialHitCount += (x == 0 ? 0 : 1);
        if (x > 0 && x < shortest2) { shortest2 = x; if (shortest2 < shortest) { shortest2 = shortest; shortest = x; } }
    }
    if (initialHitCount < p.minApproxHitsToKeep) return initialHitCount;
    if (shortest > limit3 && !p.slow) { for (int i = 0; i < n; i++) keys[i] = -1; return 0; }
    int hitsCount = initialHitCount;
    const long long EARLY = -50LL * 2000;
    while (hitsCount >= p.minApproxHitsToKeep && (sum > limit || sum / initialHitCount > limit2 || hitsCount > maxHitLists)) {
        for (int i = 0, j = 0; j < hitsCount; i++) if (lengths[i] > 0) lists[j++] = i;
        long long mn = LLONG_MAX, worstValue64 = 0; int worstIndex = -1; bool early = false;
        for (int i = 0; i < hitsCount; i++) {
            const float kw = __fmul_rn((float)keyScores[i], inv);          // weights[i]: indexed by list position, as in the reference
            const long long value = valueOfElement(offsets, n, lengths, kw, p.k, lists, hitsCount, i, p.pointsPerSite);
            if (value < mn) {
                if (mn < EARLY && i != 0) { worstIndex = i; worstValue64 = value; early = true; break; }
                mn = value; worstIndex = i;
            }
        }
        if (!early) worstValue64 = mn;
        const int worstValue = worstValue64 < INT_MIN ? INT_MIN : (worstValue64 > INT_MAX ? INT_MAX : (int)worstValue64);
        const int worst = lists[worstIndex];
        sum -= lengths[worst];
        if (worstValue > 0 || lengths[worst] < SMALL_LIST) return hitsCount;
        hitsCount--; lengths[worst] = 0; keys[worst] = -1;
    }
    return hitsCount;
}

__device__ int countHits(const DevIndex &ix, int *keys, int n, int maxLen) {
    int numHits = 0;
    for (int i = 0; i < n; i++) {
        const int key = keys[i];
        if (key >= 0) { const int len = ix.counts[key]; if (len > 0 && len < maxLen) numHits++; else keys[i] = -1; }
    }
    return numHits;
}
__device__ int shrink2(int *offsets, int *keys, int *keyScores, int n) {
    int j = 0;
    for (int i = 0; i < n; i++) if (keys[i] >= 0) { offsets[j] = offsets[i]; keys[j] = keys[i]; keyScores[j] = keyScores[i]; j++; }
    return j;
}
__device__ int getHits(const DevIndex &ix, const int *keys, int n, int block, int *starts, int *stops) {
    int numHits = 0;
    const int *bs = ix.starts[block], *st = ix.sites[block];
    for (int i = 0; i < n; i++) {
        const int key = keys[i];
        starts[i] = -1; stops[i] = -1;
        if (key >= 0 && ix.counts[key] > 0) {
            const int s0 = bs[key], x = bs[key + 1] - s0;
            if (x > 0 && st[s0] != -1) { starts[i] = s0; stops[i] = s0 + x; numHits++; }
        }
    }
    return numHits;
}

__global__ __launch_bounds__(64) void probe_kernel(const Params P) {
    const DevIndex &ix = P.ix;
    const bbidx_params &p = ix.p;
    Walker w;
    w.ix = &ix;
    w.c.shift = 31 - p.chromBits; w.c.siteMask = (int)(0xFFFFFFFFu >> (p.chromBits + 1));
    w.c.cpb = 1 << p.chromBits; w.c.lowMask = w.c.cpb - 1; w.c.highMask = ~w.c.lowMask;
    w.k = p.k; w.baseKeyHitScore = BASE_HIT_SCORE * p.k;
    w.indelPenalty = (w.baseKeyHitScore / 2) - 1; w.indelPenaltyMult = 20;
    w.maxPenalty = w.baseKeyHitScore - (1 + w.baseKeyHitScore / 8);
    w.scoreZ1Key = Z_MULT * p.k;
    w.cPrescan = w.cWalk = w.cExtend = w.cRefBytes = 0;
    unsigned cSites = 0;

    int keysOriginal[KB], keysP[KB], offsetsP[KB], keyScoresP[KB];
    int offsetsM[KB], keysM[KB], keyScoresM[KB];
    int starts[KB], stops[KB];
    int locArray[MAXLEN];
    Lists L;

    // second pass behind the wave kernel: nothing to do unless it left reads pending
    if (P.onlyPending && __hip_atomic_load(&P.queue[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    for (;;) {
        const long long r = (long long)atomicAdd(P.queue, 1u);
        if (r >= P.nreads) break;
        if (P.onlyPending && P.nsites[r] != NSITES_PENDING) continue;
        const bbidx_read rr = P.reads[r];
        const int blen = rr.len;
        int n = rr.nkeys;
        bbidx_site *out = P.sites + r * (long long)P.maxSites;
        if (n < 1 || n > KB || blen < p.k || blen > MAXLEN) { P.nsites[r] = (n < 1 || blen < p.k) ? 0 : -2; continue; }
        const uint8_t *bP = P.bases + rr.bases_off;
        const int8_t *qP = P.baseScores + rr.bases_off;
        const int *koff = P.keyinfo + rr.keys_off, *kscore = koff + n;
        if (P.rcOut) for (int i = 0; i < blen; i++) P.rcOut[rr.bases_off + i] = (uint8_t)complement_extended(bP[blen - 1 - i]);

        for (int i = 0; i < n; i++) {                                   // KeyRing.makeKeys
            int key = 0;
            for (int q = koff[i]; q < koff[i] + p.k; q++) { const int x = base_num(bP[q]); if (x < 0) { key = -1; break; } key = (key << 2) | x; }
            keysOriginal[i] = keysP[i] = key; offsetsP[i] = koff[i]; keyScoresP[i] = kscore[i];
        }
        const int maxLen = p.maxUsableLength;
        int numHits = countHits(ix, keysP, n, maxLen);
        if (numHits > 0) {
            const int trigger = (3 * n) / 4;
            if (numHits < 4 && numHits < trigger) { for (int i = 0; i < n; i++) keysP[i] = keysOriginal[i]; numHits = countHits(ix, keysP, n, (maxLen * 3) / 2); }
            if (numHits < 3 && numHits < trigger) { for (int i = 0; i < n; i++) keysP[i] = keysOriginal[i]; numHits = countHits(ix, keysP, n, maxLen * 2); }
            if (numHits < 3 && numHits < trigger) { for (int i = 0; i < n; i++) keysP[i] = keysOriginal[i]; numHits = countHits(ix, keysP, n, maxLen * 3); }
            if (numHits < 2 && numHits < trigger) { for (int i = 0; i < n; i++) keysP[i] = keysOriginal[i]; numHits = countHits(ix, keysP, n, maxLen * 5); }
        }
        const int nOriginal = n;
        if (numHits < n) n = shrink2(offsetsP, keysP, keyScoresP, n);
        if (p.trimByGreedy) {
            const int maxLists = max((int)(HIT_FRACTION_TO_RETAIN * n), MIN_LISTS_RETAIN);
            numHits = trimByGreedy(ix, offsetsP, keyScoresP, n, maxLists, keysP, w.baseKeyHitScore, starts, stops);
        }
        if (numHits < p.minApproxHitsToKeep) { P.nsites[r] = 0; continue; }
        if (numHits < n) n = shrink2(offsetsP, keysP, keyScoresP, n);
        for (int i = 0; i < n; i++) {
            offsetsM[i] = blen - (offsetsP[n - 1 - i] + p.k);
            keysM[i] = rc_key(keysP[n - 1 - i], p.k);
            keyScoresM[i] = keyScoresP[n - 1 - i];
        }
        const int mqs = maxQuickScore(w, offsetsP, keyScoresP, n);
        int bestScores[6] = {0, 0, 0, 0, 0, 0};
        const bool prescan = p.prescanQscore && numHits >= 5;
        int hitsCutoff = 0, qscoreCutoff = (int)(MIN_QSCORE_MULT * mqs);
        bool allBasesCovered = true;
        if (offsetsP[0] != 0) allBasesCovered = false;
        else if (offsetsP[n - 1] != (blen - p.k)) allBasesCovered = false;
        else for (int i = 1; i < n; i++) if (offsetsP[i] > offsetsP[i - 1] + p.k) { allBasesCovered = false; break; }
        const bool pretend = allBasesCovered || n >= nOriginal - 4 ||
                             (n >= 9 && (offsetsP[n - 1] - offsetsP[0] + p.k) > max(40, (int)(blen * .75f)));

        // prescanAllBlocks keeps one (count, score) pair per (block, strand); only the comparison against the cutoffs is
        // needed later, so up to 64 cycles are kept in two 64-bit masks plus the two running maxima.
        const int cpb = w.c.cpb;
        int precounts[64], prescores[64];
        int ncycles = 0;
        for (int chrom = p.minChrom; chrom <= p.maxChrom; chrom = ((chrom & w.c.highMask) + cpb)) ncycles += 2;
        if (ncycles > 64) { P.nsites[r] = -2; continue; }
        bool dead = false;
        if (prescan) {
            for (int i = 0; i < ncycles; i++) { precounts[i] = n; prescores[i] = mqs; }
            int bestqscore = 0, maxHits = 0, minHitsToScore = p.minApproxHitsToKeep, cycle = 0; bool earlyOut = false;
            for (int chrom = p.minChrom; chrom <= p.maxChrom && !earlyOut; chrom = ((chrom & w.c.highMask) + cpb)) {
                const int baseChrom = w.c.baseChrom(chrom);
                const int block = baseChrom >> p.chromBits;
                for (int pmi = 0; pmi < 2 && !earlyOut; pmi++, cycle++) {
                    const int *keys = pmi ? keysM : keysP, *ksc = pmi ? keyScoresM : keyScoresP, *offs = pmi ? offsetsM : offsetsP;
                    const int nh = getHits(ix, keys, n, block, starts, stops);
                    if (nh < minHitsToScore) { prescores[cycle] = -9999; precounts[cycle] = 0; }
                    else {
                        listsInit(w, L, block, starts, stops, offs, ksc, n, baseChrom);
                        int tq, th;
                        findMaxQscore2(w, L, baseChrom, minHitsToScore, bestqscore >= mqs && pretend, tq, th);
                        prescores[cycle] = tq; precounts[cycle] = th;
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
        if (dead) { P.nsites[r] = 0; continue; }

        int sumBS = 0;
        bool fullyDefined = true;
        for (int i = 0; i < blen; i++) { sumBS += qP[i]; if (bP[i] >= 128 || base_num(bP[i]) < 0) fullyDefined = false; }
        const int maxScore = 70 + (blen - 1) * 100 + sumBS;               // msa.maxQuality(baseScores)
        SiteList ssl; ssl.v = out; ssl.n = 0; ssl.cap = P.maxSites; ssl.overflow = false;
        int cycle = 0; bool quit = false;
        for (int chrom = p.minChrom; chrom <= p.maxChrom && !quit; chrom = ((chrom & w.c.highMask) + cpb)) {
            const int block = w.c.baseChrom(chrom) >> p.chromBits;
            for (int strand = 0; strand < 2 && !quit; strand++, cycle++) {
                if (!prescan || precounts[cycle] >= hitsCutoff || prescores[cycle] >= qscoreCutoff) {
                    const int *keys = strand ? keysM : keysP, *ksc = strand ? keyScoresM : keyScoresP, *offs = strand ? offsetsM : offsetsP;
                    const int nh = getHits(ix, keys, n, block, starts, stops);
                    if (nh >= p.minApproxHitsToKeep) {
                        Strand rd; rd.b = bP; rd.q = qP; rd.len = blen; rd.minus = (strand == 1);
                        slowWalk3(w, L, locArray, block, starts, stops, rd, ksc, offs, n, chrom, strand, ssl, bestScores,
                                  allBasesCovered, maxScore, fullyDefined);
                    }
                }
                if (p.quitAfterTwoPerfects && bestScores[5] >= 2) quit = true;
            }
        }
        P.nsites[r] = ssl.overflow ? -1 : ssl.n;
        cSites += (unsigned)ssl.n;
    }
    if (P.stats) {
        unsigned long long *st = P.stats + 8 * (blockIdx.x % STAT_SHARDS);
        atomicAdd(&st[0], (unsigned long long)w.cPrescan); atomicAdd(&st[1], (unsigned long long)w.cWalk);
        atomicAdd(&st[2], (unsigned long long)w.cExtend); atomicAdd(&st[3], (unsigned long long)w.cRefBytes);
        atomicAdd(&st[4], (unsigned long long)cSites);
    }
}

__global__ void build_fused_kernel(KeyEntry *out, const int *starts, const int *sites, const int *counts, int k, long long nkeys) {
    const long long key = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    const int rc = rc_key((int)key, k);
    KeyEntry e;
    e.cnt = counts[key]; e.cntRC = counts[rc];
    e.startF = starts[key]; e.lenF = starts[key + 1] - e.startF; e.firstF = e.lenF > 0 ? sites[e.startF] : 0;
    e.startR = starts[rc]; e.lenR = starts[rc + 1] - e.startR; e.firstR = e.lenR > 0 ? sites[e.startR] : 0;
    out[key] = e;
}

}  // namespace bbidx

// ------------------------------------------------------------------------------------------------ host side
#include "index_ctx.h"

static thread_local char g_ierr[256];
#define IHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_ierr, sizeof g_ierr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_ierr); return BBMAP_E_HIP; } } while (0)
static int ifail(int code, const char *m) { bbmap_set_error(m); return code; }

template <typename T>
static int upload(bbidx_ctx *c, const T *host, size_t count, const T **dev) {
    void *d = nullptr;
    IHIP(hipMalloc(&d, (count > 0 ? count : 1) * sizeof(T)));
    c->allocs.push_back(d);
    if (count > 0) IHIP(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T *)d;
    return BBMAP_OK;
}

int bbidx_finish_create(bbidx_ctx *c, const std::vector<const int *> &hs, const std::vector<const int *> &hsi) {
    const bbidx_params &p = c->dev.p;
    const size_t keyspace = (size_t)1 << (2 * p.k);
    const int nblocks = c->dev.nblocks;
    int rc = BBMAP_OK;
    {
        // fused per-key records for the wave kernel; if HBM cannot hold them the context stays on the per-lane kernel
        std::vector<const bbidx::KeyEntry *> hf((size_t)nblocks, nullptr);
        bool ok = true;
        for (int b = 0; b < nblocks && ok; b++) {
            void *f = nullptr;
            if (hipMalloc(&f, keyspace * sizeof(bbidx::KeyEntry)) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            c->allocs.push_back(f);
            hf[(size_t)b] = (const bbidx::KeyEntry *)f;
            hipLaunchKernelGGL(bbidx::build_fused_kernel, dim3((unsigned)((keyspace + 255) / 256)), dim3(256), 0, nullptr,
                               (bbidx::KeyEntry *)f, hs[(size_t)b], hsi[(size_t)b], c->dev.counts, p.k, (long long)keyspace);
            if (hipGetLastError() != hipSuccess) ok = false;
        }
        if (ok && hipDeviceSynchronize() != hipSuccess) ok = false;
        c->dev.fused = nullptr;
        if (ok) rc = upload(c, hf.data(), hf.size(), (const bbidx::KeyEntry *const **)&c->dev.fused);
        else c->kernelKind = BBIDX_KERNEL_LANE;
    }
    if (rc == BBMAP_OK) rc = bbidx_launch_init(c, &c->own);
    return rc;
}

int bbidx_launch_init(bbidx_ctx *c, bbidx_launch *ls) {
    IHIP(hipSetDevice(c->device));
    IHIP(hipMalloc(&ls->d_queue, 64));
    IHIP(hipMalloc(&ls->d_stats, bbidx::STAT_SHARDS * 64));
    IHIP(hipEventCreate(&ls->ev[0]));
    IHIP(hipEventCreate(&ls->ev[1]));
    ls->timed = false; ls->d_longWs = nullptr; ls->longBlocks = 0;
    return BBMAP_OK;
}
void bbidx_launch_free(bbidx_launch *ls) {
    if (!ls) return;
    if (ls->d_queue) (void)hipFree(ls->d_queue);
    if (ls->d_stats) (void)hipFree(ls->d_stats);
    if (ls->d_longWs) (void)hipFree(ls->d_longWs);
    if (ls->ev[0]) (void)hipEventDestroy(ls->ev[0]);
    if (ls->ev[1]) (void)hipEventDestroy(ls->ev[1]);
    *ls = bbidx_launch();
}

extern "C" int bbidx_create(int32_t device, const bbidx_index_desc *d, bbidx_ctx **out) {
    if (!d || !out) return ifail(BBMAP_E_ARG, "bbidx_create: null argument");
    *out = nullptr;
    const bbidx_params &p = d->params;
    if (p.k < 8 || p.k > 15 || p.chromBits < 0 || p.chromBits > 16 || d->nblocks < 1 || d->nchroms < 1)
        return ifail(BBMAP_E_ARG, "bbidx_create: bad index geometry (k must be 8..15)");
    if (p.profile != BBIDX_PROFILE_BBMAP && p.profile != BBIDX_PROFILE_PACBIO) return ifail(BBMAP_E_ARG, "bbidx_create: unknown profile");
    for (int b = 0; b < d->nblocks; b++)
        if (d->numSites[b] < 0 || (long long)d->numSites[b] > 0x7fffffffLL - 64)
            return ifail(BBMAP_E_ARG, "bbidx_create: a block holds more than 2^31 - 64 sites");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ifail(BBMAP_E_NODEVICE, "bbidx_create: no HIP device (no CPU path)");
    if (device < 0 || device >= ndev) return ifail(BBMAP_E_ARG, "bbidx_create: bad device ordinal");
    IHIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    IHIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ifail(BBMAP_E_NODEVICE, "bbidx_create: this build targets gfx950 only");
    bbidx_ctx *c = new (std::nothrow) bbidx_ctx();
    if (!c) return ifail(BBMAP_E_NOMEM, "bbidx_create: out of memory");
    c->device = device;
    c->kernelKind = BBIDX_KERNEL_AUTO;
    c->blocks = prop.multiProcessorCount * 8;
    c->totalSites = 0; c->maxReadLen = BBIDX_MAX_READ_LEN;
    for (int b = 0; b < d->nblocks; b++) c->totalSites += (long long)d->numSites[b];
    const size_t keyspace = (size_t)1 << (2 * p.k);
    int rc = BBMAP_OK;
    std::vector<const int *> hs((size_t)d->nblocks), hsi((size_t)d->nblocks);
    std::vector<const uint8_t *> hc((size_t)d->nchroms + 1, nullptr);
    c->dev.p = p; c->dev.nblocks = d->nblocks; c->dev.nchroms = d->nchroms;
    for (int b = 0; b < d->nblocks && rc == BBMAP_OK; b++) {
        rc = upload(c, d->starts[b], keyspace + 1, &hs[(size_t)b]);
        if (rc == BBMAP_OK) rc = upload(c, d->sites[b], (size_t)d->numSites[b], &hsi[(size_t)b]);
    }
    for (int ch = 1; ch <= d->nchroms && rc == BBMAP_OK; ch++) rc = upload(c, d->chromArr[ch], (size_t)d->chromArrLen[ch], &hc[(size_t)ch]);
    if (rc == BBMAP_OK) rc = upload(c, hs.data(), hs.size(), (const int *const **)&c->dev.starts);
    if (rc == BBMAP_OK) rc = upload(c, hsi.data(), hsi.size(), (const int *const **)&c->dev.sites);
    if (rc == BBMAP_OK) rc = upload(c, hc.data(), hc.size(), (const uint8_t *const **)&c->dev.chromArr);
    if (rc == BBMAP_OK) rc = upload(c, d->counts, keyspace, &c->dev.counts);
    if (rc == BBMAP_OK) rc = upload(c, d->lengthHistogram, (size_t)1001, &c->dev.lengthHistogram);
    if (rc == BBMAP_OK) rc = upload(c, d->chromArrLen, (size_t)d->nchroms + 1, &c->dev.chromArrLen);
    if (rc == BBMAP_OK) rc = upload(c, d->chromLengths, (size_t)d->nchroms + 1, &c->dev.chromLengths);
    if (rc == BBMAP_OK) rc = bbidx_finish_create(c, hs, hsi);
    if (rc != BBMAP_OK) { bbidx_destroy(c); return rc; }
    *out = c;
    return BBMAP_OK;
}

extern "C" void bbidx_destroy(bbidx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void *p : c->allocs) (void)hipFree(p);
    bbidx_launch_free(&c->own);
    delete c;
}

extern "C" int bbidx_find_batch_device(bbidx_ctx *c, void *stream_, int64_t n, const bbidx_read *reads,
                                       const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo,
                                       bbidx_site *sites, int32_t max_sites, int32_t *nsites) {
    return bbidx_find_batch_device_rc(c, stream_, n, reads, bases, baseScores, keyinfo, sites, max_sites, nsites, nullptr);
}

extern "C" int bbidx_find_batch_device_rc(bbidx_ctx *c, void *stream_, int64_t n, const bbidx_read *reads,
                                          const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo,
                                          bbidx_site *sites, int32_t max_sites, int32_t *nsites, uint8_t *bases_rc_out) {
    if (!c) return ifail(BBMAP_E_ARG, "bbidx_find_batch_device: null context");
    return bbidx_find_batch_device_with(c, &c->own, stream_, n, reads, bases, baseScores, keyinfo, sites, max_sites, nsites, bases_rc_out);
}

int bbidx_find_batch_device_with(bbidx_ctx *c, bbidx_launch *ls, void *stream_, int64_t n, const bbidx_read *reads,
                                 const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo,
                                 bbidx_site *sites, int32_t max_sites, int32_t *nsites, uint8_t *bases_rc_out) {
    if (!c || !ls) return ifail(BBMAP_E_ARG, "bbidx_find_batch_device: null context");
    if (n < 0 || n > 0x7fffffffLL || max_sites < 1) return ifail(BBMAP_E_ARG, "bbidx_find_batch_device: bad size");
    if (n == 0) return BBMAP_OK;
    if (!reads || !bases || !baseScores || !keyinfo || !sites || !nsites) return ifail(BBMAP_E_ARG, "bbidx_find_batch_device: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    IHIP(hipSetDevice(c->device));
    IHIP(hipMemsetAsync(ls->d_queue, 0, 64, stream));
    IHIP(hipMemsetAsync(ls->d_stats, 0, bbidx::STAT_SHARDS * 64, stream));
    bbidx::Params P;
    P.stats = ls->d_stats;
    P.ix = c->dev; P.reads = reads; P.bases = bases; P.baseScores = baseScores; P.keyinfo = keyinfo;
    P.sites = sites; P.nsites = nsites; P.nreads = n; P.maxSites = max_sites; P.queue = ls->d_queue;
    P.onlyPending = 0;
    P.rcOut = bases_rc_out;
    long long blocks = (n + 63) / 64;
    if (blocks > c->blocks) blocks = c->blocks;
    IHIP(hipEventRecord(ls->ev[0], stream));
    if (c->dev.p.profile == BBIDX_PROFILE_PACBIO || c->kernelKind == BBIDX_KERNEL_LONG) {
        // mapPacBio's reads (thousands of bases, hundreds of keys), or the long-read kernel asked for by name
        if (!c->dev.fused) return ifail(BBMAP_E_NOMEM, "bbidx_find_batch_device: the long-read kernel needs the fused key table, which could not be allocated");
        if (!ls->d_longWs) {
            ls->longBlocks = bbidx_long_blocks(c->dev.p.profile == BBIDX_PROFILE_PACBIO ? 1 : 0);
            if (ls->longBlocks < 1) return ifail(BBMAP_E_HIP, "bbidx_find_batch_device: the long-read kernel does not fit this device");
            IHIP(hipMalloc(&ls->d_longWs, (size_t)ls->longBlocks * (size_t)bbidx_long_workspace_ints_per_block() * 4));
        }
        const int rc = bbidx_launch_long(P, stream, c->dev.p.profile == BBIDX_PROFILE_PACBIO ? 1 : 0, ls->d_longWs, ls->longBlocks);
        if (rc != BBMAP_OK) return rc;
        IHIP(hipEventRecord(ls->ev[1], stream));
        ls->timed = true;
        return BBMAP_OK;
    }
    if (c->kernelKind == BBIDX_KERNEL_AUTO) {
        // one read per wavefront; reads it cannot take (more than 64 keys) are marked and picked up by the per-lane kernel
        // average list length >= 1/2: the variant with batched pops / bulk skips (it gates them per strand by list size);
        // below that (small genomes) the plain variant, which is a few per cent faster there
        bool longLists = c->totalSites * 2 >= (1LL << (2 * c->dev.p.k)) * (long long)c->dev.nblocks;
        if (const char *ev = getenv("BBIDX_LONG_LISTS")) { if (*ev) longLists = atoi(ev) != 0; }      // tests force either variant
        const int rc = bbidx_launch_wave(P, stream, longLists, c->maxReadLen);
        if (rc != BBMAP_OK) return rc;
        P.onlyPending = 1;
    }
    hipLaunchKernelGGL(bbidx::probe_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, P);
    IHIP(hipGetLastError());
    IHIP(hipEventRecord(ls->ev[1], stream));
    ls->timed = true;
    return BBMAP_OK;
}

extern "C" int bbidx_find_batch(bbidx_ctx *c, int64_t n, const bbidx_read *reads, const uint8_t *bases, const int8_t *baseScores,
                                int64_t bases_bytes, const int32_t *keyinfo, int64_t keyinfo_ints,
                                bbidx_site *sites, int32_t max_sites, int32_t *nsites) {
    if (!c) return ifail(BBMAP_E_ARG, "bbidx_find_batch: null context");
    if (n == 0) return BBMAP_OK;
    if (n < 0 || !reads || !bases || !baseScores || !keyinfo || !sites || !nsites || max_sites < 1)
        return ifail(BBMAP_E_ARG, "bbidx_find_batch: bad argument");
    for (int64_t i = 0; i < n; i++) {
        const bbidx_read &r = reads[i];
        if (r.len < 0 || r.nkeys < 0 || r.bases_off < 0 || r.keys_off < 0 || r.bases_off + r.len > bases_bytes ||
            r.keys_off + 2LL * r.nkeys > keyinfo_ints)
            return ifail(BBMAP_E_ARG, "bbidx_find_batch: a read lies outside its buffers");
        for (int q = 0; q < r.nkeys && q < BBIDX_PACBIO_MAX_KEYS; q++) {
            const int o = keyinfo[r.keys_off + q];
            if (o < 0 || o + c->dev.p.k > r.len) return ifail(BBMAP_E_ARG, "bbidx_find_batch: a key offset lies outside its read");
        }
    }
    IHIP(hipSetDevice(c->device));
    bbidx_read *dr = nullptr; uint8_t *db = nullptr; int8_t *dq = nullptr; int32_t *dk = nullptr, *dn = nullptr; bbidx_site *ds = nullptr;
    int rc = BBMAP_OK;
#define IGO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_ierr, sizeof g_ierr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_ierr); rc = BBMAP_E_HIP; goto done; } } while (0)
    IGO(hipMalloc(&dr, (size_t)n * sizeof(bbidx_read)));
    IGO(hipMalloc(&db, (size_t)(bases_bytes > 0 ? bases_bytes : 1)));
    IGO(hipMalloc(&dq, (size_t)(bases_bytes > 0 ? bases_bytes : 1)));
    IGO(hipMalloc(&dk, (size_t)(keyinfo_ints > 0 ? keyinfo_ints : 1) * 4));
    IGO(hipMalloc(&dn, (size_t)n * 4));
    IGO(hipMalloc(&ds, (size_t)n * (size_t)max_sites * sizeof(bbidx_site)));
    IGO(hipMemcpy(dr, reads, (size_t)n * sizeof(bbidx_read), hipMemcpyHostToDevice));
    IGO(hipMemcpy(db, bases, (size_t)bases_bytes, hipMemcpyHostToDevice));
    IGO(hipMemcpy(dq, baseScores, (size_t)bases_bytes, hipMemcpyHostToDevice));
    IGO(hipMemcpy(dk, keyinfo, (size_t)keyinfo_ints * 4, hipMemcpyHostToDevice));
    rc = bbidx_find_batch_device(c, nullptr, n, dr, db, dq, dk, ds, max_sites, dn);
    if (rc != BBMAP_OK) goto done;
    IGO(hipStreamSynchronize(nullptr));
    IGO(hipMemcpy(nsites, dn, (size_t)n * 4, hipMemcpyDeviceToHost));
    IGO(hipMemcpy(sites, ds, (size_t)n * (size_t)max_sites * sizeof(bbidx_site), hipMemcpyDeviceToHost));
done:
    if (dr) (void)hipFree(dr);
    if (db) (void)hipFree(db);
    if (dq) (void)hipFree(dq);
    if (dk) (void)hipFree(dk);
    if (dn) (void)hipFree(dn);
    if (ds) (void)hipFree(ds);
    return rc;
#undef IGO
}

// Work counters and duration of the last bbidx_find_batch_device launch (valid once its stream has been synchronised):
// stats[0..4] = list entries consumed by the prescan, by the walk, extendScore calls, reference bytes compared,
// site records written.
extern "C" int bbidx_last_stats(bbidx_ctx *c, int64_t *stats5, float *kernel_ms) {
    if (!c) return ifail(BBMAP_E_ARG, "bbidx_last_stats: null context");
    return bbidx_last_stats_with(c, &c->own, stats5, kernel_ms);
}
int bbidx_last_stats_with(bbidx_ctx *c, bbidx_launch *ls, int64_t *stats5, float *kernel_ms) {
    if (!c || !ls || !ls->timed) return ifail(BBMAP_E_ARG, "bbidx_last_stats: nothing launched yet");
    IHIP(hipSetDevice(c->device));
    IHIP(hipEventSynchronize(ls->ev[1]));
    if (kernel_ms) IHIP(hipEventElapsedTime(kernel_ms, ls->ev[0], ls->ev[1]));
    if (stats5) {
        std::vector<unsigned long long> h((size_t)bbidx::STAT_SHARDS * 8);
        IHIP(hipMemcpy(h.data(), ls->d_stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int j = 0; j < 5; j++) stats5[j] = 0;
        for (int s = 0; s < bbidx::STAT_SHARDS; s++) for (int j = 0; j < 5; j++) stats5[j] += (int64_t)h[(size_t)s * 8 + j];
    }
    return BBMAP_OK;
}

extern "C" int bbidx_set_max_read_len(bbidx_ctx *c, int32_t max_len) {
    if (!c || max_len < 1) return ifail(BBMAP_E_ARG, "bbidx_set_max_read_len: bad argument");
    c->maxReadLen = max_len;
    return BBMAP_OK;
}

extern "C" int bbidx_set_kernel(bbidx_ctx *c, int32_t kind) {
    if (!c || (kind != BBIDX_KERNEL_AUTO && kind != BBIDX_KERNEL_LANE && kind != BBIDX_KERNEL_LONG)) return ifail(BBMAP_E_ARG, "bbidx_set_kernel: bad argument");
    if (c->dev.p.profile == BBIDX_PROFILE_PACBIO && kind != BBIDX_KERNEL_LONG && kind != BBIDX_KERNEL_AUTO)
        return ifail(BBMAP_E_ARG, "bbidx_set_kernel: a BBIDX_PROFILE_PACBIO context only has the long-read kernel");
    if (kind != BBIDX_KERNEL_LANE && !c->dev.fused) return ifail(BBMAP_E_NOMEM, "bbidx_set_kernel: the fused key table could not be allocated");
    c->kernelKind = kind;
    return BBMAP_OK;
}
