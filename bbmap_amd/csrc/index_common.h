// Shared by the two index-probe kernels (index_probe.hip: one read per lane, any shape;
// index_probe_wave.hip: one read per wavefront, state in registers + LDS).
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "bbmap_amd.h"

namespace bbidx {

constexpr int KB = BBIDX_MAX_KEYS;
constexpr int MAXLEN = BBIDX_MAX_READ_LEN;
constexpr int BASE_HIT_SCORE = 100, Z_MULT = 20, Y_MULT = 10, SMALL_LIST = 20, MIN_LISTS_RETAIN = 6, MINGAP = 256;
constexpr float HIT_FRACTION_TO_RETAIN = 0.85f, MIN_SCORE_MULT = 0.15f, MIN_QSCORE_MULT = 0.025f, MIN_QSCORE_MULT2 = 0.1f;
constexpr float DYN_SCORE = 0.84f, DYN_QSCORE = 0.6f, DYN_QSCORE_PERFECT = 0.8f;
#define PRESCAN_QSCORE_THRESH (DYN_QSCORE * .95f)

constexpr int NSITES_PENDING = -3;   // written by the wave kernel for reads it leaves to the per-lane kernel
constexpr int STAT_SHARDS = 256;     // work counters are sharded over this many 64-byte lines

// One 32-byte record per key and block, built once by bbidx_create from the reference's arrays: everything
// BBIndex.getHits needs for a key on BOTH strands (COUNTS, list start/length, first list entry) in one cache
// line, instead of 3 dependent 4-byte gathers per strand from three 4^k-entry tables.
struct __attribute__((aligned(32))) KeyEntry {
    int cnt, cntRC;                // COUNTS[key], COUNTS[rc(key)]
    int startF, lenF, firstF;      // this block's list of `key`: starts[key], its length, sites[starts[key]]
    int startR, lenR, firstR;      // the same for rc(key)
};

struct DevIndex {
    bbidx_params p;
    int nblocks, nchroms;
    const int *const *starts;
    const int *const *sites;
    const int *counts;
    const int *lengthHistogram;
    const uint8_t *const *chromArr;
    const int *chromArrLen;
    const int *chromLengths;
    const KeyEntry *const *fused;  // per block, 4^k records (null when the table could not be allocated)
};

struct Params {
    DevIndex ix;
    const bbidx_read *reads;
    const uint8_t *bases;
    const int8_t *baseScores;
    const int *keyinfo;
    bbidx_site *sites;
    int *nsites;
    long long nreads;
    int maxSites;
    int onlyPending;               // per-lane kernel: process only reads whose nsites == NSITES_PENDING
    unsigned int *queue;           // [0] read queue of the per-lane kernel, [1] reads left pending by the wave kernel
    uint8_t *rcOut;                // optional: reverse complement of every probed read, same offsets as `bases`
    unsigned long long *stats;     // [STAT_SHARDS][8]: prescan entries, walk entries, extendScore calls, ref bytes, sites written
};

__device__ inline int base_num(int b) {
    switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; default: return -1; }
}
__device__ inline int rc_key(int kmer, int k) {
    int out = 0;
    for (int i = 0; i < k; i++) { out = (out << 2) | ((~kmer) & 3); kmer >>= 2; }
    return out;
}
// dna/AminoAcid.java:633-645 (baseToComplementExtended); 0xFF where the reference holds -1
__device__ inline int complement_extended(int b) {
    switch (b) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'M': return 'K'; case 'R': return 'Y'; case 'S': return 'W'; case 'V': return 'B';
        case 'W': return 'S'; case 'Y': return 'R'; case 'H': return 'D'; case 'K': return 'M';
        case 'D': return 'H'; case 'B': return 'V'; case 'N': return 'N'; case 'X': return 'X';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        case 'm': return 'k'; case 'r': return 'y'; case 's': return 'w'; case 'v': return 'b';
        case 'w': return 's'; case 'y': return 'r'; case 'h': return 'd'; case 'k': return 'm';
        case 'd': return 'h'; case 'b': return 'v'; case 'n': return 'n'; case 'x': return 'x';
        case 'U': return 'A'; case 'u': return 'a';
        case '?': return '?'; case ' ': return ' '; case '-': return '-'; case '*': return '*'; case '.': return '.';
    }
    return 0xFF;
}
__device__ inline int absdif(int a, int b) { return a > b ? a - b : b - a; }

struct Codec {
    int shift, siteMask, lowMask, highMask, cpb;
    __device__ inline int toNumber(int site, int chrom) const { return ((chrom & lowMask) << shift) | site; }
    __device__ inline int chromOf(int number, int baseChrom) const { return (int)((unsigned)number >> shift) + (baseChrom & highMask); }
    __device__ inline int siteOf(int number) const { return number & siteMask; }
    __device__ inline int baseChrom(int chrom) const { return max(0, chrom & highMask); }
};

__device__ inline int calcApproxHitsCutoff(const bbidx_params &p, int keys, int hits, int currentCutoff, bool perfect) {
    const int reduction = min(max(hits / p.hitReductionDiv, p.maxHitsReduction2), max(p.maximumMaxHitsReduction, keys / 8));
    int r = max(p.minApproxHitsToKeep, max(currentCutoff, hits - reduction));
    if (perfect) r = max(r, keys);
    return r;
}

// MultiStateAligner11tsJNI.calcAffineScore helpers, in plain points
__device__ inline int calcDelScoreApprox(int len) {      // MultiStateAligner11tsJNI.java:1347-1376 with approximateGaps
    if (len <= 0) return 0;
    int score = -472;
    if (len > MINGAP) { const int rem = len % 128, div = (len - 128) / 128; score += div * -2; len = rem + 128; }
    if (len > 80) { score += ((len - 80 + 3) / 4) * -1; len = 80; }
    if (len > 20) { score += (len - 20) * -1; len = 20; }
    if (len > 5) { score += (len - 5) * -9; len = 5; }
    if (len > 1) score += (len - 1) * -33;
    return score;
}
__device__ inline int insCum(int n) {                    // POINTS_INS_ARRAY_C[n], n in 1..5
    return -395 + (n > 1 ? (n - 1) * -39 : 0);
}
__device__ inline int subArr(int t) { return t > 5 ? -25 : (t > 1 ? -51 : -127); }   // POINTS_SUB_ARRAY[t]

}  // namespace bbidx

// launcher of the wave kernel (index_probe_wave.hip)
int bbidx_launch_wave(const bbidx::Params &P, hipStream_t stream, bool longLists, int maxReadLen);
