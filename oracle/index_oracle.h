/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (same rules as msa11ts_oracle.h).
 *
 * CPU restatement of BBMap's k-mer index probe (align2.BBIndex, pure Java in the reference) and of
 * the index construction it reads (align2.IndexMaker4, BBIndex.analyzeIndex).
 *
 * PARITY STATUS: restatement only.  The reference has no native code and no golden vectors for
 * this path and there is no JVM in the image, so nothing here could be checked against a run of
 * the reference.  Tests pin invariants instead (planted reads are found at their true site with
 * the perfect score; documented asserts of the Java hold).
 */
#ifndef BBMAP_ORACLE_INDEX_H
#define BBMAP_ORACLE_INDEX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Tunables the reference keeps in mutable statics (BBIndex.java:3168-3305, AbstractIndex.java:100-160,
 * adjusted by genome size in BBMap.java:367-381). */
typedef struct orc_index_params {
    int32_t k;                       /* KEYLEN, BBMap.java:48 (13) */
    int32_t chromBits;               /* NUM_CHROM_BITS */
    int32_t minChrom, maxChrom;      /* 1-based chromosome range */
    int32_t maxIndel, maxIndel2;     /* MAX_INDEL 16000, MAX_INDEL2 = 2*MAX_INDEL */
    int32_t minApproxHitsToKeep;     /* 1 */
    int32_t kfilter;                 /* KFILTER */
    int32_t maxUsableLength, maxUsableLength2;
    int32_t maxHitsReduction2, maximumMaxHitsReduction, hitReductionDiv;
    int32_t quitAfterTwoPerfects, prescanQscore, trimByGreedy, slow;
    int32_t maxAverageListToSearch, maxAverageListToSearch2, maxShortestListToSearch; /* histogram indices */
    int64_t pointsPerSite;           /* Solver.POINTS_PER_SITE after analyzeIndex */
} orc_index_params;

typedef struct orc_index {
    orc_index_params p;
    int32_t nblocks;
    int32_t **starts;        /* per block: 4^k + 1 */
    int32_t **sites;         /* per block */
    int64_t *numSites;
    int32_t *counts;         /* COUNTS[4^k] */
    int32_t lengthHistogram[1001];
    int32_t nchroms;         /* chromosomes are 1..nchroms */
    const uint8_t **chromArr;/* [nchroms+1] */
    int32_t *chromArrLen;    /* array length of each chromosome */
    int32_t *chromLengths;   /* Data.chromLengths[chrom] */
} orc_index;

/* IndexMaker4.java:303-421 + BBIndex.analyzeIndex (BBIndex.java:101-191).
 * chroms[1..n] are padded byte arrays (bytes already mapped to A,C,G,T,N).  fractionToExclude is
 * FRACTION_GENOME_TO_EXCLUDE after the genome-size adjustment. */
orc_index *orc_index_build(int k, int chromBits, int nchroms, const uint8_t **chromArr, const int32_t *chromArrLen,
                           float fractionToExclude);
void orc_index_free(orc_index *ix);
/* a view over arrays the caller holds (nothing copied); free with orc_index_free_view */
orc_index *orc_index_from_arrays(const orc_index_params *p, int nblocks, int32_t **starts, int32_t **sites, const int64_t *numSites,
                                 int32_t *counts, const int32_t *lengthHistogram, int nchroms, const uint8_t **chromArr,
                                 const int32_t *chromArrLen);
void orc_index_free_view(orc_index *ix);

typedef struct orc_site {
    int32_t chrom, strand, start, stop, hits, score, perfect, semiperfect;
    int32_t ngaps;           /* entries in gaps[] (0 = null) */
    int32_t gaps[16];
} orc_site;

/* BBIndex.findAdvanced (BBIndex.java:394-400).  Returns the number of SiteScores written (<= cap), or -1 if cap
 * is too small.  stats (optional, 4 x int64): sites consumed by prescan, by the walk, calls to extendScore,
 * reference bytes touched by extendScore. */
int orc_index_find(const orc_index *ix, const uint8_t *basesP, const uint8_t *basesM, int len,
                   const int8_t *baseScoresP, const int32_t *keyScoresP, const int32_t *offsets, int nkeys,
                   orc_site *out, int cap, int64_t *stats);

/* KeyRing.makeOffsets(readlen, blocksize, density, minKeys) (KeyRing.java:255-297, :186-229).  Returns count. */
int orc_make_offsets(int readlen, int blocksize, float density, int minKeysDesired, int32_t *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
