/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Interface of mapper_oracle.c: the mapper control flow around the probe and the DP
 * (BBMapThread.processRead / processReadPair up to the end of the rescue stage), restated on the CPU.
 */
#ifndef BBMAP_ORACLE_MAPPER_H
#define BBMAP_ORACLE_MAPPER_H
#include <stdint.h>

#include "index_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_GAPS 16

/* stream.SiteScore (current/stream/SiteScore.java:999-1011); same layout as bbmap_msite (include/bbmap_amd.h) */
typedef struct orc_msite {
    int32_t chrom, strand, start, stop, hits;
    int32_t quickScore, score, slowScore, pairedScore;
    int32_t perfect, semiperfect, rescued;
    int32_t ngaps;                 /* 0 = gaps == null */
    int32_t gaps[ORC_MAX_GAPS];
    int32_t match_job;             /* index of the fill (job log) whose traceback belongs to this site, -1 = none */
    int32_t reserved[2];
} orc_msite;                       /* 128 bytes */

/* the mapper's settings (defaults: bbmap.sh) */
typedef struct orc_map_params {
    float minRatio;                /* MINIMUM_ALIGNMENT_SCORE_RATIO 0.56 */
    int32_t slowAlignPadding;      /* 4 */
    int32_t slowRescuePadding;     /* 8 */
    int32_t extraPadding;          /* 10 */
    int32_t tipSearchDist;         /* TIP_DELETION_SEARCH_RANGE 100; 0 = off */
    int32_t maxPairDist;           /* 32000 */
    int32_t averagePairDist;       /* INITIAL_AVERAGE_PAIR_DIST 100 */
    int32_t maxRescueDist;         /* 1200 */
    int32_t maxRescueMismatches;   /* 32 */
    int32_t maxTrimSitesToRetain;  /* 800 */
    int32_t trimList;              /* 1 */
    int32_t doRescue;              /* 1 */
    int32_t alignColumns;          /* BBIndex.ALIGN_COLUMNS 3000 */
    int32_t clearzone3;            /* PENALIZE_AMBIG ? 800 : 0 */
    int32_t msaMaxRows, msaMaxColumns;   /* the MSA instance: 601 x 3000 in the reference */
    int32_t finalStage;            /* 1: the whole of processRead / processReadPair (final pairing, ambiguity, genMatchString -> realign_new,
                                    * clipping, penalties: final_stage.inc); 0: stop after the rescue stage (rounds 1-3) */
} orc_map_params;

/* what BBMap prints for a read: stream.Read's mapping fields after processRead / processReadPair (current/stream/Read.java) */
typedef struct orc_final {
    int32_t mapped, chrom, strand, start, stop, mapScore;
    int32_t paired, ambiguous, perfect, rescued;
    int32_t match_len;             /* length of Read.match (long format), 0 = null */
    int32_t nsites;                /* sites left in the read's list */
} orc_final;                       /* 48 bytes */

typedef struct orc_mjob {          /* one MSA.fillAndScoreLimited call */
    int32_t read, seq, kind;       /* kind: 0 scoreSlow fill, 1 scoreSlow wider refill, 2 slowRescue; realign_new: 3 first fill, 4 padded
                                    * refill, 5 third fill, 6 fillUnlimited */
    int32_t strand, chrom, refStartLoc, refEndLoc, minScore, ngaps;
    int32_t score_len; int32_t score[8];
    int32_t match_len, pad_;
    int64_t iterations;
} orc_mjob;                        /* 88 bytes */

/* one read of a batch: the layout of bbidx_read (include/bbmap_amd.h) */
typedef struct orc_read { int64_t bases_off, keys_off; int32_t len, nkeys; } orc_read;     /* 24 bytes */

void orc_map_default_params(orc_map_params *P);
/* with -DORC_PACBIO the same source follows align2.BBMapThreadPacBio / BBMapPacBio.setDefaults (current/align2/BBMapPacBio.java:47-69):
 * processRead, scoreSlow and trimList of the two classes differ in nothing the path up to the end of scoreSlow reads (a diff shows
 * clearzone ratios of the later ambiguity policy, ALIGN_ROWS 6020 and ALIGN_COLUMNS 7600) */
double orc_map_reads(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired,
                     const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo, int cap,
                     orc_msite *sites, int32_t *nsites,
                     orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4);
double orc_map_reads_final(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired,
                           const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo, int cap,
                           orc_msite *sites, int32_t *nsites,
                           orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4,
                           orc_final *fin, uint8_t *fmatch, int fstride);
int orc_final_reads(const orc_index *ix, const orc_map_params *P, const orc_read *recs, int64_t n_reads, int paired, const uint8_t *bases,
                    int cap, orc_msite *sites, int32_t *nsites, orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride,
                    orc_final *fin, uint8_t *fmatch, int fstride);
float orc_ratio_paired(float R);
float orc_ratio_pre_rescue(float R);
double orc_map_batch(const orc_index *ix, const orc_map_params *P, const uint8_t *reads1, const uint8_t *reads2, int64_t n, int L,
                     const int32_t *offsets, const int32_t *keyScores, int nkeys, int cap,
                     orc_msite *sites1, int32_t *nsites1, orc_msite *sites2, int32_t *nsites2,
                     orc_mjob *log, int64_t logcap, int64_t *nlog, uint8_t *match, int matchStride, int threads, int64_t *stats4);
#ifdef __cplusplus
}
#endif
#endif
