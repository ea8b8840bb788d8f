/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * Plain-C CPU restatement of BBMap's MultiStateAligner11ts affine-gap DP
 * (the `usejni=t` path).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this.  The product (bbmap_amd/) never
 * links, imports or calls it.
 *
 * Parity status: PINNED by the known answers recorded in SURVEY.md section 8c
 * (outputs of the reference C observed by the survey) and the 400-problem
 * visited-fraction statistic of SURVEY.md section 7/H1; the reference C itself
 * is unbuildable here (needs <jni.h>, absent from the image), so there is no
 * oracle/_ref.  Java-only pieces (traceback2, score2, makeGref, scoreNoIndels)
 * have no runnable reference: they are restated from source and pinned only by
 * hand-derived cases (see tests/test_oracle_msa.py).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef BBMAP_ORACLE_MSA11TS_H
#define BBMAP_ORACLE_MSA11TS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* jni/MultiStateAligner11tsJNI.c:30-39; with -DORC_PACBIO the same source restates
 * align2.MultiStateAligner9PacBio (current/align2/MultiStateAligner9PacBio.java:2359-2369) and is
 * built into liboracle_pacbio.so (fills, traceback2 and score2 only: that class's scoreNoIndels /
 * calcAffineScore variants differ from 11ts in more than constants and are NOT restated). */
#ifdef ORC_PACBIO
#define ORC_TIMEBITS 9
#define ORC_TIMEMASK 0x1FF
#define ORC_SCOREMASK ((int32_t)0xFFFFFE00)
#else
#define ORC_TIMEBITS 11
#define ORC_TIMEMASK 0x7FF
#define ORC_SCOREMASK ((int32_t)0xFFFFF800)
#endif
#define ORC_SCOREOFFSET ORC_TIMEBITS

typedef struct orc_msa {
    int maxRows, maxColumns;          /* current/align2/MSA.java:66-69 */
    int32_t *packed;                  /* 3*(maxRows+1)*(maxColumns+1), MultiStateAligner11tsJNI.java:80 */
    int32_t *vertLimit, *horizLimit;  /* :82-83 */
    uint8_t *grefbuffer;              /* maxColumns+2, :81 */
    int greflimit, greflimit2, grefRefOrigin; /* :1446-1448 */
    int64_t iterationsLimited, iterationsUnlimited; /* MSA.java:858-859 */
    int rows, columns;                /* :1662-1663 */
    int bandwidth;                    /* MSA.java:864 (static in the reference) */
    float bandwidthRatio;             /* MSA.java:865 */
} orc_msa;

orc_msa *orc_msa_new(int maxRows, int maxColumns);
void orc_msa_free(orc_msa *m);

/* Score tables (POINTS*_ARRAY), MultiStateAligner11tsJNI.java:1576-1625. 604 entries each. */
const int32_t *orc_points_ins_array(void);
const int32_t *orc_pointsoff_ins_array(void);
const int32_t *orc_points_ins_array_c(void);
const int32_t *orc_pointsoff_ins_array_c(void);
const int32_t *orc_points_sub_array(void);
const int32_t *orc_pointsoff_sub_array(void);
/* dna/AminoAcid.java:614-624 (baseToNumber) */
const int8_t *orc_base_to_number(void);

/* jni/MultiStateAligner11tsJNI.c:100-314.  result[4] = {rows,maxCol,maxState,maxScore} */
void orc_fill_unlimited_raw(orc_msa *m, const uint8_t *read, int read_len,
                            const uint8_t *ref, int ref_len,
                            int refStartLoc, int refEndLoc, int32_t *result);

/* jni/MultiStateAligner11tsJNI.c:361-704.  result[5]; bandwidth/ratio taken from m. */
void orc_fill_limited_raw(orc_msa *m, const uint8_t *read, int read_len,
                          const uint8_t *ref, int ref_len,
                          int refStartLoc, int refEndLoc, int minScore, int32_t *result);

/* MultiStateAligner11tsJNI.java:116-164 (fillLimited + Java gate + minScore-=120).
 * gaps may be NULL.  Returns 1 and fills out[4] on success, 0 when the Java returns null. */
int orc_fill_limited(orc_msa *m, const uint8_t *read, int read_len,
                     const uint8_t *ref, int ref_len, int refStartLoc, int refEndLoc,
                     int minScore, const int32_t *gaps, int ngaps, int32_t *out4);

/* MultiStateAligner11tsJNI.java:166-192 */
void orc_fill_unlimited(orc_msa *m, const uint8_t *read, int read_len,
                        const uint8_t *ref, int ref_len, int refStartLoc, int refEndLoc,
                        const int32_t *gaps, int ngaps, int32_t *out4);

/* MultiStateAligner11tsJNI.java:376-495.  Returns match-string length (written to out, cap bytes),
 * or -1 if cap is too small. */
int orc_traceback2(orc_msa *m, const uint8_t *read, const uint8_t *ref,
                   int refStartLoc, int refEndLoc, int row, int col, int state,
                   uint8_t *out, int cap);
/* :362-372 (gapped dispatches through grefbuffer) */
int orc_traceback(orc_msa *m, const uint8_t *read, const uint8_t *ref,
                  int refStartLoc, int refEndLoc, int row, int col, int state, int gapped,
                  uint8_t *out, int cap);

/* MultiStateAligner11tsJNI.java:537-658.  Returns 6 or 8 (number of ints written to out8). */
int orc_score2(orc_msa *m, const uint8_t *read, const uint8_t *ref,
               int refStartLoc, int refEndLoc, int maxRow, int maxCol, int maxState,
               int32_t *out8);
/* :499-531 */
int orc_score(orc_msa *m, const uint8_t *read, const uint8_t *ref,
              int refStartLoc, int refEndLoc, int maxRow, int maxCol, int maxState, int gapped,
              int32_t *out8);

/* MSA.java:103-134.  Returns 0 for null, else 6 or 8; max4 (optional) receives fillLimited's result. */
int orc_fill_and_score_limited(orc_msa *m, const uint8_t *read, int read_len,
                               const uint8_t *ref, int ref_len, int refStartLoc, int refEndLoc,
                               int minScore, const int32_t *gaps, int ngaps,
                               int32_t *out8, int32_t *max4);

/* MultiStateAligner11tsJNI.java:668-757.  Returns greflimit or -1 on overflow. */
int orc_make_gref(orc_msa *m, const uint8_t *ref, int ref_len, int32_t *gaps, int ngaps,
                  int refStartLoc, int refEndLoc);

/* MultiStateAligner11tsJNI.java:1034-1089 / :1115-1171 (baseScores may be NULL) */
int orc_score_no_indels(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                        const int8_t *baseScores, int refStart);
/* :1174-1318 ; match gets read_len bytes.  Returns score or -99999. */
int orc_score_no_indels_match(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                              const int8_t *baseScores, int refStart, uint8_t *match);
/* :1092-1108 */
void orc_gen_match_no_indels(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                             int refStart, uint8_t *match);
/* :871-942 and :945-1027 (minContig<=1 selects the 3-argument form) */
int orc_calc_affine_score(const int32_t *locArray, int n, const int8_t *baseScores, int minContig);

/* :1321-1344, :1347-1376, :1401-1421 */
int orc_max_quality(int numBases);
int orc_max_imperfect_score(int numBases);
int orc_calc_del_score(int len, int approximateGaps);
int orc_calc_ins_score(int len);
/* jni/MultiStateAligner11tsJNI.c:316-359 */
int32_t orc_calc_del_score_offset(int len);
int32_t orc_calc_ins_score_offset(int len);

#ifdef __cplusplus
}
#endif
#endif
