/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see msa11ts_oracle.h for the rules).
 *
 * CPU restatement of align2.MultiStateAligner11ts / its JNI C twin, written
 * from the reference's behaviour; every function names the lines it follows
 * (paths relative to /root/reference).  Arithmetic is 32-bit two's complement
 * exactly like the C/Java; a DP cell is (score<<11 | time).
 */
#include "msa11ts_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants: jni/MultiStateAligner11tsJNI.c:18-98 ------------------------------ */
enum { ST_MS = 0, ST_DEL = 1, ST_INS = 2 };
enum { K_GAPLEN = 128, K_GAPBUFFER = 64, K_GAPBUFFER2 = 128, K_MINGAP = 256,
       K_GREF_CUSHION = 128 };          /* align2/Shared.java:21-25, MSA.java:761 */
#define K_GAPC '-'

#define SH(x) ((int32_t)((x) * (1 << ORC_SCOREOFFSET)))   /* x << SCOREOFFSET without UB on negatives */

#ifdef ORC_PACBIO   /* current/align2/MultiStateAligner9PacBio.java:2375-2407 */
enum { PTS_NOREF = 0, PTS_NOCALL = 0, PTS_MATCH = 90, PTS_MATCH2 = 100,
       PTS_SUB = -137, PTS_SUBR = -157, PTS_SUB2 = -49, PTS_SUB3 = -25,
       PTS_INS = -205, PTS_INS2 = -42, PTS_INS3 = -23, PTS_INS4 = -8,
       PTS_DEL = -292, PTS_DEL2 = -37, PTS_DEL3 = -17, PTS_DEL4 = -2, PTS_DEL5 = -1,
       PTS_DEL_REF_N = -10, PTS_GAP = -2 };
enum { K_TIMESLIP = 4, K_MASK5 = 3, K_BARRIER_I1 = 1, K_BARRIER_D1 = 1,
       K_LIM3 = 5, K_LIM4 = 20, K_LIM5 = 80, K_MAX_TIME = 511 };
#define NTAB 8200     /* insNeeded / delNeeded reach the read length (<= 6019 rows) */
#else
enum { PTS_NOREF = 0, PTS_NOCALL = 0, PTS_MATCH = 70, PTS_MATCH2 = 100,
       PTS_SUB = -127, PTS_SUBR = -147, PTS_SUB2 = -51, PTS_SUB3 = -25,
       PTS_INS = -395, PTS_INS2 = -39, PTS_INS3 = -23, PTS_INS4 = -8,
       PTS_DEL = -472, PTS_DEL2 = -33, PTS_DEL3 = -9, PTS_DEL4 = -1, PTS_DEL5 = -1,
       PTS_DEL_REF_N = -10, PTS_GAP = -2 /* 0-max(1,128/64) */ };
enum { K_TIMESLIP = 4, K_MASK5 = 3, K_BARRIER_I1 = 2, K_BARRIER_D1 = 3,
       K_LIM3 = 5, K_LIM4 = 20, K_LIM5 = 80, K_MAX_TIME = 2047 };
#define NTAB 604
#endif

#define K_MAX_SCORE ((((1 << (32 - ORC_TIMEBITS - 1)) - 1)) - 2000)
#define K_MIN_SCORE (0 - K_MAX_SCORE)
#define K_BAD (K_MIN_SCORE - 1)
#define K_BADOFF SH(K_BAD)
#define K_MINOFF_SCORE SH(K_MIN_SCORE)

static int32_t T_INS[NTAB], T_INSoff[NTAB], T_INS_C[NTAB], T_INSoff_C[NTAB];
static int32_t T_SUB[NTAB], T_SUBoff[NTAB];
static int8_t  T_B2N[128];
static int tables_ready = 0;

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* MultiStateAligner11tsJNI.java:1576-1625 ; dna/AminoAcid.java:614-624 */
static void build_tables(void) {
    if (tables_ready) return;
    for (int i = 1; i < NTAB; i++) {
        int p = (i > K_LIM4) ? PTS_INS4 : (i > K_LIM3) ? PTS_INS3 : (i > 1) ? PTS_INS2 : PTS_INS;
        T_INS[i] = p;
        T_INSoff[i] = SH(p);
        T_INS_C[i] = imax(K_MIN_SCORE, p + T_INS_C[i - 1]);
        T_INSoff_C[i] = imax(K_MINOFF_SCORE, SH(p) + T_INSoff_C[i - 1]);
        int s = (i > K_LIM3) ? PTS_SUB3 : (i > 1) ? PTS_SUB2 : PTS_SUB;
        T_SUB[i] = s;
        T_SUBoff[i] = SH(s);
    }
    memset(T_B2N, -1, sizeof T_B2N);
    const char *acgt = "ACGT";
    for (int i = 0; i < 4; i++) {
        T_B2N[(int)acgt[i]] = (int8_t)i;
        T_B2N[(int)acgt[i] + 32] = (int8_t)i;   /* lower case */
    }
    T_B2N['U'] = 3; T_B2N['u'] = 3;
    tables_ready = 1;
}

const int32_t *orc_points_ins_array(void)      { build_tables(); return T_INS; }
const int32_t *orc_pointsoff_ins_array(void)   { build_tables(); return T_INSoff; }
const int32_t *orc_points_ins_array_c(void)    { build_tables(); return T_INS_C; }
const int32_t *orc_pointsoff_ins_array_c(void) { build_tables(); return T_INSoff_C; }
const int32_t *orc_points_sub_array(void)      { build_tables(); return T_SUB; }
const int32_t *orc_pointsoff_sub_array(void)   { build_tables(); return T_SUBoff; }
const int8_t  *orc_base_to_number(void)        { build_tables(); return T_B2N; }

/* AminoAcid.java:365-367 -- bytes >=128 are "negative" in Java and never defined. */
static inline int fully_defined(uint8_t b) { return b < 128 && T_B2N[b] >= 0; }

/* ---- construction: MultiStateAligner11tsJNI.java:71-113 --------------------------- */
orc_msa *orc_msa_new(int maxRows, int maxColumns) {
    build_tables();
    orc_msa *m = (orc_msa *)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->maxRows = maxRows; m->maxColumns = maxColumns;
    const size_t W = (size_t)maxColumns + 1, XY = ((size_t)maxRows + 1) * W;
    m->packed = (int32_t *)calloc(3 * XY, sizeof(int32_t));
    m->vertLimit = (int32_t *)malloc(((size_t)maxRows + 1) * sizeof(int32_t));
    m->horizLimit = (int32_t *)malloc(W * sizeof(int32_t));
    m->grefbuffer = (uint8_t *)calloc((size_t)maxColumns + 2, 1);
    if (!m->packed || !m->vertLimit || !m->horizLimit || !m->grefbuffer) { orc_msa_free(m); return NULL; }
    for (int i = 0; i <= maxRows; i++) m->vertLimit[i] = K_BADOFF;
    for (int i = 0; i <= maxColumns; i++) m->horizLimit[i] = K_BADOFF;
    for (int s = 0; s < 3; s++) {
        int32_t *plane = m->packed + s * XY;
        for (int i = 1; i <= maxRows; i++)
            for (size_t j = 0; j < W; j++) plane[i * W + j] |= K_BADOFF;
        for (int i = 0; i <= maxRows; i++) {
            int32_t prev = (i < 2) ? 0 : plane[(size_t)(i - 1) * W];
            /* rows beyond the 604-entry table keep extending with the last tier (INS4);
             * the reference only ever builds 601-row matrices for this aligner. */
#ifdef ORC_PACBIO      /* ...9PacBio.java:91-98: tiers by `i<LIMIT`, one row earlier than the 11ts array */
            int32_t step = (i < 2) ? i * SH(PTS_INS) : (i < K_LIM3) ? SH(PTS_INS2) : (i < K_LIM4) ? SH(PTS_INS3) : SH(PTS_INS4);
#else
            int32_t step = (i < NTAB) ? T_INSoff[i] : SH(PTS_INS4);
#endif
            plane[(size_t)i * W] = prev + step;
        }
    }
    m->greflimit = m->greflimit2 = m->grefRefOrigin = -1;
    return m;
}

void orc_msa_free(orc_msa *m) {
    if (!m) return;
    free(m->packed); free(m->vertLimit); free(m->horizLimit); free(m->grefbuffer);
    free(m);
}

/* ---- closed-form indel run costs: jni/MultiStateAligner11tsJNI.c:316-359 ----------- */
int32_t orc_calc_del_score_offset(int len) {
    if (len <= 0) return 0;
    int32_t s = SH(PTS_DEL);
    if (len > K_LIM5) { s += ((len - K_LIM5 + K_MASK5) / K_TIMESLIP) * SH(PTS_DEL5); len = K_LIM5; }
    if (len > K_LIM4) { s += (len - K_LIM4) * SH(PTS_DEL4); len = K_LIM4; }
    if (len > K_LIM3) { s += (len - K_LIM3) * SH(PTS_DEL3); len = K_LIM3; }
    if (len > 1) s += (len - 1) * SH(PTS_DEL2);
    return s;
}
int32_t orc_calc_ins_score_offset(int len) {
    if (len <= 0) return 0;
    build_tables();
    return T_INSoff_C[len];               /* AFFINE_ARRAYS==1 branch, :342-343 */
}

/* per-step deletion extension cost, jni/...c:229-233 */
static inline int32_t del_extend_cost(int streak) {
    if (streak == 0) return SH(PTS_DEL);
    if (streak < K_LIM3) return SH(PTS_DEL2);
    if (streak < K_LIM4) return SH(PTS_DEL3);
    if (streak < K_LIM5) return SH(PTS_DEL4);
    return ((streak & K_MASK5) == 0) ? SH(PTS_DEL5) : 0;
}
static inline int32_t clamp_time(int32_t t) { return t > K_MAX_TIME ? K_MAX_TIME - K_MASK5 : t; }

/* first strict maximum over the last row, state-major: jni/...c:292-306 and :672-686 */
static void last_row_argmax(const orc_msa *m, int rows, int columns,
                            int32_t *bestScore, int *bestCol, int *bestState) {
    const size_t W = (size_t)m->maxColumns + 1, XY = ((size_t)m->maxRows + 1) * W;
    int32_t best = INT_MIN; int bc = -1, bs = -1;
    for (int s = 0; s < 3; s++) {
        const int32_t *rowp = m->packed + s * XY + (size_t)rows * W;
        for (int c = 1; c <= columns; c++) {
            int32_t x = rowp[c] & ORC_SCOREMASK;
            if (x > best) { best = x; bc = c; bs = s; }
        }
    }
    *bestScore = best; *bestCol = bc; *bestState = bs;
}

/* ---- fillUnlimited: jni/MultiStateAligner11tsJNI.c:100-314 -------------------------- */
void orc_fill_unlimited_raw(orc_msa *m, const uint8_t *read, int read_len,
                            const uint8_t *ref, int ref_len,
                            int refStartLoc, int refEndLoc, int32_t *result) {
    (void)ref_len;
    const int rows = read_len, columns = refEndLoc - refStartLoc + 1;
    const size_t W = (size_t)m->maxColumns + 1, XY = ((size_t)m->maxRows + 1) * W;
    int32_t *M = m->packed, *D = m->packed + XY, *I = m->packed + 2 * XY;
    const int32_t maxGain = (read_len - 1) * SH(PTS_MATCH2) + SH(PTS_MATCH);
    const int32_t subfloor = 0 - 2 * maxGain;
    const int insBarLo = K_BARRIER_I1, insBarHi = rows - K_BARRIER_I1, insBarCol = columns - 1;
    const int delBarLo = K_BARRIER_D1, delBarHi = rows - K_BARRIER_D1;
    if (rows > m->maxRows || columns > m->maxColumns) { result[0] = result[1] = result[2] = result[3] = -1; return; }

    for (int row = 1; row <= rows; row++) {
        const size_t up = (size_t)(row - 1) * W, cur = (size_t)row * W;
        const uint8_t call0 = (row < 2) ? (uint8_t)'?' : read[row - 2];
        const uint8_t call1 = read[row - 1];
        for (int col = 1; col <= columns; col++) {
            m->iterationsUnlimited++;
            const uint8_t ref0 = (col < 2) ? (uint8_t)'!' : ref[refStartLoc + col - 2];
            const uint8_t ref1 = ref[refStartLoc + col - 1];
            const int match = (call1 == ref1 && ref1 != 'N');
            const int prevMatch = (call0 == ref0 && ref0 != 'N');
            const int gap = (ref1 == K_GAPC);

            if (gap) {
                M[cur + col] = subfloor;
            } else {
                const int32_t dm = M[up + col - 1] & ORC_SCOREMASK;
                const int32_t dd = D[up + col - 1] & ORC_SCOREMASK;
                const int32_t di = I[up + col - 1] & ORC_SCOREMASK;
                const int32_t streak = M[up + col - 1] & ORC_TIMEMASK;
                int32_t a, bonus, tA;
                if (match) {
                    a = dm + (prevMatch ? SH(PTS_MATCH2) : SH(PTS_MATCH));
                    bonus = SH(PTS_MATCH);
                    tA = prevMatch ? streak + 1 : 1;
                } else {
                    if (ref1 != 'N' && call1 != 'N')
                        a = dm + (prevMatch ? (streak <= 1 ? SH(PTS_SUBR) : SH(PTS_SUB)) : T_SUBoff[streak + 1]);
                    else
                        a = dm + SH(PTS_NOCALL);
                    bonus = SH(PTS_SUB);
                    tA = prevMatch ? 1 : streak + 1;
                }
                const int32_t b = dd + bonus, c = di + bonus;
                int32_t score, time;
                if (a >= b && a >= c) { score = a; time = tA; }
                else if (b >= c)      { score = b; time = 1; }
                else                  { score = c; time = 1; }
                M[cur + col] = score | clamp_time(time);
            }

            if (row < delBarLo || row > delBarHi) {
                D[cur + col] = subfloor;
            } else {
                const int32_t streak = D[cur + col - 1] & ORC_TIMEMASK;
                int32_t a = (M[cur + col - 1] & ORC_SCOREMASK) + SH(PTS_DEL);
                int32_t b = (D[cur + col - 1] & ORC_SCOREMASK) + del_extend_cost(streak);
                if (ref1 == 'N') { a += SH(PTS_DEL_REF_N); b += SH(PTS_DEL_REF_N); }
                else if (gap)    { a += SH(PTS_GAP);       b += SH(PTS_GAP); }
                int32_t score, time;
                if (a >= b) { score = a; time = 1; } else { score = b; time = streak + 1; }
                D[cur + col] = score | clamp_time(time);
            }

            if (gap || (row < insBarLo && col > 1) || (row > insBarHi && col < insBarCol)) {
                I[cur + col] = subfloor;
            } else {
                const int32_t streak = I[up + col] & ORC_TIMEMASK;
                const int32_t a = (M[up + col] & ORC_SCOREMASK) + SH(PTS_INS);
                const int32_t b = (I[up + col] & ORC_SCOREMASK) + T_INSoff[streak + 1];
                int32_t score, time;
                if (a >= b) { score = a; time = 1; } else { score = b; time = streak + 1; }
                I[cur + col] = score | clamp_time(time);
            }
        }
    }
    int32_t best; int bc, bs;
    last_row_argmax(m, rows, columns, &best, &bc, &bs);
    result[0] = rows; result[1] = bc; result[2] = bs; result[3] = best >> ORC_SCOREOFFSET;
}

/* ---- fillLimitedX: jni/MultiStateAligner11tsJNI.c:361-704 --------------------------- */
void orc_fill_limited_raw(orc_msa *m, const uint8_t *read, int read_len,
                          const uint8_t *ref, int ref_len,
                          int refStartLoc, int refEndLoc, int minScore, int32_t *result) {
    (void)ref_len;
    const int rows = read_len, columns = refEndLoc - refStartLoc + 1;
    const size_t W = (size_t)m->maxColumns + 1, XY = ((size_t)m->maxRows + 1) * W;
    int32_t *M = m->packed, *D = m->packed + XY, *I = m->packed + 2 * XY;
    const int bandwidth = m->bandwidth; const float bandwidthRatio = m->bandwidthRatio;

    /* :392-393 (the float product is IEEE single, truncated toward zero) */
    int halfband = 0;
    if (!(bandwidth < 1 && bandwidthRatio <= 0)) {
        const int bwA = (bandwidth < 1) ? 9999999 : bandwidth;
        const int bwB = (bandwidthRatio <= 0) ? 9999999 : 8 + (int)((float)rows * bandwidthRatio);
        halfband = imax(imin(bwA, bwB), columns - rows + 8) / 2;
    }
    const int insBarHi = rows - K_BARRIER_I1, insBarCol = columns - 1;
    const int delBarHi = rows - K_BARRIER_D1;

    /* :398-403 last row starts out BAD */
    for (int s = 0; s < 3; s++)
        for (int c = 1; c <= columns; c++) m->packed[s * XY + (size_t)rows * W + c] = K_BADOFF;

    int minGoodCol = 1, maxGoodCol = columns;
    const int32_t minScore_off = SH(minScore);
    const int32_t maxGain = (read_len - 1) * SH(PTS_MATCH2) + SH(PTS_MATCH);
    const int32_t floorv = minScore_off - maxGain;
    const int32_t subfloor = floorv - 5 * SH(PTS_MATCH2);

    /* :413-438 best-case remaining gain bounds */
    int32_t *vertLimit = m->vertLimit, *horizLimit = m->horizLimit;
    vertLimit[rows] = minScore_off;
    int prevDef = 0;
    for (int i = rows - 1; i >= 0; i--) {
        if (fully_defined(read[i])) {
            vertLimit[i] = imax(vertLimit[i + 1] - (prevDef ? SH(PTS_MATCH2) : SH(PTS_MATCH)), floorv);
            prevDef = 1;
        } else {
            vertLimit[i] = imax(vertLimit[i + 1] - SH(PTS_NOCALL), floorv);
            prevDef = 0;
        }
    }
    horizLimit[columns] = minScore_off;
    prevDef = 0;
    for (int i = columns - 1; i >= 0; i--) {
        const uint8_t c = ref[refStartLoc + i];
        if (fully_defined(c)) {
            horizLimit[i] = imax(horizLimit[i + 1] - (prevDef ? SH(PTS_MATCH2) : SH(PTS_MATCH)), floorv);
            prevDef = 1;
        } else {
            horizLimit[i] = imax(horizLimit[i + 1] - ((prevDef && c == K_GAPC) ? SH(PTS_DEL) : SH(PTS_NOREF)), floorv);
            prevDef = 0;
        }
    }

    for (int row = 1; row <= rows; row++) {
        const int colStart = (halfband < 1) ? minGoodCol : imax(minGoodCol, row - halfband);
        const int colStop = (halfband < 1) ? maxGoodCol : imin(maxGoodCol, row + halfband * 2 - 1);
        minGoodCol = -1; maxGoodCol = -2;
        const int32_t vlimit = vertLimit[row];
        if (colStart < 0 || colStop < colStart) break;

        const size_t up = (size_t)(row - 1) * W, cur = (size_t)row * W;
        if (colStart > 1) {            /* :451-456 left sentinel */
            M[cur + colStart - 1] = subfloor; I[cur + colStart - 1] = subfloor; D[cur + colStart - 1] = subfloor;
        }
        const uint8_t call0 = (row < 2) ? (uint8_t)'?' : read[row - 2];
        const uint8_t call1 = read[row - 1];

        for (int col = colStart; col <= columns; col++) {
            const uint8_t ref0 = (col < 2) ? (uint8_t)'!' : ref[refStartLoc + col - 2];
            const uint8_t ref1 = ref[refStartLoc + col - 1];
            const int gap = (ref1 == K_GAPC);
            const int match = (call1 == ref1 && ref1 != 'N');
            const int prevMatch = (call0 == ref0 && ref0 != 'N');

            m->iterationsLimited++;
            const int32_t limit = imax(vlimit, horizLimit[col]);
            const int32_t limit3 = imax(floorv, match ? limit - SH(PTS_MATCH2) : limit - SH(PTS_SUB3));
            const int delNeeded = imax(0, row - col - 1);
            const int insNeeded = imax(0, (rows - row) - (columns - col) - 1);
            const int32_t delPenalty = orc_calc_del_score_offset(delNeeded);
            const int32_t insPenalty = (insNeeded <= 0) ? 0 : T_INSoff_C[insNeeded];

            const int32_t dm = M[up + col - 1] & ORC_SCOREMASK;
            const int32_t dd = D[up + col - 1] & ORC_SCOREMASK;
            const int32_t di = I[up + col - 1] & ORC_SCOREMASK;
            const int32_t lm = M[cur + col - 1] & ORC_SCOREMASK;
            const int32_t ld = D[cur + col - 1] & ORC_SCOREMASK;
            const int32_t um = M[up + col] & ORC_SCOREMASK;
            const int32_t ui = I[up + col] & ORC_SCOREMASK;

            /* match/sub plane :491-564 */
            if (gap || (dm <= limit3 && dd <= limit3 && di <= limit3)) {
                M[cur + col] = subfloor;
            } else {
                const int32_t streak = M[up + col - 1] & ORC_TIMEMASK;
                int32_t a, bonus, tA;
                if (match) {
                    a = dm + (prevMatch ? SH(PTS_MATCH2) : SH(PTS_MATCH));
                    bonus = SH(PTS_MATCH);
                    tA = prevMatch ? streak + 1 : 1;
                } else {
                    if (ref1 != 'N' && call1 != 'N')
                        a = dm + (prevMatch ? (streak <= 1 ? SH(PTS_SUBR) : SH(PTS_SUB)) : T_SUBoff[streak + 1]);
                    else
                        a = dm + SH(PTS_NOCALL);
                    bonus = SH(PTS_SUB);
                    tA = prevMatch ? 1 : streak + 1;
                }
                const int32_t b = dd + bonus, c = di + bonus;
                int32_t score, time;
                if (a >= b && a >= c) { score = a; time = tA; }
                else if (b >= c)      { score = b; time = 1; }
                else                  { score = c; time = 1; }
                const int32_t limit2 = (delNeeded > 0) ? limit - delPenalty
                                     : (insNeeded > 0) ? limit - insPenalty : limit;
                if (score >= limit2) { maxGoodCol = col; if (minGoodCol < 0) minGoodCol = col; }
                else score = subfloor;
                M[cur + col] = score | clamp_time(time);
            }

            /* deletion plane :566-617 */
            if ((lm <= limit && ld <= limit) || row < K_BARRIER_D1 || row > delBarHi) {
                D[cur + col] = subfloor;
            } else {
                const int32_t streak = D[cur + col - 1] & ORC_TIMEMASK;
                int32_t a = lm + SH(PTS_DEL);
                int32_t b = ld + del_extend_cost(streak);
                if (ref1 == 'N') { a += SH(PTS_DEL_REF_N); b += SH(PTS_DEL_REF_N); }
                else if (gap)    { a += SH(PTS_GAP);       b += SH(PTS_GAP); }
                int32_t score, time;
                if (a >= b) { score = a; time = 1; } else { score = b; time = streak + 1; }
                int32_t limit2;
                if (insNeeded > 0) limit2 = limit - insPenalty;
                else if (delNeeded > 0)
                    limit2 = limit - orc_calc_del_score_offset(time + delNeeded) + orc_calc_del_score_offset(time);
                else limit2 = limit;
                if (score >= limit2) { maxGoodCol = col; if (minGoodCol < 0) minGoodCol = col; }
                else score = subfloor;
                D[cur + col] = score | clamp_time(time);
            }

            /* insertion plane :619-658 */
            if (gap || (um <= limit && ui <= limit) || (row < K_BARRIER_I1 && col > 1)
                    || (row > insBarHi && col < insBarCol)) {
                I[cur + col] = subfloor;
            } else {
                const int32_t streak = I[up + col] & ORC_TIMEMASK;
                const int32_t a = um + SH(PTS_INS);
                const int32_t b = ui + T_INSoff[streak + 1];
                int32_t score, time;
                if (a >= b) { score = a; time = 1; } else { score = b; time = streak + 1; }
                int32_t limit2;
                if (delNeeded > 0) limit2 = limit - delPenalty;
                else if (insNeeded > 0)
                    limit2 = limit - orc_calc_ins_score_offset(time + insNeeded) + orc_calc_ins_score_offset(time);
                else limit2 = limit;
                if (score >= limit2) { maxGoodCol = col; if (minGoodCol < 0) minGoodCol = col; }
                else score = subfloor;
                I[cur + col] = score | clamp_time(time);
            }

            /* :660-668 row end + right sentinel in the row above */
            if (col >= colStop) {
                if (col > colStop && (maxGoodCol < col || halfband > 0)) break;
                if (row > 1) {
                    M[up + col + 1] = subfloor; I[up + col + 1] = subfloor; D[up + col + 1] = subfloor;
                }
            }
        }
    }

    int32_t best; int bc, bs;
    last_row_argmax(m, rows, columns, &best, &bc, &bs);
    result[0] = rows; result[1] = bc; result[2] = bs;
    if (best < minScore_off) { result[3] = best; result[4] = 1; }          /* :688-695 */
    else { result[3] = best >> ORC_SCOREOFFSET; result[4] = 0; }
}

/* ---- Java-side wrappers: MultiStateAligner11tsJNI.java:116-192 ---------------------- */
static int java_halfband(const orc_msa *m, int rows, int columns) {   /* :137-138 */
    if (m->bandwidth < 1 && m->bandwidthRatio <= 0) return 0;
    const int bwA = (m->bandwidth < 1) ? 9999999 : m->bandwidth;
    const int bwB = (m->bandwidthRatio <= 0) ? 9999999 : 8 + (int)((float)rows * m->bandwidthRatio);
    return imax(imin(bwA, bwB), columns - rows + 8) / 2;
}

static int fill_limited_x(orc_msa *m, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                          int refStartLoc, int refEndLoc, int minScore, int32_t *out4) {   /* :132-164 */
    m->rows = read_len; m->columns = refEndLoc - refStartLoc + 1;
    const int rows = m->rows, columns = m->columns;
    const int halfband = java_halfband(m, rows, columns);
    if (minScore < 1 || (columns + rows < 90)
            || ((halfband < 1 || halfband * 3 > columns) && (columns > read_len + imin(170, read_len + 20)))) {
        orc_fill_unlimited_raw(m, read, read_len, ref, ref_len, refStartLoc, refEndLoc, out4);
        return 1;
    }
    minScore -= 120;
    int32_t r5[5];
    orc_fill_limited_raw(m, read, read_len, ref, ref_len, refStartLoc, refEndLoc, minScore, r5);
    if (r5[4] == 1) return 0;
    memcpy(out4, r5, 4 * sizeof(int32_t));
    return 1;
}

int orc_fill_limited(orc_msa *m, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                     int refStartLoc, int refEndLoc, int minScore, const int32_t *gaps, int ngaps,
                     int32_t *out4) {                                                     /* :116-128 */
    if (!gaps) return fill_limited_x(m, read, read_len, ref, ref_len, refStartLoc, refEndLoc, minScore, out4);
    int32_t *g = (int32_t *)malloc((size_t)ngaps * sizeof(int32_t));
    memcpy(g, gaps, (size_t)ngaps * sizeof(int32_t));
    int lim = orc_make_gref(m, ref, ref_len, g, ngaps, refStartLoc, refEndLoc);
    free(g);
    if (lim < 0) return 0;
    return fill_limited_x(m, read, read_len, m->grefbuffer, m->maxColumns + 2, 0, m->greflimit, minScore, out4);
}

void orc_fill_unlimited(orc_msa *m, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                        int refStartLoc, int refEndLoc, const int32_t *gaps, int ngaps, int32_t *out4) { /* :166-192 */
    if (!gaps) { orc_fill_unlimited_raw(m, read, read_len, ref, ref_len, refStartLoc, refEndLoc, out4); return; }
    int32_t *g = (int32_t *)malloc((size_t)ngaps * sizeof(int32_t));
    memcpy(g, gaps, (size_t)ngaps * sizeof(int32_t));
    orc_make_gref(m, ref, ref_len, g, ngaps, refStartLoc, refEndLoc);
    free(g);
    orc_fill_unlimited_raw(m, read, read_len, m->grefbuffer, m->maxColumns + 2, 0, m->greflimit, out4);
}

/* ---- path walk shared by traceback2 and score2 -------------------------------------- */
static inline int32_t cell(const orc_msa *m, int state, int row, int col) {
    const size_t W = (size_t)m->maxColumns + 1, XY = ((size_t)m->maxRows + 1) * W;
    return m->packed[(size_t)state * XY + (size_t)row * W + (size_t)col];
}
/* predecessor-state rule, MultiStateAligner11tsJNI.java:389-443 == :577-611 */
static int prev_state(const orc_msa *m, int state, int row, int col) {
    const int32_t time = cell(m, state, row, col) & ORC_TIMEMASK;
    if (time > 1) return state;
    if (state == ST_MS) {
        const int32_t a = cell(m, ST_MS, row - 1, col - 1) & ORC_SCOREMASK;
        const int32_t b = cell(m, ST_DEL, row - 1, col - 1) & ORC_SCOREMASK;
        const int32_t c = cell(m, ST_INS, row - 1, col - 1) & ORC_SCOREMASK;
        if (a >= b && a >= c) return ST_MS;
        return (b >= c) ? ST_DEL : ST_INS;
    } else if (state == ST_DEL) {
        const int32_t a = cell(m, ST_MS, row, col - 1) & ORC_SCOREMASK;
        const int32_t b = cell(m, ST_DEL, row, col - 1) & ORC_SCOREMASK;
        return (a >= b) ? ST_MS : ST_DEL;
    } else {
        const int32_t a = cell(m, ST_MS, row - 1, col) & ORC_SCOREMASK;
        const int32_t b = cell(m, ST_INS, row - 1, col) & ORC_SCOREMASK;
        return (a >= b) ? ST_MS : ST_INS;
    }
}

/* MultiStateAligner11tsJNI.java:376-495 */
int orc_traceback2(orc_msa *m, const uint8_t *read, const uint8_t *ref,
                   int refStartLoc, int refEndLoc, int row, int col, int state,
                   uint8_t *out, int cap) {
    (void)refEndLoc;
    const int tmpCap = row + col - 1 + 1;
    uint8_t *tmp = (uint8_t *)malloc((size_t)(tmpCap > 0 ? tmpCap : 1));
    int n = 0, gaps = 0;
    while (row > 0 && col > 0) {
        const int prev = prev_state(m, state, row, col);
        if (state == ST_MS) {
            const uint8_t c = read[row - 1], r = ref[refStartLoc + col - 1];
            if (c == r) tmp[n] = 'm';
            else if (!fully_defined(c) || !fully_defined(r)) tmp[n] = 'N';
            else tmp[n] = 'S';
            row--; col--;
        } else if (state == ST_DEL) {
            const uint8_t r = ref[refStartLoc + col - 1];
            if (r == K_GAPC) { tmp[n] = '-'; gaps++; } else tmp[n] = 'D';
            col--;
        } else {
            if (col == 0) tmp[n] = 'X';
            else if (col >= m->columns) tmp[n] = 'Y';
            else tmp[n] = 'I';
            row--;
        }
        state = prev; n++;
    }
    if (col != row) {
        while (row > 0) { tmp[n++] = 'X'; row--; col--; }
    }
    const int total = n + gaps * (K_GAPLEN - 1);
    if (total > cap) { free(tmp); return -1; }
    int j = 0;
    for (int i = n - 1; i >= 0; i--) {
        if (tmp[i] != K_GAPC) out[j++] = tmp[i];
        else { for (int k = 0; k < K_GAPLEN; k++) out[j++] = 'D'; }
    }
    free(tmp);
    return total;
}

/* MultiStateAligner11tsJNI.java:537-658 */
int orc_score2(orc_msa *m, const uint8_t *read, const uint8_t *ref,
               int refStartLoc, int refEndLoc, int maxRow, int maxCol, int maxState, int32_t *out8) {
    (void)read; (void)ref;
    int row = maxRow, col = maxCol, state = maxState;
    int32_t score = cell(m, maxState, maxRow, maxCol) & ORC_SCOREMASK;
    if (row < m->rows) {
        int difR = m->rows - row; const int difC = m->columns - col;
        while (difR > difC) { score += SH(PTS_NOREF); difR--; }
        row += difR; col += difR;
    }
    const int bestRefStop = refStartLoc + col - 1;
    int stateTime = 0;
    while (row > 0 && col > 0) {
        const int prev = prev_state(m, state, row, col);
        if (state == ST_MS) { row--; col--; }
        else if (state == ST_DEL) { col--; }
        else { row--; }
        if (col < 0) break;
        if (state == prev) stateTime++; else stateTime = 0;
        state = prev;
    }
    if (row > col) col -= row;
    const int bestRefStart = refStartLoc + col;
    score >>= ORC_SCOREOFFSET;
    int padLeft = 0, padRight = 0;
    if (bestRefStart < refStartLoc) padLeft = imax(0, refStartLoc - bestRefStart);
    else if (bestRefStart == refStartLoc && state == ST_INS) padLeft = stateTime;
    if (bestRefStop > refEndLoc) padRight = imax(0, bestRefStop - refEndLoc);
    else if (bestRefStop == refEndLoc && maxState == ST_INS)
        padRight = cell(m, maxState, maxRow, maxCol) & ORC_TIMEMASK;
    out8[0] = score; out8[1] = bestRefStart; out8[2] = bestRefStop;
    out8[3] = maxRow; out8[4] = maxCol; out8[5] = maxState;
    if (padLeft > 0 || padRight > 0) { out8[6] = padLeft; out8[7] = padRight; return 8; }
    return 6;
}

/* ---- gapped reference: MultiStateAligner11tsJNI.java:668-801 ------------------------- */
int orc_make_gref(orc_msa *m, const uint8_t *ref, int ref_len, int32_t *gaps, int ngaps,
                  int refStartLoc, int refEndLoc) {
    const int g0_old = gaps[0], gN_old = gaps[ngaps - 1];
    gaps[0] = imin(gaps[0], refStartLoc);
    gaps[ngaps - 1] = imax(gN_old, refEndLoc);
    m->grefRefOrigin = gaps[0];
    uint8_t *gref = m->grefbuffer;
    const int glen = m->maxColumns + 2;
    int gpos = 0;
    for (int i = 0; i < ngaps; i += 2) {
        const int x = gaps[i], y = gaps[i + 1];
        for (int r = x; r <= y; r++, gpos++) { if (gpos >= glen) goto overflow; gref[gpos] = ref[r]; }
        if (i + 2 < ngaps) {
            const int z = gaps[i + 2];
            const int gap = z - y - 1;
            const int rem = gap % K_GAPLEN;
            const int lim = y + K_GAPBUFFER + rem;
            const int div = (gap - K_GAPBUFFER2) / K_GAPLEN;
            for (int r = y + 1; r <= lim; r++, gpos++) { if (gpos >= glen) goto overflow; gref[gpos] = ref[r]; }
            for (int g = 0; g < div; g++, gpos++)      { if (gpos >= glen) goto overflow; gref[gpos] = K_GAPC; }
            for (int r = z - K_GAPBUFFER; r < z; r++, gpos++) { if (gpos >= glen) goto overflow; gref[gpos] = ref[r]; }
        }
    }
    m->greflimit = gpos;
    {
        const int lim = imin(glen, m->greflimit + K_GREF_CUSHION);
        for (int i = m->greflimit, r = refEndLoc + 1; i < lim; i++, r++) {
            gref[i] = (r < ref_len) ? ref[r] : (uint8_t)'N';
            m->greflimit2 = i;
        }
    }
    gaps[0] = g0_old; gaps[ngaps - 1] = gN_old;
    return m->greflimit;
overflow:
    gaps[0] = g0_old; gaps[ngaps - 1] = gN_old;
    return -1;
}

static int from_gapped(const orc_msa *m, int point) {    /* :759-779 */
    if (point <= 0) return m->grefRefOrigin + point;
    for (int i = 0, j = m->grefRefOrigin; i < m->greflimit2; i++) {
        if (i == point) return j;
        j += (m->grefbuffer[i] == K_GAPC) ? K_GAPLEN : 1;
    }
    return INT_MIN;
}
static int to_gapped(const orc_msa *m, int point) {      /* :781-801 */
    if (point <= m->grefRefOrigin) return point - m->grefRefOrigin;
    for (int i = 0, j = m->grefRefOrigin; i < m->greflimit2; i++) {
        if (j == point) return i;
        j += (m->grefbuffer[i] == K_GAPC) ? K_GAPLEN : 1;
    }
    return INT_MIN;
}

int orc_traceback(orc_msa *m, const uint8_t *read, const uint8_t *ref, int refStartLoc, int refEndLoc,
                  int row, int col, int state, int gapped, uint8_t *out, int cap) {    /* :362-372 */
    if (!gapped) return orc_traceback2(m, read, ref, refStartLoc, refEndLoc, row, col, state, out, cap);
    const int gstart = to_gapped(m, refStartLoc), gstop = to_gapped(m, refEndLoc);
    return orc_traceback2(m, read, m->grefbuffer, gstart, gstop, row, col, state, out, cap);
}

int orc_score(orc_msa *m, const uint8_t *read, const uint8_t *ref, int refStartLoc, int refEndLoc,
              int maxRow, int maxCol, int maxState, int gapped, int32_t *out8) {       /* :499-531 */
    if (!gapped) return orc_score2(m, read, ref, refStartLoc, refEndLoc, maxRow, maxCol, maxState, out8);
    const int gstart = to_gapped(m, refStartLoc), gstop = to_gapped(m, refEndLoc);
    const int n = orc_score2(m, read, m->grefbuffer, gstart, gstop, maxRow, maxCol, maxState, out8);
    out8[1] = from_gapped(m, out8[1]);
    out8[2] = from_gapped(m, out8[2]);
    return n;
}

/* MSA.java:103-134 */
int orc_fill_and_score_limited(orc_msa *m, const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                               int refStartLoc, int refEndLoc, int minScore, const int32_t *gaps, int ngaps,
                               int32_t *out8, int32_t *max4) {
    const int a = imax(0, refStartLoc);
    int b = imin(ref_len - 1, refEndLoc);
    int32_t mx[4];
    if (!gaps) {
        if (b - a >= m->maxColumns) b = imin(ref_len - 1, a + m->maxColumns - 1);
        if (!orc_fill_limited(m, read, read_len, ref, ref_len, a, b, minScore, NULL, 0, mx)) return 0;
        if (max4) memcpy(max4, mx, sizeof mx);
        return orc_score(m, read, ref, a, b, mx[0], mx[1], mx[2], 0, out8);
    }
    if (!orc_fill_limited(m, read, read_len, ref, ref_len, a, b, minScore, gaps, ngaps, mx)) return 0;
    if (max4) memcpy(max4, mx, sizeof mx);
    return orc_score(m, read, ref, a, b, mx[0], mx[1], mx[2], 1, out8);
}

/* ---- ungapped scoring: MultiStateAligner11tsJNI.java:1034-1318 ---------------------- */
static int score_no_indels_core(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                                const int8_t *baseScores, int refStart, uint8_t *match) {
    int score = 0, mode = -1, timeInMode = 0;
    int readStart = 0, readStop = read_len;
    const int refStop = refStart + read_len;
    if (refStart < 0) { readStart = -refStart; score += PTS_NOREF * readStart; }
    if (refStop > ref_len) { const int dif = refStop - ref_len; readStop -= dif; score += PTS_NOREF * dif; }
    for (int i = readStart; i < readStop; i++) {
        const uint8_t c = read[i], r = ref[refStart + i];
        if (c == r && c != 'N') {
            if (mode == ST_MS) { timeInMode++; score += PTS_MATCH2; }
            else { timeInMode = 0; score += PTS_MATCH; }
            if (baseScores) score += baseScores[i];
            if (match) match[i] = 'm';
            mode = ST_MS;
        } else if (c >= 128 || c == 'N') {
            score += PTS_NOCALL; if (match) match[i] = 'N';
        } else if (r >= 128 || r == 'N') {
            score += PTS_NOREF; if (match) match[i] = 'N';
        } else {
            if (match) match[i] = 'S';
            if (mode == 3) timeInMode++; else timeInMode = 0;
            score += T_SUB[timeInMode + 1];
            mode = 3;
        }
    }
    return score;
}
int orc_score_no_indels(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                        const int8_t *baseScores, int refStart) {
    build_tables();
    return score_no_indels_core(read, read_len, ref, ref_len, baseScores, refStart, NULL);
}
int orc_score_no_indels_match(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                              const int8_t *baseScores, int refStart, uint8_t *match) {
    build_tables();
    if (refStart < 0 || refStart + read_len > ref_len) return -99999;   /* :1186, :1257 */
    return score_no_indels_core(read, read_len, ref, ref_len, baseScores, refStart, match);
}
void orc_gen_match_no_indels(const uint8_t *read, int read_len, const uint8_t *ref, int ref_len,
                             int refStart, uint8_t *match) {           /* :1092-1108 */
    for (int i = 0, j = refStart; i < read_len; i++, j++) {
        const uint8_t c = read[i];
        const uint8_t r = (j < 0 || j >= ref_len) ? (uint8_t)'N' : ref[j];
        match[i] = (c == 'N' || r == 'N') ? 'N' : (c == r) ? 'm' : 'S';
    }
}

/* deletion run cost in plain points with long-gap compression, :1347-1376 */
int orc_calc_del_score(int len, int approximateGaps) {
    if (len <= 0) return 0;
    int score = PTS_DEL;
    if (approximateGaps && len > K_MINGAP) {
        const int rem = len % K_GAPLEN, div = (len - K_GAPBUFFER2) / K_GAPLEN;
        score += div * PTS_GAP;
        len = rem + K_GAPBUFFER2;
    }
    if (len > K_LIM5) { score += ((len - K_LIM5 + K_MASK5) / K_TIMESLIP) * PTS_DEL5; len = K_LIM5; }
    if (len > K_LIM4) { score += (len - K_LIM4) * PTS_DEL4; len = K_LIM4; }
    if (len > K_LIM3) { score += (len - K_LIM3) * PTS_DEL3; len = K_LIM3; }
    if (len > 1) score += (len - 1) * PTS_DEL2;
    return score;
}
int orc_calc_ins_score(int len) { build_tables(); return len <= 0 ? 0 : T_INS_C[len]; }   /* :1401-1405 */
int orc_max_quality(int numBases) { return PTS_MATCH + (numBases - 1) * PTS_MATCH2; }       /* :1321-1323 */
int orc_max_imperfect_score(int numBases) {                                                /* :1331-1336 */
    return orc_max_quality(numBases) + imin(PTS_DEL, PTS_INS - PTS_MATCH2);
}

/* index location-array scoring, :871-942 (3-arg) and :945-1027 (minContig) */
int orc_calc_affine_score(const int32_t *locArray, int n, const int8_t *baseScores, int minContig) {
    build_tables();
    int contig = 0, maxContig = 0;
    int score = 0, lastLoc = -3, lastValue = -1, timeInMode = 0;
    for (int i = 0; i < n; i++) {
        const int loc = locArray[i];
        if (loc > 0) {
            if (loc == lastValue) { contig++; score += PTS_MATCH2 + baseScores[i]; }
            else if (loc == lastLoc || lastLoc < 0) {
                maxContig = imax(maxContig, contig); contig = 1;
                score += PTS_MATCH + baseScores[i];
            } else if (loc < lastLoc) {
                maxContig = imax(maxContig, contig); contig = 0;
                score += PTS_MATCH + baseScores[i];
                score += orc_calc_del_score(lastLoc - loc + 1, 1);   /* same tiering as :888-912 */
                timeInMode = 1;
            } else {
                maxContig = imax(maxContig, contig); contig = 0;
#ifdef ORC_PACBIO   /* MultiStateAligner9PacBio.java:1727-1742: dif = min(loc-lastLoc+1, 5), tier formula (= the cumulative table for dif <= 5) */
                score += PTS_MATCH + baseScores[i] + T_INS_C[imin(loc - lastLoc + 1, 5)];
#else
                score += PTS_MATCH + baseScores[i] + T_INS_C[imin(loc - lastLoc, 5)];
#endif
                timeInMode = 1;
            }
            lastLoc = loc;
        } else if (loc == -1) {
            if (lastValue < 0 && timeInMode > 0) { timeInMode++; score += T_SUB[timeInMode]; }
            else { score += PTS_SUB; timeInMode = 1; }
        } else {
            timeInMode = 0; score += PTS_NOCALL;
        }
        lastValue = loc;
    }
    if (minContig > 1 && imax(contig, maxContig) < minContig) score = imin(score, -50 * n);
    return score;
}
