// Shared constants/types for the MultiStateAligner11ts HIP kernels (gfx950).
// Score/time cell encoding and point values follow jni/MultiStateAligner11tsJNI.c:18-98.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bbmap_amd.h"

namespace bbmsa {

constexpr int kScoreOffset = 11;
constexpr int kTimeMask = 0x7FF;
constexpr int kScoreMask = (int)0xFFFFF800;
constexpr int kMaxTime = 2047;

#define BBMSA_PT(x) ((x) * 2048)
constexpr int P_MATCH = BBMSA_PT(70), P_MATCH2 = BBMSA_PT(100);
constexpr int P_SUB = BBMSA_PT(-127), P_SUBR = BBMSA_PT(-147), P_SUB2 = BBMSA_PT(-51), P_SUB3 = BBMSA_PT(-25);
constexpr int P_INS = BBMSA_PT(-395), P_INS2 = BBMSA_PT(-39), P_INS3 = BBMSA_PT(-23), P_INS4 = BBMSA_PT(-8);
constexpr int P_DEL = BBMSA_PT(-472), P_DEL2 = BBMSA_PT(-33), P_DEL3 = BBMSA_PT(-9), P_DEL4 = BBMSA_PT(-1),
              P_DEL5 = BBMSA_PT(-1);
constexpr int P_DEL_REF_N = BBMSA_PT(-10), P_GAP = BBMSA_PT(-2);
constexpr int kBadOff = (-(((1 << 20) - 1) - 2000) - 1) * 2048;   // BADoff
constexpr int kMinOffScore = (-(((1 << 20) - 1) - 2000)) * 2048;  // MINoff_SCORE
constexpr int kNegInf = -(1 << 30);   // "no limit" for the unlimited fill; never reached by any score
constexpr int kGapLen = 128;

constexpr int kTableLen = 3072;       // upper bound of the delC / insC LDS tables: index time (<2048) + rows (<=640)
// A context's tables only need min(longer side + 2, 2048) + maxRows + 8 entries: a streak (time) never exceeds the longer
// matrix side + 1 (and is clamped below 2048), the "still needed" indel length never exceeds the rows.  Ints of LDS in front of the per-job areas:
__host__ __device__ inline int lds_table_ints(int tableLen) { return 2 * tableLen + 320; }   // + delExt[128], insExt[32], subExt[8] (padded to 192), mTab[32][4]

// Ints of LDS one job of the wavefront kernel takes behind the tables: horizLimit + ONE per column (an int), the window's reference
// bytes (one byte per column; until round 4 an int2 per column held both: 8 bytes, which kept the second DP context's 640-column
// jobs at two blocks per CU) and the reversed match string.
__host__ __device__ inline int lds_job_ints(int fastCols, int tmpBytes) { return (fastCols + 2) + ((fastCols + 2 + 3) >> 2) + ((tmpBytes + 3) >> 2); }

// closed forms of calcDelScoreOffset (jni/...c:316-336) and of the cumulative
// POINTSoff_INS_ARRAY_C table (MultiStateAligner11tsJNI.java:1582-1601)
__host__ __device__ inline int calc_del_off(int len) {
    if (len <= 0) return 0;
    int s = P_DEL;
    if (len > 80) { s += ((len - 80 + 3) / 4) * P_DEL5; len = 80; }
    if (len > 20) { s += (len - 20) * P_DEL4; len = 20; }
    if (len > 5) { s += (len - 5) * P_DEL3; len = 5; }
    if (len > 1) s += (len - 1) * P_DEL2;
    return s;
}
__host__ __device__ inline int calc_ins_cum_off(int len) {
    if (len <= 0) return 0;
    long long s = P_INS;
    int n = len;
    if (n > 20) { s += (long long)(n - 20) * P_INS4; n = 20; }
    if (n > 5) { s += (long long)(n - 5) * P_INS3; n = 5; }
    if (n > 1) s += (long long)(n - 1) * P_INS2;
    return s < kMinOffScore ? kMinOffScore : (int)s;
}

// dna/AminoAcid.java:614-624 + :365-367: A,C,G,T,U in either case are "fully defined".
__host__ __device__ inline bool fully_defined(int b) {
    const int u = b & ~32;   // fold case
    return b < 128 && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U');
}

// ---- scoring schemes.  The wavefront and narrow kernels are written for the 11ts constants above; the generic kernel is
// a template over one of these, so that the PacBio parameter set runs through the same literal statement.
struct Scheme11ts {      // jni/MultiStateAligner11tsJNI.c:18-98
    static constexpr int OFF = kScoreOffset, TMASK = kTimeMask, SMASK = kScoreMask, MAXT = kMaxTime;
    static constexpr int MATCH = P_MATCH, MATCH2 = P_MATCH2, SUB = P_SUB, SUBR = P_SUBR, SUB2 = P_SUB2, SUB3 = P_SUB3;
    static constexpr int INS = P_INS, INS2 = P_INS2, INS3 = P_INS3, INS4 = P_INS4;
    static constexpr int DEL = P_DEL, DEL2 = P_DEL2, DEL3 = P_DEL3, DEL4 = P_DEL4, DEL5 = P_DEL5;
    static constexpr int DEL_REF_N = P_DEL_REF_N, GAP = P_GAP;
    static constexpr int BAR_I1 = 2, BAR_D1 = 3;
    static constexpr int BADOFF = kBadOff;
    __host__ __device__ static inline int del_off(int len) { return calc_del_off(len); }
    __host__ __device__ static inline int ins_cum_off(int len) { return calc_ins_cum_off(len); }
    __host__ __device__ static inline int col0(int row) { return calc_ins_cum_off(row); }   // ...JNI.java:101-112
};
struct Scheme9PacBio {   // current/align2/MultiStateAligner9PacBio.java:2359-2439
    static constexpr int OFF = 9, TMASK = 0x1FF, SMASK = (int)0xFFFFFE00, MAXT = 511;
#define BBMSA_PB(x) ((x) * 512)
    static constexpr int MATCH = BBMSA_PB(90), MATCH2 = BBMSA_PB(100);
    static constexpr int SUB = BBMSA_PB(-137), SUBR = BBMSA_PB(-157), SUB2 = BBMSA_PB(-49), SUB3 = BBMSA_PB(-25);
    static constexpr int INS = BBMSA_PB(-205), INS2 = BBMSA_PB(-42), INS3 = BBMSA_PB(-23), INS4 = BBMSA_PB(-8);
    static constexpr int DEL = BBMSA_PB(-292), DEL2 = BBMSA_PB(-37), DEL3 = BBMSA_PB(-17), DEL4 = BBMSA_PB(-2), DEL5 = BBMSA_PB(-1);
    static constexpr int DEL_REF_N = BBMSA_PB(-10), GAP = BBMSA_PB(-2);
    static constexpr int BAR_I1 = 1, BAR_D1 = 1;
    static constexpr int BADOFF = (-(((1 << 22) - 1) - 2000) - 1) * 512;
#undef BBMSA_PB
    __host__ __device__ static inline int del_off(int len) {            // calcDelScoreOffset, ...9PacBio.java:2254-2275
        if (len <= 0) return 0;
        int s = DEL;
        if (len > 80) { s += ((len - 80 + 3) / 4) * DEL5; len = 80; }
        if (len > 20) { s += (len - 20) * DEL4; len = 20; }
        if (len > 5) { s += (len - 5) * DEL3; len = 5; }
        if (len > 1) s += (len - 1) * DEL2;
        return s;
    }
    __host__ __device__ static inline int ins_cum_off(int len) {        // calcInsScoreOffset, :2295-2310
        if (len <= 0) return 0;
        int s = INS;
        if (len > 20) { s += (len - 20) * INS4; len = 20; }
        if (len > 5) { s += (len - 5) * INS3; len = 5; }
        if (len > 1) s += (len - 1) * INS2;
        return s;
    }
    // column 0 as the constructor fills it (:91-98): its tiers switch at row 5 and row 20, one row earlier than
    // calcInsScoreOffset's, so it is its own closed form
    __host__ __device__ static inline int col0(int row) {
        if (row <= 0) return 0;
        int s = INS;
        if (row >= 20) { s += (row - 19) * INS4; row = 19; }
        if (row >= 5) { s += (row - 4) * INS3; row = 4; }
        if (row >= 2) s += (row - 1) * INS2;
        return s;
    }
};

// number of jobs of a launch: a host value, or a counter a previous kernel of the same stream left on the device
__device__ inline long long job_count(long long njobs, const unsigned int *njobs_dev) {
    if (!njobs_dev) return njobs;
    const long long n = (long long)__hip_atomic_load(njobs_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return n < njobs ? n : njobs;
}

struct FillParams {
    const bbmsa_job *jobs;
    const uint8_t *reads;
    const uint8_t *refs;
    bbmsa_result *results;
    uint8_t *match;
    long long njobs;
    const unsigned int *njobs_dev; // when set, the job count is read from here on the device (njobs = capacity)
    unsigned int *queue;          // work-queue head (zeroed before launch)
    unsigned int *dirbuf;         // traceback direction nibbles, one slot per resident job
    long long dir_slot_dwords;
    const int *list;              // job indices to process (NULL = all njobs), filled by the narrow kernel or the width sort
    const unsigned int *list_count;
    int priority;                 // s_setprio for the launch's wavefronts (0..3): the one-job-per-block passes are a dependent chain per job
                                  // and share their SIMDs with the throughput passes of the other stream
    int *slow_list;               // jobs the fast kernel hands to the generic kernel
    unsigned int *slow_count;
    int match_stride;
    int lanesPerJob;              // 16, 32 or 64
    int fastCols;                 // LDS capacity (columns) per job in the fast kernel
    int tmpBytes;                 // LDS bytes per job for the reversed match string
    int tableLen;                 // entries of the delC / insC tables (see lds_table_ints)
    int maxRows, maxColumns;      // context limits (MSA(maxRows_, maxColumns_))
    int bandwidth;
    float bandwidthRatio;
    // matrix-materialising mode (the legacy per-call JNI shape, msa_legacy.hip; kernels instantiated with MAT only): job j's three
    // score planes go to planes + plane_off[j], each rows x columns ints (state-major; rows 1..rows, columns 1..columns of the matrix), its
    // vertLimit[0..rows] / horizLimit[0..columns] to limits + limits_off[j] (rows + 1 ints, then columns + 1)
    int *planes;
    const long long *plane_off;
    int *limits;
    const long long *limits_off;
};

// one job per lane, a band of diagonals in registers (msa_fill_narrow.hip)
struct NarrowParams {
    const bbmsa_job *jobs;
    const uint8_t *reads;
    const uint8_t *refs;
    bbmsa_result *results;
    uint8_t *match;
    long long njobs;
    const unsigned int *njobs_dev; // see FillParams
    unsigned int *queue;          // work-queue head (zeroed before launch)
    int *fast_list;               // jobs left to the wavefront kernel
    unsigned int *fast_count;
    unsigned long long *dirbuf;   // per resident wave: (maxRows + 1) x 64 lanes x 8 bytes of direction nibbles
    unsigned int *stats;          // [0] jobs finished here, [1] candidates that left the band (handed on)
    int match_stride;
    int maxRows, maxColumns;
    int bandwidth;
    float bandwidthRatio;
    int maxSlack;                 // candidate filter: maxQuality(rows) - minScore (points) at most this
};

struct GenericParams {
    const bbmsa_job *jobs;
    const uint8_t *reads;
    const uint8_t *refs;
    bbmsa_result *results;
    uint8_t *match;
    const int *list;              // job indices to process (NULL = all)
    const unsigned int *list_count;
    long long njobs;
    const unsigned int *njobs_dev; // see FillParams
    int *matrix;                  // per-thread-slot 3*(maxRows+1)*(maxColumns+1) ints
    int *limits;                  // per-thread-slot vertLimit/horizLimit
    unsigned int *queue;
    int match_stride;
    int maxRows, maxColumns;
    int bandwidth;
    float bandwidthRatio;
};

}  // namespace bbmsa
