/*
 * bbmap_amd.h -- C ABI of libbbmap_amd.so, the MI355X (gfx950) replacement for the
 * native side of BBMap's `usejni=t` seed-and-extend path.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point names the
 * reference interface it replaces (paths relative to the reference checkout).
 * Return convention: 0 = ok, <0 = error (see BBMAP_E_*); the library never calls exit()
 * (the reference's native code does: jni/MultiStateAligner11tsJNI.c:130-132).
 *
 * The library REQUIRES a HIP device: there is no CPU fallback.  bbmsa_create() fails with
 * BBMAP_E_NODEVICE when no gfx950 device is usable.
 */
#ifndef BBMAP_AMD_H
#define BBMAP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBMAP_AMD_ABI_VERSION 6

enum {
    BBMAP_OK = 0,
    BBMAP_E_NODEVICE = -1,   /* no HIP device / wrong architecture */
    BBMAP_E_ARG = -2,        /* bad argument (null pointer, negative size, shape beyond limits) */
    BBMAP_E_NOMEM = -3,      /* device or host allocation failed */
    BBMAP_E_HIP = -4,        /* a HIP runtime call failed; see bbmap_last_error() */
    BBMAP_E_SHAPE = -5       /* a job exceeds maxRows/maxColumns of the context */
};

/* Human-readable text of the last error on the calling thread. */
const char *bbmap_last_error(void);
int bbmap_abi_version(void);

/* =====================================================================================
 * MultiStateAligner11ts (affine-gap multi-state DP)
 *   replaces jni/MultiStateAligner11tsJNI.c (fillUnlimited :100-314, fillLimitedX :361-704)
 *   and fuses the Java-side walkers that read its `packed` matrix:
 *   current/align2/MultiStateAligner11tsJNI.java traceback2 :376-495, score2 :537-658,
 *   plus the call shape of current/align2/MSA.java fillAndScoreLimited :103-134.
 * ===================================================================================== */

/* job.flags: low 3 bits = fill mode, the rest are option bits */
enum {
    BBMSA_FILL_LIMITED_RAW   = 0, /* == C fillLimitedX(): minScore used as given, result[5]        */
    BBMSA_FILL_UNLIMITED_RAW = 1, /* == C fillUnlimited(): result[0..3]                             */
    BBMSA_FILL_LIMITED       = 2, /* == Java fillLimited(gaps==null): unlimited fallback gate +     */
                                  /*    minScore-=120 (MultiStateAligner11tsJNI.java:132-164)       */
    BBMSA_MODE_MASK          = 7,
    BBMSA_CLAMP_WINDOW = 1 << 3,  /* a=max(0,start), b=min(ref_len-1,end), MSA.java:104-105,118-121 */
    BBMSA_DO_SCORE     = 1 << 4,  /* run score2 on a non-null fill                                  */
    BBMSA_DO_TRACEBACK = 1 << 5,  /* run traceback2 on a non-null fill, write the match string      */
    BBMSA_NO_ITERATIONS = 1 << 6, /* the caller does not need result.iterations.  Accepted and without effect: until ABI 6 the     */
                                  /* library first tried such a job with a tighter minScore; fillLimitedX's pruning turned out not */
                                  /* to be admissible (a tighter bound can change a non-null result), so it never does now          */
    BBMSA_TRACE_KEEP_GAPS = 1 << 7, /* traceback: leave each gap symbol '-' of a gapped reference in the match string instead of expanding
                                    * it to 128 'D' (MultiStateAligner11tsJNI.java:481-493): the string then always fits rows + columns
                                    * bytes, and the caller expands it (a 16 kb deletion is 125 symbols, 16,000 'D') */
    BBMSA_INTERNAL_GAPPED = 1 << 8 /* set by the library on the jobs it derives for gapped references; never by callers */
};
/* the composite the mapper calls most: MSA.fillAndScoreLimited(read, ref, start, stop, minScore, null) */
#define BBMSA_FILL_AND_SCORE_LIMITED (BBMSA_FILL_LIMITED | BBMSA_CLAMP_WINDOW | BBMSA_DO_SCORE)

typedef struct bbmsa_job {
    int64_t read_off;     /* byte offset of read[0] inside the `reads` buffer                       */
    int64_t ref_off;      /* byte offset of ref[0] (the array refStartLoc/refEndLoc index into)     */
    int32_t read_len;     /* rows                                                                    */
    int32_t ref_len;      /* ref.length, used by BBMSA_CLAMP_WINDOW                                  */
    int32_t refStartLoc;
    int32_t refEndLoc;
    int32_t minScore;
    int32_t flags;
} bbmsa_job;              /* 40 bytes */

/* status values */
enum {
    BBMSA_ST_OK = 0,
    BBMSA_ST_NULL = 1,        /* the Java call would have returned null (below minScore)            */
    BBMSA_ST_BAD_SHAPE = 2    /* rows/columns outside the context limits: nothing was computed      */
};

typedef struct bbmsa_result {
    int32_t result[5];    /* {rows, maxCol, maxState, maxScore, belowMin} exactly as the C fills it  */
    int32_t status;
    int64_t iterations;   /* cells visited (what the C adds to iterationsLimited / ...Unlimited)     */
    int32_t score[8];     /* score2: {score,bestRefStart,bestRefStop,maxRow,maxCol,maxState,padL,padR} */
    int32_t score_len;    /* 0 (not run / null), 6 or 8                                              */
    int32_t match_len;    /* bytes of match string written (0 = none, -1 = slot too small)           */
    int32_t fill_kind;    /* 0 = limited fill ran, 1 = unlimited fill ran                            */
    int32_t columns;      /* columns actually aligned (after clamping)                               */
} bbmsa_result;           /* 80 bytes */

typedef struct bbmsa_ctx bbmsa_ctx;

typedef struct bbmsa_config {
    int32_t device;          /* HIP device ordinal                                                   */
    int32_t maxRows;         /* like MSA(maxRows_, maxColumns_), MSA.java:66-69; <= 640              */
    int32_t maxColumns;      /* <= 4096                                                              */
    int32_t bandwidth;       /* MSA.bandwidth (static in the reference, MSA.java:864)                */
    float   bandwidthRatio;  /* MSA.bandwidthRatio (MSA.java:865)                                    */
    int32_t reserved[3];     /* [0] lanes per job (0 = auto), [1] first-pass column buffer (0 = auto),
                              * [2] scoring scheme: BBMSA_SCHEME_11TS (0) or BBMSA_SCHEME_9PACBIO            */
} bbmsa_config;

/* Scoring schemes.  11ts: align2.MultiStateAligner11ts[JNI] (jni/MultiStateAligner11tsJNI.c:18-98), maxRows <= 640,
 * maxColumns <= 4096.  9PacBio: align2.MultiStateAligner9PacBio (current/align2/MultiStateAligner9PacBio.java:2359-2439:
 * 9 time bits, its own point values and barriers, column 0 as its constructor fills it), maxRows <= 6100,
 * maxColumns <= 8192: the strip-tiled wavefront kernel (banded fills: the one-job-per-thread kernel). */
#define BBMSA_SCHEME_11TS 0
#define BBMSA_SCHEME_9PACBIO 1
/* OR-ed into reserved[2]: a context for bbmsa_fill_submit / _collect / _packed only (what the per-call JNI symbols use, ONE per
 * process and MSA shape, shared by all mapping threads): persistent pinned + device staging for two batches of calls
 * (BBMSA_LEGACY_POOL_MB, default 64 MB each) and one scratch matrix, no batch buffers */
#define BBMSA_LEGACY_ONLY 0x100

int bbmsa_create(const bbmsa_config *cfg, bbmsa_ctx **out);
void bbmsa_destroy(bbmsa_ctx *ctx);

/* Device-resident batch: every pointer is a device pointer valid on cfg->device; the call only
 * enqueues work on `stream` (a hipStream_t passed as void*, NULL = default stream) and returns.
 * `match` receives one slot of `match_stride` bytes per job (may be NULL when no job asks for
 * BBMSA_DO_TRACEBACK). */
int bbmsa_align_batch_device(bbmsa_ctx *ctx, void *stream, int64_t n_jobs,
                             const bbmsa_job *jobs, const uint8_t *reads, const uint8_t *refs,
                             bbmsa_result *results, uint8_t *match, int32_t match_stride);

/* The same with the job count left on the device: the kernels read min(*n_jobs_dev, max_jobs) when they start (max_jobs =
 * capacity of jobs/results/match).  Lets a pipeline chain the site filter (which counts the jobs it writes) and the DP in
 * one stream without a host round trip. */
int bbmsa_align_batch_device_indirect(bbmsa_ctx *ctx, void *stream, const uint32_t *n_jobs_dev, int64_t max_jobs,
                                      const bbmsa_job *jobs, const uint8_t *reads, const uint8_t *refs,
                                      bbmsa_result *results, uint8_t *match, int32_t match_stride);

/* Host-buffer batch: copies in, runs, copies out, synchronises. */
int bbmsa_align_batch(bbmsa_ctx *ctx, int64_t n_jobs, const bbmsa_job *jobs,
                      const uint8_t *reads, int64_t reads_bytes,
                      const uint8_t *refs, int64_t refs_bytes,
                      bbmsa_result *results, uint8_t *match, int32_t match_stride);

/* Legacy per-call shape: ONE raw fill (mode = BBMSA_FILL_LIMITED_RAW or BBMSA_FILL_UNLIMITED_RAW) whose three score planes end up
 * in the caller's `packed` array exactly where the reference's native code leaves them
 * (state * (maxRows+1)*(maxColumns+1) + row * (maxColumns+1) + col; jni/MultiStateAligner11tsJNI.c:124-127, :707-812), so
 * that the unmodified Java score2 / traceback2 (current/align2/MultiStateAligner11tsJNI.java:376-658) can read them.  This
 * is what a drop-in Java_align2_MultiStateAligner11tsJNI_fill*JNI symbol calls.  The context must
 * have been created with BBMSA_LEGACY_ONLY; it may be shared by any number of threads, and should be: calls that arrive while
 * another is on the device are combined into one launch (one wavefront per fill), so throughput grows with the number of
 * calling threads.
 *   bbmsa_fill_submit   blocks until the fill has run and its planes sit in the context's pinned staging area; writes
 *                       result5 (result[0..4] of the native fill; [4] only for the limited fill) and INCREMENTS *iterations;
 *   bbmsa_fill_collect  copies the planes of that fill (rows 1..rows, columns 1..columns) into `packed`, and -- limited fill --
 *                       vertLimit[0..rows] / horizLimit[0..columns] as the native code leaves them (jni/...c:413-438); any of
 *                       the three may be NULL.  Pure memcpy: safe inside a JNI critical region.  Every successful submit must
 *                       be collected exactly once (the staging area is reused when its last fill has been collected);
 *   bbmsa_fill_packed   = submit + collect. */
typedef struct bbmsa_ticket {
    int32_t batch, slot;
    int64_t gen;             /* < 0: nothing to collect */
    int32_t rows, columns;
} bbmsa_ticket;
int bbmsa_fill_submit(bbmsa_ctx *ctx, const uint8_t *read, int32_t read_len, const uint8_t *ref, int32_t ref_len,
                      int32_t refStartLoc, int32_t refEndLoc, int32_t minScore, int32_t mode,
                      int32_t *result5, int64_t *iterations, bbmsa_ticket *ticket);
int bbmsa_fill_collect(bbmsa_ctx *ctx, bbmsa_ticket *ticket, int32_t *packed, int32_t *vertLimit, int32_t *horizLimit);
int bbmsa_fill_packed(bbmsa_ctx *ctx, const uint8_t *read, int32_t read_len, const uint8_t *ref, int32_t ref_len,
                      int32_t refStartLoc, int32_t refEndLoc, int32_t minScore, int32_t mode,
                      int32_t *result5, int64_t *iterations, int32_t *packed);
/* stats6: calls served, device batches launched, fills the wavefront kernel handed to the one-thread kernel; nanoseconds the
 * leaders spent in the wavefront pass (upload .. first synchronisation), in the hand-over pass, and waiting for collectors */
int bbmsa_legacy_stats(bbmsa_ctx *ctx, int64_t *stats6);

/* Gapped reference windows: job i is MSA.fillAndScoreLimited(read, ref, refStartLoc, refEndLoc, minScore, gaps)
 * (current/align2/MSA.java:103-134) for a SiteScore that carries a gap array.  When gaps[i].ngaps > 0 the library
 * builds the gapped reference (MultiStateAligner11tsJNI.makeGref, current/align2/MultiStateAligner11tsJNI.java:
 * 668-757: long gaps shrink to 64+rem bases, GAPC symbols, 64 bases), fills it as fillLimited(..., gaps) does
 * (:116-128), runs score(..., gapped=true) / traceback(..., gapped=true) (:362-372, :499-531) and translates
 * score[1], score[2] back to reference coordinates (:759-779).  Jobs with ngaps == 0 behave exactly as in
 * bbmsa_align_batch_device.  For gapped jobs the mode bits of job.flags select the Java fillLimited (any mode but
 * BBMSA_FILL_UNLIMITED_RAW) or fillUnlimited(read, ref, a, b, gaps) (:166-176; BBMSA_FILL_UNLIMITED_RAW), always with the
 * window clamp and score; BBMSA_DO_TRACEBACK is honoured.  The reference asserts gstart == 0 (:514); a job that
 * would break that, or whose gapped reference exceeds maxColumns + 2 bytes, gets BBMSA_ST_BAD_SHAPE. */
#define BBMSA_MAX_GAPS 16
typedef struct bbmsa_gaps { int32_t ngaps; int32_t gaps[BBMSA_MAX_GAPS]; } bbmsa_gaps;   /* 68 bytes */
int bbmsa_align_gapped_batch_device(bbmsa_ctx *ctx, void *stream, int64_t n_jobs, const bbmsa_job *jobs,
                                    const bbmsa_gaps *gaps, const uint8_t *reads, const uint8_t *refs,
                                    bbmsa_result *results, uint8_t *match, int32_t match_stride);
int bbmsa_align_gapped_batch_device_indirect(bbmsa_ctx *ctx, void *stream, const uint32_t *n_jobs_dev, int64_t max_jobs,
                                             const bbmsa_job *jobs, const bbmsa_gaps *gaps, const uint8_t *reads,
                                             const uint8_t *refs, bbmsa_result *results, uint8_t *match, int32_t match_stride);
int bbmsa_align_gapped_batch(bbmsa_ctx *ctx, int64_t n_jobs, const bbmsa_job *jobs, const bbmsa_gaps *gaps,
                             const uint8_t *reads, int64_t reads_bytes, const uint8_t *refs, int64_t refs_bytes,
                             bbmsa_result *results, uint8_t *match, int32_t match_stride);

/* Timing of the last bbmsa_align_batch_device launch sequence on this context, measured with
 * HIP events on the launch stream.  Valid after the stream has been synchronised. */
int bbmsa_last_kernel_ms(bbmsa_ctx *ctx, float *ms_fast, float *ms_slow);
/* The same per kernel: ms3 = {narrow-window kernel, wavefront kernel, generic kernel}. */
int bbmsa_last_kernel_ms3(bbmsa_ctx *ctx, float *ms3);
/* Which kernel took how many jobs of the last launch sequence: counts4 = {finished by the narrow-window kernel (one job
 * per lane, a band of diagonals in registers), candidates it handed on because their window left the band, jobs given
 * to the wavefront kernel in total, jobs the wavefront kernel handed to the generic kernel}. */
int bbmsa_last_counts(bbmsa_ctx *ctx, int64_t *counts4);

/* =====================================================================================
 * BandedAligner (unit-cost edit distance in a diagonal band)
 *   replaces jni/BandedAlignerJNI.c (alignForward :123-239, alignForwardRC :241-355,
 *   alignReverse :357-470, alignReverseRC :472-585; JNI glue :588-757, header
 *   jni/align2_BandedAlignerJNI.h:17-41).  The reference ships two semantics that disagree
 *   (its Java class current/align2/BandedAlignerConcrete.java:100-551 is the live one,
 *   BandedAligner.java:18-22); the context selects which one is reproduced.
 * ===================================================================================== */
enum {
    BBBAND_FORWARD = 0, BBBAND_FORWARD_RC = 1, BBBAND_REVERSE = 2, BBBAND_REVERSE_RC = 3,
    BBBAND_DIR_MASK = 3,
    BBBAND_EXACT = 1 << 2       /* the `exact` argument: undefined bases only match themselves */
};
enum {
    BBBAND_SEMANTICS_JNI_C = 0,       /* big=999, width=min(maxWidth,2*maxEdits+1), off-centre penalty +i        */
    BBBAND_SEMANTICS_JAVA = 1         /* big=99999999, width also capped by 2*max(len)+2 and |1, penalty max(i,x) */
};

typedef struct bbband_job {
    int64_t query_off;    /* byte offsets into the `seqs` buffer */
    int64_t ref_off;
    int32_t query_len;
    int32_t ref_len;
    int32_t qstart;
    int32_t rstart;
    int32_t maxEdits;
    int32_t flags;        /* direction | BBBAND_EXACT */
} bbband_job;             /* 40 bytes */

typedef struct bbband_result {
    int32_t edits;        /* the function's return value */
    int32_t lastQueryLoc; /* returnVals[0..4] of the JNI call, jni/BandedAlignerJNI.c:604-630 */
    int32_t lastRefLoc;
    int32_t lastRow;
    int32_t lastEdits;
    int32_t lastOffset;
    int32_t status;       /* 0 ok, 2 bad shape (index outside its sequence) */
    int32_t reserved;
} bbband_result;          /* 32 bytes */

typedef struct bbband_ctx bbband_ctx;
typedef struct bbband_config {
    int32_t device;
    int32_t width;        /* constructor argument of BandedAligner: maxWidth = max(width,3)|1, <= 1023 */
    int32_t semantics;    /* BBBAND_SEMANTICS_* */
    int32_t reserved;
} bbband_config;

int bbband_create(const bbband_config *cfg, bbband_ctx **out);
void bbband_destroy(bbband_ctx *ctx);
int bbband_align_batch_device(bbband_ctx *ctx, void *stream, int64_t n_jobs, const bbband_job *jobs,
                              const uint8_t *seqs, bbband_result *results);
int bbband_align_batch(bbband_ctx *ctx, int64_t n_jobs, const bbband_job *jobs,
                       const uint8_t *seqs, int64_t seq_bytes, bbband_result *results);

/* BandedAligner's orchestration over the four directions (current/align2/BandedAligner.java:24-55), batched over
 * (query, ref) pairs held in one host buffer `seqs`; edits[i] = the method's return value for pair i.
 *   alignQuadruple(query, ref, maxEdits, exact)                        :39-48
 *   alignQuadrupleProgressive(query, ref, minEdits, maxEdits, exact)   :24-37  (minEdits >= 1, as every caller passes)
 *   alignDouble(query, ref, maxEdits, exact)                           :50-55 */
typedef struct bbband_pair { int64_t query_off, ref_off; int32_t query_len, ref_len; } bbband_pair;   /* 24 bytes */
int bbband_align_quadruple_batch(bbband_ctx *ctx, int64_t n_pairs, const bbband_pair *pairs, const uint8_t *seqs, int64_t seq_bytes,
                                 int32_t maxEdits, int32_t exact, int32_t *edits);
int bbband_align_quadruple_progressive_batch(bbband_ctx *ctx, int64_t n_pairs, const bbband_pair *pairs, const uint8_t *seqs,
                                             int64_t seq_bytes, int32_t minEdits, int32_t maxEdits, int32_t exact, int32_t *edits);
int bbband_align_double_batch(bbband_ctx *ctx, int64_t n_pairs, const bbband_pair *pairs, const uint8_t *seqs, int64_t seq_bytes,
                              int32_t maxEdits, int32_t exact, int32_t *edits);

/* =====================================================================================
 * k-mer index probe (align2.BBIndex.findAdvanced)
 *   The reference has NO native boundary for the index: BBIndex is pure Java.  This is the new seam
 *   SURVEY.md section 0/R3 proposes: AbstractIndex.findAdvanced(basesP, basesM, qual, baseScoresP,
 *   keyScoresP, offsets, id) -> ArrayList<SiteScore>  (current/align2/AbstractIndex.java:83; sole call
 *   site current/align2/AbstractMapThread.java:736), batched over reads.  The index itself
 *   (Block.sites/starts, current/align2/Block.java:162-165; AbstractIndex.COUNTS; the chromosome byte
 *   arrays) is uploaded once and stays resident in HBM.
 * ===================================================================================== */
/* Which of the reference's two index / mapper class families a context follows.  BBMAP: align2.BBIndex + BBMapThread + BBMap.setDefaults
 * + MultiStateAligner11ts (bbmap.sh).  PACBIO: align2.BBIndexPacBio + BBMapThreadPacBio + BBMapPacBio.setDefaults +
 * MultiStateAligner9PacBio (mapPacBio.sh).  BBIndexPacBio is BBIndex with other constants (current/align2/BBIndexPacBio.java:2461-2596:
 * k 12, MAX_INDEL 100 / MAX_INDEL2 800, Z_SCORE_MULT 25, INDEL_PENALTY KEY/8-1 and x25, HIT_FRACTION_TO_RETAIN .97, MIN_HIT_LISTS_TO_RETAIN
 * 12, SMALL_GENOME_LIST 80, MIN_SCORE_MULT .02, MIN_QSCORE_MULT(2) .005, DYNAMIC_SCORE_THRESH .64, MAX_HITS_REDUCTION_PERFECT 2,
 * clumpy-key constants 2800 / .8, FRACTION_GENOME_TO_EXCLUDE .005, the retry thresholds of find() 20/18/16/14 (:409-421)), reads of
 * up to 6000 bases and up to 2047 keys (:2394-2396), and its location-array score is MultiStateAligner9PacBio.calcAffineScore. */
enum { BBIDX_PROFILE_BBMAP = 0, BBIDX_PROFILE_PACBIO = 1 };

typedef struct bbidx_params {   /* the reference's mutable statics, BBIndex.java:3168-3305, AbstractIndex.java:100-160 */
    int32_t k, chromBits, minChrom, maxChrom;
    int32_t maxIndel, maxIndel2, minApproxHitsToKeep, kfilter;
    int32_t maxUsableLength, maxUsableLength2;
    int32_t maxHitsReduction2, maximumMaxHitsReduction, hitReductionDiv;
    int32_t quitAfterTwoPerfects, prescanQscore, trimByGreedy, slow;
    int32_t maxAverageListToSearch, maxAverageListToSearch2, maxShortestListToSearch;
    int64_t pointsPerSite;      /* Solver.POINTS_PER_SITE after analyzeIndex */
    int32_t profile;            /* BBIDX_PROFILE_*: the constants that are `static final` in the reference */
    int32_t reserved;
} bbidx_params;                 /* 96 bytes */

typedef struct bbidx_index_desc {   /* host pointers; everything is copied to the device by bbidx_create */
    bbidx_params params;
    int32_t nblocks;                /* blocks are index[baseChrom]; block b holds chromosomes b<<chromBits .. */
    int32_t nchroms;                /* chromosomes are numbered 1..nchroms */
    const int32_t *const *starts;   /* per block: 4^k + 1 ints */
    const int32_t *const *sites;    /* per block */
    const int64_t *numSites;        /* per block */
    const int32_t *counts;          /* AbstractIndex.COUNTS, 4^k ints */
    const int32_t *lengthHistogram; /* 1001 ints */
    const uint8_t *const *chromArr; /* [nchroms+1], entry 0 unused */
    const int32_t *chromArrLen;     /* array length of each chromosome */
    const int32_t *chromLengths;    /* Data.chromLengths[chrom] */
} bbidx_index_desc;

typedef struct bbidx_read {
    int64_t bases_off;      /* offset of basesP in `bases`; baseScoresP sits at the same offset in `baseScores` */
    int64_t keys_off;       /* offset (in ints) into `keyinfo`: offsets[nkeys] followed by keyScoresP[nkeys]   */
    int32_t len;
    int32_t nkeys;
} bbidx_read;               /* 24 bytes */

#define BBIDX_MAX_GAPS 16
typedef struct bbidx_site { /* stream.SiteScore as the probe emits it (current/stream/SiteScore.java:999-1011) */
    int32_t chrom, strand, start, stop, hits, score, perfect, semiperfect;
    int32_t ngaps;
    int32_t gaps[BBIDX_MAX_GAPS];
} bbidx_site;               /* 100 bytes */

enum { BBIDX_MAX_KEYS = 128, BBIDX_MAX_READ_LEN = 600 };          /* BBIDX_PROFILE_BBMAP (BBIndex keeps 256 keys, 600 bases) */
enum { BBIDX_PACBIO_MAX_KEYS = 2047, BBIDX_PACBIO_MAX_READ_LEN = 6016 };   /* BBIDX_PROFILE_PACBIO: HEAP_LENGTH 2047; reads are split at 6000 */

typedef struct bbidx_ctx bbidx_ctx;
int bbidx_create(int32_t device, const bbidx_index_desc *desc, bbidx_ctx **out);
void bbidx_destroy(bbidx_ctx *ctx);
/* Device-resident batch.  sites: n_reads x max_sites records; nsites[i] = sites found for read i, or -1 when
 * max_sites was too small, -2 when the read exceeds BBIDX_MAX_KEYS / BBIDX_MAX_READ_LEN. */
int bbidx_find_batch_device(bbidx_ctx *ctx, void *stream, int64_t n_reads, const bbidx_read *reads,
                            const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo,
                            bbidx_site *sites, int32_t max_sites, int32_t *nsites);
/* The same, and every probed read's reverse complement (AminoAcid.reverseComplementBases, which the mapper computes once
 * per read as basesM, current/align2/AbstractMapThread.java:643-655) is written to bases_rc_out at the read's offset: the
 * kernel has it in LDS anyway, which saves the separate bbpipe_revcomp_device pass.  Reads without a usable key (nsites 0
 * because len < k or no keys, or -2) are not written. */
int bbidx_find_batch_device_rc(bbidx_ctx *ctx, void *stream, int64_t n_reads, const bbidx_read *reads,
                               const uint8_t *bases, const int8_t *baseScores, const int32_t *keyinfo,
                               bbidx_site *sites, int32_t max_sites, int32_t *nsites, uint8_t *bases_rc_out);
int bbidx_find_batch(bbidx_ctx *ctx, int64_t n_reads, const bbidx_read *reads,
                     const uint8_t *bases, const int8_t *baseScores, int64_t bases_bytes,
                     const int32_t *keyinfo, int64_t keyinfo_ints,
                     bbidx_site *sites, int32_t max_sites, int32_t *nsites);

/* Work counters and HIP-event duration of the last bbidx_find_batch_device launch; valid once its
 * stream has been synchronised.  stats5 = {list entries consumed by the prescan, by the walk, extendScore calls,
 * reference bytes compared, site records written}. */
int bbidx_last_stats(bbidx_ctx *ctx, int64_t *stats5, float *kernel_ms);

/* Index construction on the device: align2.IndexMaker4 (current/align2/IndexMaker4.java:303-421: count -> prefix sum ->
 * fill, lists in genome order) + BBIndex.analyzeIndex (current/align2/BBIndex.java:101-191: COUNTS, clumpy keys, length
 * histogram, MAX_USABLE_LENGTH) + the genome-size tuning of BBMap.loadIndex (current/align2/BBMap.java:367-381).
 * chromArr[1..nchroms] are host pointers to the chromosome byte arrays (entry 0 unused); chromBits < 0 = automatic
 * (BBMap.java:317-321).  The result is a ready-to-probe context; nothing but a few thousand histogram counters visits the
 * host. */
int bbidx_build(int32_t device, int32_t k, int32_t chromBits, int32_t nchroms,
                const uint8_t *const *chromArr, const int32_t *chromArrLen, bbidx_ctx **out);
/* The same for either profile (bbidx_build = BBIDX_PROFILE_BBMAP); k <= 0 takes the profile's default (13 / 12). */
int bbidx_build_profile(int32_t device, int32_t profile, int32_t k, int32_t chromBits, int32_t nchroms,
                        const uint8_t *const *chromArr, const int32_t *chromArrLen, bbidx_ctx **out);
/* The tunables a context works with (derived by bbidx_build, or as given to bbidx_create). */
int bbidx_get_params(bbidx_ctx *ctx, bbidx_params *out);
/* Copies one block's arrays back to the host (any pointer may be NULL): starts 4^k+1 ints, sites starts[4^k] ints
 * (sites_cap = capacity of the buffer), counts 4^k ints, lengthHistogram 1001 ints. */
int bbidx_export_block(bbidx_ctx *ctx, int32_t block, int32_t *starts, int32_t *sites, int64_t sites_cap,
                       int32_t *counts, int32_t *lengthHistogram);

/* Which probe kernel a context launches.  AUTO (default): one read per wavefront (registers + LDS), with the
 * one-read-per-lane kernel taking the reads that do not fit it (more than 64 keys).  LANE: the per-lane kernel
 * for every read (any shape up to BBIDX_MAX_KEYS / BBIDX_MAX_READ_LEN; kept as the cross-check). */
enum { BBIDX_KERNEL_AUTO = 0, BBIDX_KERNEL_LANE = 1,
       BBIDX_KERNEL_LONG = 2 };   /* the long-read kernel (up to 6016 bases, 2047 keys) for every read: what a BBIDX_PROFILE_PACBIO
                                   * context always runs; selectable on a BBMAP context as a third cross-check */
int bbidx_set_kernel(bbidx_ctx *ctx, int32_t kind);

/* Longest read of the batches to come (default BBIDX_MAX_READ_LEN).  Sizing hint, not a limit: the wavefront kernel keeps a
 * read's per-base arrays in LDS, and with reads of at most 160 bases it needs a third less of it and runs 8 waves per SIMD
 * instead of 6; reads longer than announced are still answered (by the per-lane kernel). */
int bbidx_set_max_read_len(bbidx_ctx *ctx, int32_t max_len);

/* =====================================================================================
 * The probe's per-read inputs, host side: AbstractMapThread.quickMap up to its findAdvanced call
 *   (current/align2/AbstractMapThread.java:642-728): key error probabilities from the qualities (QualityTools.makeKeyProbs,
 *   current/align2/QualityTools.java:188-279), key placement (KeyRing.makeOffsets3, current/align2/KeyRing.java:396-506, with the
 *   density window of :663-676), key scores (QualityTools.makeKeyScores :125-133) and base scores (makeByteScoreArray :145-181).
 *   Float code on the mapping thread in the reference, host code here; its integer outputs are the probe's inputs.
 * ===================================================================================== */
typedef struct bbkeys_config {
    int32_t k;                      /* KEYLEN */
    float keyDensity, maxKeyDensity, minKeyDensity;   /* BBMap.java:52-54 (1.9 / 3 / 1.5); BBMapPacBio.java:55-57 (3.5 / 4.5 / 2.8) */
    int32_t maxDesiredKeys;         /* 15 / 63 */
    int32_t minApproxHitsToKeep;    /* AbstractIndex.MIN_APPROX_HITS_TO_KEEP, 1 */
    int32_t semiperfectMode;        /* PERFECTMODE || SEMIPERFECTMODE */
    int32_t reserved;
} bbkeys_config;
int bbkeys_default_config(int32_t profile, bbkeys_config *cfg);
/* One read.  quality: numeric phred values (not ASCII), or NULL for a read without qualities.  Writes baseScores[len]; returns the
 * number of keys written to offsets[] / keyScores[] (cap entries each), 0 when quickMap would return without probing (read shorter
 * than k, mostly undefined, no usable key, all keys probably wrong), < 0 on a bad argument. */
int bbkeys_make(const bbkeys_config *cfg, const uint8_t *bases, const uint8_t *quality, int32_t len,
                int32_t *offsets, int32_t *keyScores, int32_t cap, int8_t *baseScores);
/* A batch: read i occupies bases[bases_off[i] .. + lens[i]) (qualities at the same offsets, or quality == NULL).  Fills reads[i],
 * keyinfo (offsets[nkeys] then keyScores[nkeys] per read, keys_off in ints) and baseScores; *keyinfo_used = ints written. */
int bbkeys_make_batch(const bbkeys_config *cfg, int64_t n_reads, const int64_t *bases_off, const int32_t *lens,
                      const uint8_t *bases, const uint8_t *quality, bbidx_read *reads, int32_t *keyinfo, int64_t keyinfo_cap,
                      int8_t *baseScores, int64_t *keyinfo_used);

/* =====================================================================================
 * Batch helpers (device-resident) used by the mapper below; also callable on their own.
 * ===================================================================================== */
/* bases_out[read] = reverse complement of bases_in[read] (AminoAcid.reverseComplementBases) */
int bbpipe_revcomp_device(void *stream, int64_t n_reads, const bbidx_read *reads,
                          const uint8_t *bases_in, uint8_t *bases_out);

/* Paired-read rescue scan: AbstractMapThread.quickRescue(bases, chrom, strand, loc, searchDist, searchRight, idealStart,
 * maxAllowedMismatches, POINTS_MATCH, POINTS_MATCH2) (current/align2/AbstractMapThread.java:2300-2391), batched.  `reads`
 * holds the mate's bases already on the strand to search; chromosome c occupies refs[chrom_off[c] .. + chrom_len[c]) and
 * chrom_min_index[c] is ChromosomeArray.minIndex.  result.found: 1 = a SiteScore (start, stop, score; mismatches is what
 * the reference parks in ss.slowScore; perfect/semiperfect from SiteScore.setPerfect), 0 = null, -2 = read longer than
 * 600.  use_affine selects the reference's USE_AFFINE_SCORE score formula (:2376-2380). */
typedef struct bbresc_job {
    int64_t read_off;
    int32_t read_len, chrom, loc, searchDist, idealStart, maxAllowedMismatches;
    int32_t flags;            /* bit 0: searchRight */
    int32_t reserved;
} bbresc_job;                 /* 40 bytes */
typedef struct bbresc_result { int32_t found, start, stop, score, mismatches, perfect, semiperfect, maxContig; } bbresc_result;   /* 32 bytes */
int bbpipe_quick_rescue_device(void *stream, int64_t n_jobs, const bbresc_job *jobs, const uint8_t *reads,
                               const int64_t *chrom_off, const int32_t *chrom_len, const int32_t *chrom_min_index,
                               const uint8_t *refs, bbresc_result *results,
                               int32_t points_match, int32_t points_match2, int32_t use_affine, int32_t base_hit_score);


/* =====================================================================================
 * Mapper control flow around the two hot kernels, device-resident.
 *   The part of align2.BBMapThread.processRead / processReadPair (current/align2/BBMapThread.java:389-490, :943-1098)
 *   that decides WHICH probe sites are aligned, with which window and minScore, in which order, and which rescue searches
 *   run, kept on the device so that a read batch stays in HBM from the probe to the last alignment:
 *     quickMap tail      AbstractMapThread.java:736-751 (findAdvanced; removeOutOfBounds :2444-2476)
 *     pairing + trimming BBMapThread.java:736-940 pairSiteScoresInitial, :140-249 trimList (Tools.java:654-674, :1113-1161)
 *     scoreNoIndels      AbstractMapThread.java:762-856;  findTipDeletions :1075-1141, :2178-2292
 *     scoreSlow          BBMapThread.java:252-386 -- the exact per-read SEQUENCE: every site's minScore follows the previous
 *                        sites' results (minMsaLimit, :376) and a fill that asks for more padding is repeated wider (:312-335),
 *                        so the DP runs in rounds (round j = the j-th fill of every read that still has one)
 *     mergeDuplicateSites Tools.java:697-759
 *     rescue / slowRescue AbstractMapThread.java:1144-1306 (quickRescue :2303-2404), mate 1 as anchor, then mate 2
 *   Configuration: bbmap.sh defaults (BBMap.setDefaults, BBMap.java:45-65), reads without qualities.
 *     the final stage    BBMapThread.java:492-732 / :1116-1356 (cfg.finalStage): final pairing (pairSiteScoresFinal), the ambiguity
 *                        policy, genMatchString -> genMatchStringForSite -> realign_new (AbstractMapThread.java:860-1068,
 *                        TranslateColorspaceRead.java:229-653: up to three fillLimited and one fillUnlimited per call, again in rounds),
 *                        fixXY / clipTipIndels / toLocalAlignment, applyClearzone3 and the tip penalty -> bbmap_final per read
 *   Not carried over: per-thread adaptive state of the Java mapper (DYNAMIC_INSERT_LENGTH: the caller feeds averagePairDist, see
 *   bbmap_set_average_pair_dist; the "mating is not working" skip of rescue()), scaffold boundaries inside a chromosome, the
 *   non-default output policies (ambiguous=toss/random/all, secondary alignments, identity / edit filters, local alignment).
 *   Added product: every successful fill also returns its traceback string (as the quickmatch=t branch obtains it,
 *   BBMapThread.java:345, without fixXY / clipTipIndels); site state follows the default (quickmatch=f) flow.
 * ===================================================================================== */
typedef struct bbmap_msite {       /* stream.SiteScore, current/stream/SiteScore.java:999-1011 */
    int32_t chrom, strand, start, stop, hits;
    int32_t quickScore, score, slowScore, pairedScore;
    int32_t perfect, semiperfect, rescued;
    int32_t ngaps;                 /* 0 = gaps == null */
    int32_t gaps[BBMSA_MAX_GAPS];
    int32_t match_job;             /* fill whose result (limits, traceback string) this site carries: index into the job log,
                                    * bit 30 set = the gapped log; -1 = none */
    int32_t reserved[2];
} bbmap_msite;                     /* 128 bytes */

#define BBMAP_NSITES_OVERFLOW (-1)
#define BBMAP_NSITES_MATE_OVERFLOW (-2)
#define BBMAP_NSITES_IN_TIER (-3)
#define BBMAP_MAX_SITES_LIMIT 4096

typedef struct bbmap_jobinfo {     /* one entry per fill, parallel to the job / result arrays */
    int32_t read;                  /* read the fill belongs to */
    int32_t seq;                   /* its position in that read's sequence of fillAndScoreLimited calls; -1 = a fill issued ahead
                                    * of time that the sequence turned out not to contain (ignore it) */
    int32_t kind;                  /* 0 scoreSlow fill, 1 scoreSlow wider refill, 2 slowRescue; final stage (realign_new,
                                    * current/align2/TranslateColorspaceRead.java:370, :413, :450, :457): 3 first fill, 4 padded refill,
                                    * 5 third fill, 6 fillUnlimited */
    int32_t site;                  /* list position of the site when the fill was issued */
} bbmap_jobinfo;                   /* 16 bytes */

typedef struct bbmap_config {
    int32_t device;
    int32_t paired;                /* 0: processRead per read; 1: processReadPair, reads 2p and 2p+1 are mates */
    int32_t max_reads;             /* capacity in reads (not pairs) */
    int32_t max_read_len;          /* <= 600 (BBIDX_PROFILE_BBMAP), <= 6016 (BBIDX_PROFILE_PACBIO: maxReadLength() = ALIGN_ROWS - 1 = 6019,
                                    * current/align2/BBMapThreadPacBio.java:28,70; pieces are cut at 6000) */
    int32_t max_sites;             /* per-read capacity of the probe output and of the mapper's site list (<= 4096); reads that
                                    * need more go to the overflow tier */
    float   minRatio;              /* MINIMUM_ALIGNMENT_SCORE_RATIO (0.56) */
    int32_t slowAlignPadding;      /* 4 */
    int32_t slowRescuePadding;     /* 8 */
    int32_t extraPadding;          /* 10 */
    int32_t tipSearchDist;         /* TIP_DELETION_SEARCH_RANGE, 100; 0 switches findTipDeletions off */
    int32_t maxPairDist;           /* 32000 */
    int32_t averagePairDist;       /* INITIAL_AVERAGE_PAIR_DIST, 100 */
    int32_t maxRescueDist;         /* 1200 */
    int32_t maxRescueMismatches;   /* 32 */
    int32_t maxTrimSitesToRetain;  /* 800 */
    int32_t trimList;              /* 1 */
    int32_t doRescue;              /* 1 */
    int32_t alignColumns;          /* BBIndex.ALIGN_COLUMNS, 3000 */
    int32_t clearzone3;            /* PENALIZE_AMBIG ? 800 : 0 */
    int32_t msaMaxColumns;         /* columns of the MSA instance (3000 in the reference): limit of the second DP context */
    int32_t fastCols;              /* column limit of the first DP context, which takes the ordinary windows (0 = 256); wider
                                    * windows and gapped references go to the second context (the "gapped" log) */
    int32_t jobsPerRead;           /* STARTING capacity of the job log = jobsPerRead * max_reads (0 = 3); the logs grow on demand.
                                    * < 0: exactly -jobsPerRead entries in each log to start with (tests of the growth path) */
    int32_t finalStage;            /* 1 (default, BBIDX_PROFILE_BBMAP): the whole of processRead / processReadPair -- after rescue the final
                                    * pairing, the ambiguity policy, genMatchString -> genMatchStringForSite -> realign_new (the fills that
                                    * produce the printed start / stop / score and the match string), clipping and the score penalties;
                                    * 0: stop after the rescue stage (the site lists as scoreSlow and rescue leave them) */
    int32_t reserved[4];           /* [0] != 0: strictly one fill per read and round (no fills ahead of time; for tests)
                                    * [1] overflow tier: reads it can hold per batch (0 = 4096, < 0 = no tier)
                                    * [2] overflow tier: its max_sites (0 = 1024)
                                    * [3] BBIDX_PROFILE_*: which mapper / aligner classes are followed (must equal the index's) */
} bbmap_config;

/* What BBMap prints for a read: stream.Read's mapping fields when processRead / processReadPair return (current/stream/Read.java;
 * set by genMatchString, current/align2/AbstractMapThread.java:946-959, and the policy behind it).  Filled when cfg.finalStage. */
typedef struct bbmap_final {
    int32_t mapped;                /* Read.mapped() */
    int32_t chrom, strand, start, stop;   /* -1, 0, -1, -1 when not mapped */
    int32_t mapScore;
    int32_t paired, ambiguous, perfect, rescued;
    int32_t match_len;             /* length of Read.match (long format: m S N D I X Y C), 0 = null */
    int32_t nsites;                /* sites left in the read's list (the list itself: bbmap_output.sites), or the overflow flags */
    int64_t match_off;             /* byte offset of the string in bbmap_output.final_match (valid when match_len > 0) */
    int32_t reserved[2];
} bbmap_final;                     /* 64 bytes */

typedef struct bbmap_output {      /* device pointers, valid until the next bbmap_map_batch_device / bbmap_destroy */
    const bbmap_msite *sites;      /* n_reads x cap */
    const int32_t *nsites;         /* per read: sites in its list, or BBMAP_NSITES_OVERFLOW (-1: the list did not fit max_sites and
                                    * the overflow tier could not take the read either: not mapped), BBMAP_NSITES_MATE_OVERFLOW
                                    * (-2: its mate's list did not fit), BBMAP_NSITES_IN_TIER (-3: mapped by the overflow tier, see
                                    * bbmap_get_overflow_output) */
    int32_t cap;
    int32_t match_stride, gmatch_stride;
    int32_t reserved;
    int64_t n_jobs, n_gapped_jobs; /* fills in the two logs */
    const bbmsa_job *jobs;   const bbmsa_result *results;  const bbmap_jobinfo *jobinfo;  const uint8_t *match;
    const bbmsa_job *gjobs;  const bbmsa_result *gresults; const bbmap_jobinfo *gjobinfo; const uint8_t *gmatch;
    const bbmsa_gaps *ggaps;
    /* the final alignment stage (cfg.finalStage): one record per read and the pool its match strings live in.  A site that was given a
     * match string of its own refers to it in the same pool: sites[].reserved[0] = byte offset / 4 + 1 (0 = none),
     * reserved[1] & 0xffff = its length.  NULL / 0 when the stage is off. */
    const bbmap_final *final;
    const uint8_t *final_match;
    int64_t final_match_bytes;     /* bytes of the pool in use */
    int64_t n_final_fills;         /* fills the stage issued (they are in the two logs, kinds 3..6) */
} bbmap_output;

typedef struct bbmap_stats {
    /* reads_overflowed: reads whose list fitted neither max_sites nor the overflow tier (left unmapped, flagged) */
    int64_t reads, reads_overflowed, reads_without_site, fills, gapped_fills, refills, rescue_scans, rescue_fills, rounds;
    int64_t fills_dropped;         /* fills issued ahead of time that the reference's sequence turned out not to contain (dropped;
                                    * their log entries keep seq = -1) */
    float ms_probe, ms_begin, ms_score, ms_slow, ms_finish, ms_rescue, ms_total;
    float ms_dp_narrow, ms_dp_wave, ms_dp_generic, ms_dp_gapped, ms_quick_rescue;
    int64_t probe_stats[5];        /* bbidx_last_stats of the probe launch */
    int64_t reads_reprobed;        /* reads the overflow tier mapped (pairs count both mates); fills etc. above include the tier's */
    float ms_overflow;             /* the overflow tier's whole pass (included in ms_total) */
    float log_growths;             /* times a fill log had to grow during the batch (the logs start at jobsPerRead entries per read) */
    float ms_dp_wave_max;          /* the longest single wavefront-kernel pass of the plain DP context in the batch (round 1's, as a rule);
                                    * ms_dp_wave is the sum over all rounds and rescue passes */
    float ms_final;                /* the final alignment stage (included in ms_total) */
    int64_t final_fills, final_rounds, final_local;   /* its fills, rounds, reads that went through toLocalAlignment */
} bbmap_stats;

typedef struct bbmap_ctx bbmap_ctx;
int bbmap_default_config(bbmap_config *cfg);
/* bbmap.sh's defaults (BBMap.setDefaults, current/align2/BBMap.java:45-65) or mapPacBio.sh's (BBMapPacBio.setDefaults,
 * current/align2/BBMapPacBio.java:47-69: minRatio 0.46, padding 8 / 16, tip search 15, 7600 columns, single-ended) */
int bbmap_default_config_profile(int32_t profile, bbmap_config *cfg);
/* The context borrows `index` (which must outlive it) and owns two DP contexts (plain and gapped-reference), the overflow tier
 * (a second, small set of the same) and every intermediate buffer.  One batch at a time per context, from one thread at a time; the
 * call itself uses two internal streams and a helper thread (the overflow tier's pass runs beside the main one) and has joined
 * them when it returns.  Two contexts over the same index must not map at the same time (the probe keeps its queue and counters
 * in the index context). */
/* Side effect on the borrowed index context: its quitAfterTwoPerfects tunable is set to !cfg->paired, as BBMap does with the index
 * class's static (`if(paired){BBIndex.QUIT_AFTER_TWO_PERFECTS=false;}`, current/align2/BBMap.java:434). */
int bbmap_create(bbidx_ctx *index, const bbmap_config *cfg, bbmap_ctx **out);
void bbmap_destroy(bbmap_ctx *ctx);
/* AVERAGE_PAIR_DIST for the batches to come.  The reference's mapping threads move it as they see pairs (DYNAMIC_INSERT_LENGTH:
 * `if(numMated>1000 && r.paired()){AVERAGE_PAIR_DIST=(int)(innerLengthSum*1f/numMated);}`, current/align2/BBMapThread.java:1307-1309,
 * per thread); a host that carries that running value sets it here between batches (cfg.averagePairDist is only the initial value,
 * INITIAL_AVERAGE_PAIR_DIST).  It enters pairSiteScoresInitial / Final and the rescue search (:1086, :1093). */
int bbmap_set_average_pair_dist(bbmap_ctx *ctx, int32_t average_pair_dist);
/* Maps a batch that is resident on the device.  reads[i].bases_off addresses the plus strand inside `bases`; the call writes
 * every read's reverse complement at bases_off + minus_delta.  Enqueues on `stream` and waits for it: the call returns when the
 * batch is done (the rounds of scoreSlow need the job counts on the host). */
int bbmap_map_batch_device(bbmap_ctx *ctx, void *stream, int64_t n_reads, const bbidx_read *reads, uint8_t *bases,
                           int64_t minus_delta, const int8_t *baseScores, const int32_t *keyinfo);
int bbmap_get_output(bbmap_ctx *ctx, bbmap_output *out);
/* The final alignment stage ALONE over site lists the caller provides (a host that runs its own pairing / rescue policy, or a test
 * that wants lists the mapper would rarely produce): sites = n_reads x cfg.max_sites records, nsites[r] = sites of read r, as
 * bbmap_map_batch_device leaves them after rescue with finalStage = 0 (sorted as processRead / processReadPair leave them; scores
 * set).  `bases` holds the plus strands and, at + minus_delta, the reverse complements (the caller's: bbpipe_revcomp_device writes
 * them).  Results as after bbmap_map_batch_device: bbmap_get_output (the two fill logs hold this call's fills only, numbered from
 * 0 per read; sites / nsites are the lists after the stage), bbmap_get_final.  No overflow tier is involved. */
int bbmap_final_batch_device(bbmap_ctx *ctx, void *stream, int64_t n_reads, const bbidx_read *reads, uint8_t *bases, int64_t minus_delta,
                             const bbmap_msite *sites, const int32_t *nsites);
/* Host-buffer form, for a host that owns no device memory (the JNI glue, jni/hip_glue.c: Java cannot allocate HBM).  bases holds the
 * plus strands only (bases_bytes bytes, baseScores the same length; both required).  The call uploads the batch into device
 * buffers the context keeps, maps it as bbmap_map_batch_device does and returns the site lists without their empty slots:
 * read r has nsites_out[r] sites at sites_out[offsets_out[r] ...] -- or nsites_out[r] = BBMAP_NSITES_OVERFLOW /
 * BBMAP_NSITES_MATE_OVERFLOW when even the overflow tier could not hold its list.  offsets_out has n_reads + 1 entries.  Lists
 * the overflow tier produced are appended behind the others.  *total_out = records produced; records beyond sites_cap are not
 * written (call again with a larger array).  The fill logs stay on the device (bbmap_get_output). */
int bbmap_map_batch(bbmap_ctx *ctx, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases, int64_t bases_bytes,
                    const int8_t *baseScores, const int32_t *keyinfo, int64_t keyinfo_ints, int32_t *nsites_out,
                    int64_t *offsets_out, bbmap_msite *sites_out, int64_t sites_cap, int64_t *total_out);
/* The overflow tier's results for the last batch: n_reads = 0 when no read needed it.  Tier read i is read read_ids[i] of the
 * batch (ascending; pairs keep their two mates adjacent); `out` is laid out like the main output with the tier's own cap, and its
 * job logs number the reads 0..n_reads-1 in tier order. */
typedef struct bbmap_overflow_output {
    int64_t n_reads;
    const int32_t *read_ids;       /* device pointer */
    bbmap_output out;
} bbmap_overflow_output;
int bbmap_get_overflow_output(bbmap_ctx *ctx, bbmap_overflow_output *out);
/* The last batch's final records on the host, overflow tier included (a read the tier mapped gets the tier's record): out[n_reads];
 * the match strings are packed into match_out in read order and out[r].match_off is rewritten to the string's offset THERE.
 * *match_bytes = bytes the strings take; strings that do not fit match_cap are not written (call again with a larger buffer).
 * match_out may be NULL (records only).  BBMAP_E_ARG when the context runs without the final stage. */
int bbmap_get_final(bbmap_ctx *ctx, int64_t n_reads, bbmap_final *out, uint8_t *match_out, int64_t match_cap, int64_t *match_bytes);
int bbmap_last_stats(bbmap_ctx *ctx, bbmap_stats *out);
/* The last batch's site lists without their empty slots, for a host that copies them back: counts (n_reads + 1 ints), offsets
 * (n_reads + 1 int64: exclusive prefix sums, offsets[n_reads] = total) and packed (packed_cap records) are device buffers of the
 * caller's; read r's counts[r] sites are packed[offsets[r] ...] (0 for a read without a list, a flagged one, or one the overflow
 * tier mapped).  Enqueues on `stream`; records beyond packed_cap are not written (compare offsets[n_reads] with packed_cap). */
int bbmap_pack_sites_device(bbmap_ctx *ctx, void *stream, int64_t n_reads, int32_t *counts, int64_t *offsets, bbmap_msite *packed,
                            int64_t packed_cap);
/* synchronous device-to-host copy of (part of) an output array */
int bbmap_copy_to_host(void *dst, const void *src_device, int64_t bytes);

/* device tables of an index context: chromArr[c] (device pointer to chromosome c's bytes), chromArrLen[c]; host copies */
int bbidx_get_chrom_table(bbidx_ctx *ctx, int32_t *nchroms, const uint8_t **chromArr_host_copy, int32_t *chromArrLen_host_copy, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
