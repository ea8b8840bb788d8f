// Host side of the MultiStateAligner11ts C ABI (include/bbmap_amd.h): context, scratch sizing,
// launch geometry and the two launches (wavefront kernel, then the generic kernel over the jobs
// it handed back).  No CPU compute path exists here: without a HIP device every entry fails.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "msa_common.h"

int bbmsa_align_impl(bbmsa_ctx *c, void *stream_, int64_t n_jobs, const uint32_t *n_jobs_dev, const bbmsa_job *jobs,
                     const uint8_t *reads, const uint8_t *refs, bbmsa_result *results, uint8_t *match, int32_t match_stride);

namespace bbmsa {
struct StripParams {            // msa_fill_strip.hip
    const bbmsa_job *jobs; const uint8_t *reads; const uint8_t *refs; bbmsa_result *results; uint8_t *match;
    long long njobs; const unsigned int *njobs_dev; unsigned int *queue; unsigned int *dirbuf;
    long long dir_slot_dwords, dir_strip_dwords; int *boundary; uint8_t *tmpbuf; int *slow_list; unsigned int *slow_count;
    int match_stride; int maxRows, maxColumns; int bandwidth; float bandwidthRatio;
    int pipeK, pipeSlots; int *pipeBoundary; int *pipeSync; int pipeSpinLimit;
};
int strip_rows_per_lane();
const void *strip_kernel_pacbio();
const void *strip_kernel_pacbio_pipelined();
int strip_pipe_sync_ints(int K);
const void *fast_kernel_for(int R, bool banded);
template <class S> __global__ void msa_fill_generic_kernel(const GenericParams p);
__global__ void msa_fill_narrow_kernel(const NarrowParams p);
}  // namespace bbmsa

namespace bbmsa {
constexpr int WIDTH_BUCKETS = 1024;            // columns / 8, clamped
__device__ inline int job_width_bucket(const bbmsa_job &t, int maxColumns) {
    int a = t.refStartLoc, b = t.refEndLoc;
    if (t.flags & BBMSA_CLAMP_WINDOW) { a = max(0, a); b = min(t.ref_len - 1, b); if (b - a >= maxColumns) b = min(t.ref_len - 1, a + maxColumns - 1); }
    const int cols = max(0, b - a + 1);
    return WIDTH_BUCKETS - 1 - min(WIDTH_BUCKETS - 1, cols >> 3);            // bucket 0 = the widest
}
__global__ void width_hist_kernel(const bbmsa_job *jobs, long long n, int maxColumns, unsigned *hist) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&hist[job_width_bucket(jobs[i], maxColumns)], 1u);
}
__global__ void width_scan_kernel(unsigned *hist, unsigned *listCount, unsigned n) {      // one block of WIDTH_BUCKETS threads: exclusive prefix sums in place
    __shared__ unsigned s[WIDTH_BUCKETS];
    const int t = threadIdx.x;
    s[t] = hist[t];
    __syncthreads();
    for (int d = 1; d < WIDTH_BUCKETS; d <<= 1) { const unsigned v = t >= d ? s[t - d] : 0u; __syncthreads(); s[t] += v; __syncthreads(); }
    hist[t] = s[t] - hist[t];
    if (t == 0) *listCount = n;
}
__global__ void width_scatter_kernel(const bbmsa_job *jobs, long long n, int maxColumns, unsigned *cursor, int *list) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[atomicAdd(&cursor[job_width_bucket(jobs[i], maxColumns)], 1u)] = (int)i;
}
}  // namespace bbmsa

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_));  \
            return BBMAP_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#include "msa_ctx.h"

extern "C" const char *bbmap_last_error(void) { return g_err; }
void bbmap_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
extern "C" int bbmap_abi_version(void) { return BBMAP_AMD_ABI_VERSION; }

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// The wavefront kernel's second geometry: 64 lanes per job, one job per 64-thread block, an LDS column buffer as wide as maxColumns.
// It takes the windows wider than the first pass's buffer (the wide pass) and, at the caller's request, whole launches that hold
// too few jobs to be anything but a wavefront's latency (bbmsa_set_latency_jobs).
static int setup_wide_pass(bbmsa_ctx *c) {
    if (c->wideBlocks > 0) return BBMAP_OK;
    c->wideR = (c->cfg.maxRows + 63) / 64;
    c->wideCols = c->cfg.maxColumns;
    c->wideTmpBytes = ((64 * c->wideR + c->wideCols + 8) + 3) & ~3;
    const int perJob = bbmsa::lds_job_ints(c->wideCols, c->wideTmpBytes);
    c->wideLdsBytes = (bbmsa::lds_table_ints(c->wideTableLen) + perJob) * 4;
    const void *wfn = bbmsa::fast_kernel_for(c->wideR, c->banded);
    if (wfn && c->wideLdsBytes <= 160 * 1024) {
        if (c->wideLdsBytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(wfn, hipFuncAttributeMaxDynamicSharedMemorySize, c->wideLdsBytes));
        int per = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, wfn, 64, c->wideLdsBytes));
        if (per < 1) per = 1;
        if (per > 8) per = 8;
        const long long slotDwords = (long long)(((c->wideCols + 64 - 1) >> 3) + 1) * c->wideR * 64;
        HIP_TRY(hipMalloc(&c->d_wideDir, (size_t)((long long)c->numCUs * per * slotDwords * 4)));
        c->wideDirSlotDwords = slotDwords;
        c->wideBlocks = c->numCUs * per;
    }
    return BBMAP_OK;
}

extern "C" int bbmsa_create(const bbmsa_config *cfg, bbmsa_ctx **out) {
    if (!cfg || !out) return fail(BBMAP_E_ARG, "bbmsa_create: null argument");
    *out = nullptr;
    const int scheme = cfg->reserved[2] & 0xFF;
    const bool legacyOnly = (cfg->reserved[2] & BBMSA_LEGACY_ONLY) != 0;
    if (scheme != BBMSA_SCHEME_11TS && scheme != BBMSA_SCHEME_9PACBIO) return fail(BBMAP_E_ARG, "bbmsa_create: unknown scoring scheme");
    if (scheme == BBMSA_SCHEME_11TS && (cfg->maxRows < 1 || cfg->maxRows > 640 || cfg->maxColumns < 1 || cfg->maxColumns > 4096))
        return fail(BBMAP_E_ARG, "bbmsa_create: maxRows must be 1..640 and maxColumns 1..4096");
    if (scheme == BBMSA_SCHEME_9PACBIO && (cfg->maxRows < 1 || cfg->maxRows > 6100 || cfg->maxColumns < 1 || cfg->maxColumns > 8192))
        return fail(BBMAP_E_ARG, "bbmsa_create: the PacBio scheme takes maxRows 1..6100 and maxColumns 1..8192");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BBMAP_E_NODEVICE, "bbmsa_create: no HIP device (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(BBMAP_E_ARG, "bbmsa_create: bad device ordinal");
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BBMAP_E_NODEVICE, "bbmsa_create: device is %s, this build targets gfx950 only", prop.gcnArchName);

    bbmsa_ctx *c = new (std::nothrow) bbmsa_ctx();
    if (!c) return fail(BBMAP_E_NOMEM, "bbmsa_create: out of host memory");
    memset(c, 0, sizeof *c);
    c->cfg = *cfg;
    c->device = cfg->device;
    c->numCUs = prop.multiProcessorCount;

    c->scheme = scheme;
    c->legacyOnly = legacyOnly;
    // every failure below leaves through bbmsa_destroy(c): nothing allocated so far is leaked
    struct Guard { bbmsa_ctx *c; ~Guard() { if (c) bbmsa_destroy(c); } } guard{c};
    if (legacyOnly) {
        // a context for bbmsa_fill_packed only (the per-call JNI shape): one scratch matrix, no batch buffers
        HIP_TRY(hipMalloc(&c->d_counters, 64));
        HIP_TRY(hipMemset(c->d_counters, 0, 64));
        const long long planeInts = (long long)(cfg->maxRows + 1) * (cfg->maxColumns + 2);
        c->genThreads = 1;
        HIP_TRY(hipMalloc(&c->d_matrix, (size_t)(3 * planeInts * 4)));
        HIP_TRY(hipMalloc(&c->d_limits, (size_t)((cfg->maxRows + cfg->maxColumns + 4) * 4)));
        for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&c->ev[i]));
        { const int rc = bbmsa_legacy_create(c); if (rc != BBMAP_OK) return rc; }
        guard.c = nullptr;
        *out = c;
        return BBMAP_OK;
    }
    if (scheme != BBMSA_SCHEME_11TS) {
        // 9PacBio: the strip-tiled wavefront kernel (msa_fill_strip.hip), one alignment per wavefront; banded fills and windows
        // narrower than the read are handed to the one-job-per-thread kernel
        HIP_TRY(hipMalloc(&c->d_counters, 64));
        HIP_TRY(hipMemset(c->d_counters, 0, 64));
        const long long planeInts = (long long)(cfg->maxRows + 1) * (cfg->maxColumns + 2);
        const long long perThread = 3 * planeInts * 4;
        long long budget = (long long)env_int("BBMSA_GENERIC_SCRATCH_MB", 40960) << 20;   // 6019 x 7600 (mapPacBio) needs 35 GB for one wavefront of matrices
        long long threads = budget / perThread;
        if (threads > 4096) threads = 4096;
        threads = (threads / 64) * 64;
        if (threads < 64) return fail(BBMAP_E_NOMEM, "bbmsa_create: BBMSA_GENERIC_SCRATCH_MB cannot hold one wavefront of scratch matrices for this maxRows x maxColumns");
        c->genThreads = (int)threads;
        HIP_TRY(hipMalloc(&c->d_matrix, (size_t)(threads * perThread)));
        HIP_TRY(hipMalloc(&c->d_limits, (size_t)(threads * (cfg->maxRows + cfg->maxColumns + 4) * 4)));
        {
            const int R = bbmsa::strip_rows_per_lane();
            c->stripLds = (cfg->maxColumns + 2) * 4 + ((cfg->maxColumns + 2 + 7) & ~7);     // horizLimit ints + reference bytes
            const void *kfn = bbmsa::strip_kernel_pacbio();
            if (c->stripLds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, c->stripLds));
            // Resident wavefronts per CU, from the LDS a wavefront takes (160 KB per CU; the kernel holds ~128 VGPRs, up to 4 waves per
            // SIMD, so LDS is what limits it).  Not asked of hipOccupancyMaxActiveBlocksPerMultiprocessor: that query was seen to fail with
            // hipErrorUnknown depending on what the process had done before (after the CPU oracle had run in it), for reasons the
            // runtime's log does not give; nothing here depends on the exact figure (the pipelined form no longer assumes co-residency).
            int per = (160 * 1024) / (c->stripLds + 512);
            if (per < 1) per = 1;
            if (per > 8) per = 8;
            c->stripBlocks = c->numCUs * per;
            const int strips = (cfg->maxRows + 64 * R - 1) / (64 * R);
            c->stripDwords = (long long)(((cfg->maxColumns + 64) >> 3) + 1) * R * 64;
            c->stripSlotDwords = c->stripDwords * strips;
            // traceback records: 4 bits per cell of every resident job (23 MB per 6,000 x 7,600 job)
            long long dirBudget = (long long)env_int("BBMSA_STRIP_DIR_MB", 32768) << 20;
            while (c->stripBlocks > c->numCUs && (long long)c->stripBlocks * c->stripSlotDwords * 4 > dirBudget) c->stripBlocks -= c->numCUs;
            while (c->stripBlocks > 1 && (long long)c->stripBlocks * c->stripSlotDwords * 4 > dirBudget) c->stripBlocks /= 2;
            {   // the pipelined form for launches with few jobs: the strips of one job in `strips` wavefronts
                const void *kp = bbmsa::strip_kernel_pacbio_pipelined();
                if (c->stripLds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, c->stripLds));
                int perP = (160 * 1024) / (c->stripLds + 512);
                if (perP < 1) perP = 1;
                if (perP > 8) perP = 8;
                c->pipeK = strips;
                c->pipeSlots = (c->numCUs * perP) / strips;                  // every wavefront of every slot is resident: the hand-shakes need it
                if (c->pipeSlots > c->stripBlocks) c->pipeSlots = c->stripBlocks;
                if (c->pipeSlots > 256) c->pipeSlots = 256;
                c->pipeJobsMax = env_int("BBMSA_STRIP_PIPE_JOBS", 512);      // launches with at most this many jobs take the pipelined form
                if (strips < 2 || c->pipeSlots < 1) c->pipeJobsMax = 0;
                if (c->pipeJobsMax > 0) {
                    HIP_TRY(hipMalloc(&c->d_pipeBoundary, (size_t)((long long)c->pipeSlots * strips * 3 * (cfg->maxColumns + 2) * 4)));
                    HIP_TRY(hipMalloc(&c->d_pipeSync, (size_t)((long long)c->pipeSlots * bbmsa::strip_pipe_sync_ints(strips) * 4)));
                }
            }
            HIP_TRY(hipMalloc(&c->d_dir, (size_t)((long long)c->stripBlocks * c->stripSlotDwords * 4)));
            HIP_TRY(hipMalloc(&c->d_stripBoundary, (size_t)((long long)c->stripBlocks * 6 * (cfg->maxColumns + 2) * 4)));
            HIP_TRY(hipMalloc(&c->d_stripTmp, (size_t)((long long)c->stripBlocks * (cfg->maxRows + cfg->maxColumns + 8))));
        }
        for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&c->ev[i]));
        guard.c = nullptr;
        *out = c;
        return BBMAP_OK;
    }
    // lanes per job / rows per lane: smallest lane group whose <=10 rows per lane cover maxRows
    int G = cfg->reserved[0];
    if (G != 16 && G != 32 && G != 64) {
        G = env_int("BBMSA_LANES_PER_JOB", 0);
        if (G != 16 && G != 32 && G != 64) G = (cfg->maxRows <= 320) ? 32 : 64;
    }
    while (G < 64 && (cfg->maxRows + G - 1) / G > 10) G *= 2;
    c->G = G;
    c->R = (cfg->maxRows + G - 1) / G;
    int fastCols = cfg->reserved[1] > 0 ? cfg->reserved[1] : env_int("BBMSA_FAST_COLS", 0);
    if (fastCols <= 0) fastCols = cfg->maxColumns < 640 ? cfg->maxColumns : 640;
    if (fastCols > cfg->maxColumns) fastCols = cfg->maxColumns;
    c->fastCols = fastCols;
    c->tmpBytes = ((G * c->R + fastCols + 8) + 3) & ~3;
    const int jobsPerWave = 64 / G;
    const int perJobInts = bbmsa::lds_job_ints(fastCols, c->tmpBytes);
    {   // index = time + needed: time <= min(longer side + 1, 2047 (clamped)), needed <= rows.  The first pass only takes windows of
        // up to fastCols columns, so its tables are sized for those; the wide pass has its own (wideTableLen).
        const int side = (cfg->maxColumns > cfg->maxRows ? cfg->maxColumns : cfg->maxRows) + 2;
        c->wideTableLen = (side < 2048 ? side : 2048) + cfg->maxRows + 8;
        const int sideF = (fastCols > cfg->maxRows ? fastCols : cfg->maxRows) + 2;
        c->tableLen = (sideF < 2048 ? sideF : 2048) + cfg->maxRows + 8;
    }
    if (c->tableLen > bbmsa::kTableLen) c->tableLen = bbmsa::kTableLen;
    if (c->wideTableLen > bbmsa::kTableLen) c->wideTableLen = bbmsa::kTableLen;
    c->tableLen = (c->tableLen + 3) & ~3; c->wideTableLen = (c->wideTableLen + 3) & ~3;
    c->ldsBytes = (bbmsa::lds_table_ints(c->tableLen) + 4 * jobsPerWave * perJobInts) * 4;
    if (c->ldsBytes > 160 * 1024) return fail(BBMAP_E_ARG, "bbmsa_create: fast-path LDS budget exceeded; lower reserved[1] (fastCols)");

    c->banded = !(cfg->bandwidth < 1 && cfg->bandwidthRatio <= 0.0f);
    const void *kfn = bbmsa::fast_kernel_for(c->R, c->banded);
    if (!kfn) return fail(BBMAP_E_ARG, "bbmsa_create: no kernel for this rows-per-lane");
    if (c->ldsBytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, c->ldsBytes));
    int blocksPerCU = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCU, kfn, 256, c->ldsBytes));
    if (blocksPerCU < 1) blocksPerCU = 1;
    const int capBlocks = env_int("BBMSA_BLOCKS_PER_CU", 0);
    if (capBlocks > 0 && capBlocks < blocksPerCU) blocksPerCU = capBlocks;
    c->blocks = c->numCUs * blocksPerCU;

    const int maxSteps = fastCols + G - 1;
    c->dirSlotDwords = (long long)((maxSteps >> 3) + 1) * c->R * G;
    const long long slots = (long long)c->blocks * 4 * jobsPerWave;
    HIP_TRY(hipMalloc(&c->d_dir, (size_t)(slots * c->dirSlotDwords * 4)));
    HIP_TRY(hipMalloc(&c->d_counters, 64));
    HIP_TRY(hipMemset(c->d_counters, 0, 64));

    // generic kernel scratch: as many threads as a 2 GiB matrix budget allows (at least one wave)
    const long long planeInts = (long long)(cfg->maxRows + 1) * (cfg->maxColumns + 2);
    const long long perThread = 3 * planeInts * 4;
    long long budget = (long long)env_int("BBMSA_GENERIC_SCRATCH_MB", 2048) << 20;
    long long threads = budget / perThread;
    if (threads > 16384) threads = 16384;
    threads = (threads / 64) * 64;
    if (threads < 64) return fail(BBMAP_E_NOMEM, "bbmsa_create: BBMSA_GENERIC_SCRATCH_MB cannot hold one wavefront of scratch matrices for this maxRows x maxColumns");
    c->genThreads = (int)threads;
    HIP_TRY(hipMalloc(&c->d_matrix, (size_t)(threads * perThread)));
    HIP_TRY(hipMalloc(&c->d_limits, (size_t)(threads * (cfg->maxRows + cfg->maxColumns + 4) * 4)));
    // wide pass geometry (only when some windows can exceed the first pass's column buffer)
    c->wideBlocks = 0; c->latencyJobs = 0;
    if (cfg->maxColumns > fastCols) { const int rc = setup_wide_pass(c); if (rc != BBMAP_OK) return rc; }
    // narrow-window kernel: only without a band (a band changes the window rule); BBMSA_NARROW=0 disables it
    c->narrowBlocks = 0; c->narrowOff = false; c->narrowUsed = false;
    c->sortByWidth = false; c->d_widthHist = nullptr;
    c->narrowSlack = env_int("BBMSA_NARROW_SLACK", 2000);
    if (!c->banded && env_int("BBMSA_NARROW", 1) != 0) {
        int perCU = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *)bbmsa::msa_fill_narrow_kernel, 64, 0));
        if (perCU < 1) perCU = 1;
        if (perCU > 16) perCU = 16;
        c->narrowBlocks = c->numCUs * perCU;
        HIP_TRY(hipMalloc(&c->d_narrowDir, (size_t)c->narrowBlocks * (size_t)(cfg->maxRows + 1) * 64 * 8));
    }
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&c->ev[i]));
    guard.c = nullptr;
    *out = c;
    return BBMAP_OK;
}

extern "C" void bbmsa_destroy(bbmsa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    bbmsa_legacy_destroy(c);
    if (c->d_dir) (void)hipFree(c->d_dir);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_widthHist) (void)hipFree(c->d_widthHist);

    if (c->d_slowList) (void)hipFree(c->d_slowList);
    if (c->d_matrix) (void)hipFree(c->d_matrix);
    if (c->d_limits) (void)hipFree(c->d_limits);
    if (c->d_wideDir) (void)hipFree(c->d_wideDir);
    if (c->d_slowList2) (void)hipFree(c->d_slowList2);
    if (c->d_narrowDir) (void)hipFree(c->d_narrowDir);
    if (c->d_fastList) (void)hipFree(c->d_fastList);
    if (c->d_gref) (void)hipFree(c->d_gref);
    if (c->d_gaux) (void)hipFree(c->d_gaux);
    if (c->d_gjobs) (void)hipFree(c->d_gjobs);
    if (c->d_stripBoundary) (void)hipFree(c->d_stripBoundary);
    if (c->d_stripTmp) (void)hipFree(c->d_stripTmp);
    if (c->d_pipeBoundary) (void)hipFree(c->d_pipeBoundary);
    if (c->d_pipeSync) (void)hipFree(c->d_pipeSync);
    for (int i = 0; i < 4; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete c;
}

extern "C" int bbmsa_align_batch_device(bbmsa_ctx *c, void *stream_, int64_t n_jobs,
                                        const bbmsa_job *jobs, const uint8_t *reads, const uint8_t *refs,
                                        bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    return bbmsa_align_impl(c, stream_, n_jobs, nullptr, jobs, reads, refs, results, match, match_stride);
}

extern "C" int bbmsa_align_batch_device_indirect(bbmsa_ctx *c, void *stream_, const uint32_t *n_jobs_dev, int64_t max_jobs,
                                                 const bbmsa_job *jobs, const uint8_t *reads, const uint8_t *refs,
                                                 bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!n_jobs_dev) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device_indirect: null counter");
    return bbmsa_align_impl(c, stream_, max_jobs, n_jobs_dev, jobs, reads, refs, results, match, match_stride);
}

// n_jobs_dev == NULL: n_jobs jobs.  Otherwise n_jobs is the capacity of the buffers and the kernels read the real count
// from *n_jobs_dev when they run.
int bbmsa_align_impl(bbmsa_ctx *c, void *stream_, int64_t n_jobs, const uint32_t *n_jobs_dev,
                     const bbmsa_job *jobs, const uint8_t *reads, const uint8_t *refs,
                     bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!c) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device: null context");
    if (n_jobs < 0 || n_jobs > 0x7fffffffLL) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device: n_jobs out of range");
    if (n_jobs == 0) return BBMAP_OK;
    if (!jobs || !reads || !refs || !results) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device: null buffer");
    if (match && match_stride < 1) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device: match_stride must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    if (c->legacyOnly) return fail(BBMAP_E_ARG, "bbmsa_align_batch_device: this context was created for bbmsa_fill_packed only (BBMSA_LEGACY_ONLY)");
    if (c->scheme != BBMSA_SCHEME_11TS) {
        if (n_jobs > c->slowCap) {
            if (c->d_slowList) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_slowList)); c->d_slowList = nullptr; }
            HIP_TRY(hipMalloc(&c->d_slowList, (size_t)n_jobs * 4));
            c->slowCap = n_jobs;
        }
        HIP_TRY(hipMemsetAsync(c->d_counters, 0, 64, stream));
        HIP_TRY(hipEventRecord(c->ev[0], stream));
        HIP_TRY(hipEventRecord(c->ev[3], stream));
        bbmsa::StripParams sp;
        sp.jobs = jobs; sp.reads = reads; sp.refs = refs; sp.results = results; sp.match = match; sp.njobs = n_jobs; sp.njobs_dev = n_jobs_dev;
        sp.queue = c->d_counters; sp.dirbuf = c->d_dir; sp.dir_slot_dwords = c->stripSlotDwords; sp.dir_strip_dwords = c->stripDwords;
        sp.boundary = c->d_stripBoundary; sp.tmpbuf = c->d_stripTmp; sp.slow_list = c->d_slowList; sp.slow_count = c->d_counters + 1;
        sp.match_stride = match_stride; sp.maxRows = c->cfg.maxRows; sp.maxColumns = c->cfg.maxColumns;
        sp.bandwidth = c->cfg.bandwidth; sp.bandwidthRatio = c->cfg.bandwidthRatio;
        long long sblocks = n_jobs < c->stripBlocks ? n_jobs : c->stripBlocks;
        sp.pipeK = 0; sp.pipeSlots = 0; sp.pipeBoundary = nullptr; sp.pipeSync = nullptr;
        sp.pipeSpinLimit = env_int("BBMSA_PIPE_SPIN_LIMIT", 1 << 21);          // polls before a wave of the pipelined form gives up (~3 s; tests force timeouts)
        if (sp.pipeSpinLimit < 1) sp.pipeSpinLimit = 1;
        void *sargs[] = {&sp};
        if (!n_jobs_dev && c->pipeJobsMax > 0 && n_jobs <= c->pipeJobsMax) {
            // few jobs (the late scoreSlow rounds of mapPacBio): a lone 6,000 x 6,100 fill is one wavefront's dependent chain, 370 ms;
            // with its strips pipelined over `pipeK` wavefronts it is ~50
            const long long slots = n_jobs < c->pipeSlots ? n_jobs : c->pipeSlots;
            sp.pipeK = c->pipeK; sp.pipeSlots = (int)slots; sp.pipeBoundary = c->d_pipeBoundary; sp.pipeSync = c->d_pipeSync;
            HIP_TRY(hipMemsetAsync(c->d_pipeSync, 0, (size_t)(slots * bbmsa::strip_pipe_sync_ints(c->pipeK) * 4), stream));
            HIP_TRY(hipLaunchKernel(bbmsa::strip_kernel_pacbio_pipelined(), dim3((unsigned)(slots * c->pipeK)), dim3(64), sargs, (size_t)c->stripLds, stream));
        } else
        HIP_TRY(hipLaunchKernel(bbmsa::strip_kernel_pacbio(), dim3((unsigned)sblocks), dim3(64), sargs, (size_t)c->stripLds, stream));
        HIP_TRY(hipEventRecord(c->ev[1], stream));
        bbmsa::GenericParams gp;
        gp.jobs = jobs; gp.reads = reads; gp.refs = refs; gp.results = results; gp.match = match;
        gp.list = c->d_slowList; gp.list_count = c->d_counters + 1; gp.njobs = n_jobs; gp.njobs_dev = n_jobs_dev;
        gp.matrix = c->d_matrix; gp.limits = c->d_limits; gp.queue = c->d_counters + 2;
        gp.match_stride = match_stride; gp.maxRows = c->cfg.maxRows; gp.maxColumns = c->cfg.maxColumns;
        gp.bandwidth = c->cfg.bandwidth; gp.bandwidthRatio = c->cfg.bandwidthRatio;
        hipLaunchKernelGGL(bbmsa::msa_fill_generic_kernel<bbmsa::Scheme9PacBio>, dim3(c->genThreads / 64), dim3(64), 0, stream, gp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[2], stream));
        c->timed = true;
        return BBMAP_OK;
    }
    if (n_jobs > c->slowCap) {
        if (c->d_slowList) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_slowList)); c->d_slowList = nullptr; }
        if (c->d_slowList2) { HIP_TRY(hipFree(c->d_slowList2)); c->d_slowList2 = nullptr; }
        HIP_TRY(hipMalloc(&c->d_slowList, (size_t)n_jobs * 4));
        if (c->wideBlocks > 0) HIP_TRY(hipMalloc(&c->d_slowList2, (size_t)n_jobs * 4));
        c->slowCap = n_jobs;
    }
    const bool sortJobs = c->sortByWidth && !n_jobs_dev && n_jobs >= 256 && n_jobs > c->latencyJobs && !(c->narrowBlocks > 0 && !c->narrowOff);
    if ((c->narrowBlocks > 0 || sortJobs) && n_jobs > c->fastCap) {
        if (c->d_fastList) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(c->d_fastList)); c->d_fastList = nullptr; }
        HIP_TRY(hipMalloc(&c->d_fastList, (size_t)n_jobs * 4));
        c->fastCap = n_jobs;
    }
    // counters: [0] fast queue, [1] slow count, [2] generic queue, [3] narrow queue, [4] fast-list count,
    //           [5] jobs finished by the narrow kernel, [6] candidates it handed on
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 64, stream));
    HIP_TRY(hipEventRecord(c->ev[0], stream));
    const bool useNarrow = c->narrowBlocks > 0 && !c->narrowOff;
    c->narrowUsed = useNarrow;
    if (useNarrow) {
        bbmsa::NarrowParams np;
        np.jobs = jobs; np.reads = reads; np.refs = refs; np.results = results; np.match = match; np.njobs = n_jobs; np.njobs_dev = n_jobs_dev;
        np.queue = c->d_counters + 3; np.fast_list = c->d_fastList; np.fast_count = c->d_counters + 4;
        np.dirbuf = c->d_narrowDir; np.stats = c->d_counters + 5;
        np.match_stride = match_stride; np.maxRows = c->cfg.maxRows; np.maxColumns = c->cfg.maxColumns;
        np.bandwidth = c->cfg.bandwidth; np.bandwidthRatio = c->cfg.bandwidthRatio; np.maxSlack = c->narrowSlack;
        long long nb = (n_jobs + 63) / 64;
        if (nb > c->narrowBlocks) nb = c->narrowBlocks;
        hipLaunchKernelGGL(bbmsa::msa_fill_narrow_kernel, dim3((unsigned)nb), dim3(64), 0, stream, np);
        HIP_TRY(hipGetLastError());
    }
    if (sortJobs) {                    // (never together with the narrow kernel: both write the wavefront kernel's list)
        if (!c->d_widthHist) HIP_TRY(hipMalloc(&c->d_widthHist, bbmsa::WIDTH_BUCKETS * 4));
        HIP_TRY(hipMemsetAsync(c->d_widthHist, 0, bbmsa::WIDTH_BUCKETS * 4, stream));
        const unsigned sb = (unsigned)((n_jobs + 255) / 256);
        hipLaunchKernelGGL(bbmsa::width_hist_kernel, dim3(sb), dim3(256), 0, stream, jobs, (long long)n_jobs, c->cfg.maxColumns, c->d_widthHist);
        hipLaunchKernelGGL(bbmsa::width_scan_kernel, dim3(1), dim3(bbmsa::WIDTH_BUCKETS), 0, stream, c->d_widthHist, c->d_counters + 4, (unsigned)n_jobs);
        hipLaunchKernelGGL(bbmsa::width_scatter_kernel, dim3(sb), dim3(256), 0, stream, jobs, (long long)n_jobs, c->cfg.maxColumns, c->d_widthHist, c->d_fastList);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev[3], stream));

    bbmsa::FillParams fp;
    fp.jobs = jobs; fp.reads = reads; fp.refs = refs; fp.results = results; fp.match = match;
    fp.njobs = n_jobs; fp.njobs_dev = n_jobs_dev;
    fp.queue = c->d_counters; fp.dirbuf = c->d_dir; fp.dir_slot_dwords = c->dirSlotDwords;
    fp.list = (useNarrow || sortJobs) ? c->d_fastList : nullptr; fp.list_count = c->d_counters + 4; fp.priority = 0;
    fp.slow_list = c->d_slowList; fp.slow_count = c->d_counters + 1;
    fp.match_stride = match_stride; fp.lanesPerJob = c->G; fp.fastCols = c->fastCols; fp.tmpBytes = c->tmpBytes; fp.tableLen = c->tableLen;
    fp.maxRows = c->cfg.maxRows; fp.maxColumns = c->cfg.maxColumns;
    fp.bandwidth = c->cfg.bandwidth; fp.bandwidthRatio = c->cfg.bandwidthRatio;

    // A launch with few jobs is a wavefront's latency, not throughput: (columns + lanes - 1) steps of one dependent chain.  The wide
    // pass's geometry (64 lanes x 3 rows, one job per block) has the shorter chain per step (3 rows instead of 5: ~450 instead of
    // ~740 instructions), so such launches go to it directly (bbmsa_set_latency_jobs; the mapper's late rounds hold a few hundred fills).
    const bool latency = c->wideBlocks > 0 && !n_jobs_dev && !useNarrow && n_jobs <= c->latencyJobs;
    const int jobsPerBlock = 4 * (64 / c->G);
    long long blocks = (n_jobs + jobsPerBlock - 1) / jobsPerBlock;
    if (blocks > c->blocks) blocks = c->blocks;
    void *args[] = {&fp};
    if (!latency)
        HIP_TRY(hipLaunchKernel(bbmsa::fast_kernel_for(c->R, c->banded), dim3((unsigned)blocks), dim3(256), args, (size_t)c->ldsBytes, stream));
    const int *genList = c->d_slowList;
    const unsigned int *genCount = c->d_counters + 1;
    if (c->wideBlocks > 0) {
        // wide pass over the first pass's hand-overs; what it cannot take either (banded rows with holes) goes on to the
        // generic kernel through the second list ([7] = its count, [8] = wide queue)
        bbmsa::FillParams wp = fp;
        wp.queue = c->d_counters + 8; wp.dirbuf = c->d_wideDir; wp.dir_slot_dwords = c->wideDirSlotDwords;
        wp.list = c->d_slowList; wp.list_count = c->d_counters + 1;
        if (latency) { wp.list = nullptr; wp.list_count = nullptr; }          // every job of the launch
        wp.slow_list = c->d_slowList2; wp.slow_count = c->d_counters + 7;
        wp.lanesPerJob = 64; wp.fastCols = c->wideCols; wp.tmpBytes = c->wideTmpBytes; wp.tableLen = c->wideTableLen;
        static const int widePrio = env_int("BBMSA_WIDE_PRIORITY", 2);
        wp.priority = widePrio;
        void *wargs[] = {&wp};
        const long long wblocks = latency && n_jobs < c->wideBlocks ? n_jobs : c->wideBlocks;
        HIP_TRY(hipLaunchKernel(bbmsa::fast_kernel_for(c->wideR, c->banded), dim3((unsigned)wblocks), dim3(64), wargs, (size_t)c->wideLdsBytes, stream));
        genList = c->d_slowList2; genCount = c->d_counters + 7;
    }
    HIP_TRY(hipEventRecord(c->ev[1], stream));

    bbmsa::GenericParams gp;
    gp.jobs = jobs; gp.reads = reads; gp.refs = refs; gp.results = results; gp.match = match;
    gp.list = genList; gp.list_count = genCount; gp.njobs = n_jobs; gp.njobs_dev = n_jobs_dev;
    gp.matrix = c->d_matrix; gp.limits = c->d_limits; gp.queue = c->d_counters + 2;
    gp.match_stride = match_stride; gp.maxRows = c->cfg.maxRows; gp.maxColumns = c->cfg.maxColumns;
    gp.bandwidth = c->cfg.bandwidth; gp.bandwidthRatio = c->cfg.bandwidthRatio;
    hipLaunchKernelGGL(bbmsa::msa_fill_generic_kernel<bbmsa::Scheme11ts>, dim3(c->genThreads / 64), dim3(64), 0, stream, gp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[2], stream));
    c->timed = true;
    return BBMAP_OK;
}

void bbmsa_use_narrow(bbmsa_ctx *c, bool on) { if (c) c->narrowOff = !on; }
void bbmsa_sort_by_width(bbmsa_ctx *c, bool on) { if (c) c->sortByWidth = on; }
int bbmsa_set_latency_jobs(bbmsa_ctx *c, int64_t n) {
    if (!c || c->scheme != BBMSA_SCHEME_11TS || c->legacyOnly) return BBMAP_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (n > 0) { const int rc = setup_wide_pass(c); if (rc != BBMAP_OK) return rc; }
    if (n > 0 && c->slowCap > 0 && !c->d_slowList2) HIP_TRY(hipMalloc(&c->d_slowList2, (size_t)c->slowCap * 4));   // (its hand-over list)
    c->latencyJobs = c->wideBlocks > 0 ? n : 0;
    return BBMAP_OK;
}
int bbmsa_wait_first_pass(bbmsa_ctx *c, void *waiter) {
    if (!c || !c->timed) return BBMAP_OK;
    HIP_TRY(hipStreamWaitEvent((hipStream_t)waiter, c->ev[3], 0));       // recorded right in front of the last launch's first pass
    return BBMAP_OK;
}

extern "C" int bbmsa_last_kernel_ms(bbmsa_ctx *c, float *ms_fast, float *ms_slow) {
    if (!c || !c->timed) return fail(BBMAP_E_ARG, "bbmsa_last_kernel_ms: nothing launched yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    if (ms_fast) *ms_fast = a;
    if (ms_slow) *ms_slow = b;
    return BBMAP_OK;
}

extern "C" int bbmsa_last_kernel_ms3(bbmsa_ctx *c, float *ms3) {
    if (!c || !c->timed || !ms3) return fail(BBMAP_E_ARG, "bbmsa_last_kernel_ms3: nothing launched yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms3[0], c->ev[0], c->ev[3]));
    HIP_TRY(hipEventElapsedTime(&ms3[1], c->ev[3], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms3[2], c->ev[1], c->ev[2]));
    return BBMAP_OK;
}

extern "C" int bbmsa_last_counts(bbmsa_ctx *c, int64_t *counts4) {
    if (!c || !c->timed || !counts4) return fail(BBMAP_E_ARG, "bbmsa_last_counts: nothing launched yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    unsigned h[16];
    HIP_TRY(hipMemcpy(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    counts4[0] = h[5]; counts4[1] = h[6]; counts4[2] = c->narrowUsed ? h[4] : 0; counts4[3] = c->wideBlocks > 0 ? h[7] : h[1];
    if (getenv("BBMAP_DP_COUNTS") && c->wideBlocks > 0) fprintf(stderr, "   (first pass handed %u jobs to the wide pass)\n", h[1]);
    return BBMAP_OK;
}

extern "C" int bbmsa_align_batch(bbmsa_ctx *c, int64_t n_jobs, const bbmsa_job *jobs,
                                 const uint8_t *reads, int64_t reads_bytes,
                                 const uint8_t *refs, int64_t refs_bytes,
                                 bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!c) return fail(BBMAP_E_ARG, "bbmsa_align_batch: null context");
    if (n_jobs == 0) return BBMAP_OK;
    if (n_jobs < 0 || !jobs || !reads || !refs || !results || reads_bytes < 0 || refs_bytes < 0)
        return fail(BBMAP_E_ARG, "bbmsa_align_batch: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    // every job must stay inside the buffers it was given
    for (int64_t i = 0; i < n_jobs; i++) {
        const bbmsa_job &j = jobs[i];
        if (j.read_len < 0 || j.read_off < 0 || j.read_off + j.read_len > reads_bytes)
            return fail(BBMAP_E_ARG, "bbmsa_align_batch: a read lies outside the reads buffer");
        if (j.ref_len < 0 || j.ref_off < 0 || j.ref_off + j.ref_len > refs_bytes)
            return fail(BBMAP_E_ARG, "bbmsa_align_batch: a reference array lies outside the refs buffer");
        if (!(j.flags & BBMSA_CLAMP_WINDOW) && (j.refStartLoc < 0 || j.refEndLoc >= j.ref_len))
            return fail(BBMAP_E_ARG, "bbmsa_align_batch: window outside its reference array (set BBMSA_CLAMP_WINDOW to clamp)");
    }
    bbmsa_job *d_jobs = nullptr; uint8_t *d_reads = nullptr, *d_refs = nullptr, *d_match = nullptr; bbmsa_result *d_res = nullptr;
    int rc = BBMAP_OK;
#define TRY_GOTO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_)); rc = BBMAP_E_HIP; goto done; } } while (0)
    TRY_GOTO(hipMalloc(&d_jobs, (size_t)n_jobs * sizeof(bbmsa_job)));
    TRY_GOTO(hipMalloc(&d_reads, (size_t)(reads_bytes > 0 ? reads_bytes : 1)));
    TRY_GOTO(hipMalloc(&d_refs, (size_t)(refs_bytes > 0 ? refs_bytes : 1)));
    TRY_GOTO(hipMalloc(&d_res, (size_t)n_jobs * sizeof(bbmsa_result)));
    if (match) TRY_GOTO(hipMalloc(&d_match, (size_t)n_jobs * (size_t)match_stride));
    TRY_GOTO(hipMemcpy(d_jobs, jobs, (size_t)n_jobs * sizeof(bbmsa_job), hipMemcpyHostToDevice));
    TRY_GOTO(hipMemcpy(d_reads, reads, (size_t)reads_bytes, hipMemcpyHostToDevice));
    TRY_GOTO(hipMemcpy(d_refs, refs, (size_t)refs_bytes, hipMemcpyHostToDevice));
    TRY_GOTO(hipMemset(d_res, 0xff, (size_t)n_jobs * sizeof(bbmsa_result)));
    rc = bbmsa_align_batch_device(c, nullptr, n_jobs, d_jobs, d_reads, d_refs, d_res, d_match, match_stride);
    if (rc != BBMAP_OK) goto done;
    TRY_GOTO(hipStreamSynchronize(nullptr));
    TRY_GOTO(hipMemcpy(results, d_res, (size_t)n_jobs * sizeof(bbmsa_result), hipMemcpyDeviceToHost));
    if (match) TRY_GOTO(hipMemcpy(match, d_match, (size_t)n_jobs * (size_t)match_stride, hipMemcpyDeviceToHost));
done:
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_reads) (void)hipFree(d_reads);
    if (d_refs) (void)hipFree(d_refs);
    if (d_res) (void)hipFree(d_res);
    if (d_match) (void)hipFree(d_match);
    return rc;
#undef TRY_GOTO
}
