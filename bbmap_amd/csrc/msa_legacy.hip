// The legacy per-call fills (`usejni=t` with the unmodified Java classes): jni/MultiStateAligner11tsJNI.c:707-812 hands ONE fill
// per JNI call to the native code and reads the three score planes back out of `packed`.  A GPU cannot serve that shape one call
// at a time, so this file is a process-wide service behind bbmsa_fill_submit / bbmsa_fill_collect / bbmsa_fill_packed:
//
//  * every context created with BBMSA_LEGACY_ONLY owns persistent buffers (pinned host + device, two batches) and its own stream:
//    no allocation, no default-stream synchronisation per call;
//  * concurrent callers (BBMap runs one mapping thread per core, each with its own MSA object) are combined: a caller copies its
//    read and its reference WINDOW into the open batch's pinned arena; the first caller that finds no leader becomes the leader,
//    closes the batch, uploads it, runs it as ONE launch of the wavefront kernel in matrix-materialising mode (one 64-lane
//    wavefront per fill, msa_fill_fast.hip MAT) and downloads planes, limits and results; followers sleep on a condition variable.
//    While a batch is on the device the next one fills up, so the batch size adapts to the arrival rate (1 when there is one
//    caller);
//  * fills the wavefront kernel hands back (a window more than two columns narrower than the read, banded rows with holes) are
//    redone by the one-thread generic kernel into the scratch matrix and copied into the same staging layout on the device;
//  * bbmsa_fill_collect is a pure memcpy out of the pinned staging area into the caller's `packed` (Java layout) -- the JNI shim
//    calls it inside its one short critical region -- and also returns vertLimit / horizLimit, which the reference's native code
//    fills as a side effect (jni/...c:413-438).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "msa_common.h"
#include "msa_ctx.h"

void bbmap_set_error(const char *msg);

namespace bbmsa {
const void *fast_kernel_mat_for(int R, bool banded);
template <class S> __global__ void msa_fill_generic_kernel(const GenericParams p);
}  // namespace bbmsa

namespace {

int lfail(int code, const char *msg) { bbmap_set_error(msg); return code; }
#define L_TRY(expr)                                                                               \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char b_[400]; snprintf(b_, sizeof b_, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            bbmap_set_error(b_);                                                                  \
            return BBMAP_E_HIP;                                                                   \
        }                                                                                         \
    } while (0)

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

enum { OPEN = 0, RUNNING = 1, DONE = 2 };

struct Batch {
    // pinned input arena and its device mirror: [jobs | plane_off | limits_off | sequence bytes]
    uint8_t *h_in, *d_in;
    // outputs: device pools and their pinned copies
    int *d_planes, *h_planes;
    int *d_limits, *h_limits;
    bbmsa_result *d_results, *h_results;
    unsigned int *d_counters, *h_counters;       // [0] queue, [1] hand-over count, [2] generic queue
    int *d_slow, *h_slow;
    int njobs;
    long long usedBytes, usedInts, usedLimits;
    int state, readers, rc;
    long long gen;
    char err[200];
    bbmsa_job *jobs() const { return (bbmsa_job *)h_in; }
};

}  // namespace

struct bbmsa_legacy {
    std::mutex mu;
    std::condition_variable cv;
    hipStream_t stream;
    Batch B[2];
    int open;
    bool leaderActive;
    long long nextGen;
    // capacities
    int capJobs;
    long long capBytes, capInts, capLimits;
    long long offPlaneOff, offLimitsOff, offBytes;      // section offsets inside the input arena
    // kernel geometry (64 lanes per fill, one fill per 64-thread block, column buffer as wide as maxColumns)
    int R, cols, tmpBytes, tableLen, ldsBytes, blocks;
    long long dirSlotDwords;
    unsigned int *d_dir;
    bool wave;                                           // false: every fill goes to the generic kernel (9PacBio scheme)
    // statistics
    long long calls, launches, handed;
    long long nsWave, nsHanded, nsWaitReaders;      // leader time: first synchronisation, hand-over pass, waiting for collectors
};

namespace {

int alloc_batch(bbmsa_legacy *S, Batch &b) {
    memset(&b, 0, sizeof b);
    const size_t inBytes = (size_t)(S->offBytes + S->capBytes);
    L_TRY(hipHostMalloc((void **)&b.h_in, inBytes, hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&b.d_in, inBytes));
    L_TRY(hipMalloc((void **)&b.d_planes, (size_t)S->capInts * 4));
    L_TRY(hipHostMalloc((void **)&b.h_planes, (size_t)S->capInts * 4, hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&b.d_limits, (size_t)S->capLimits * 4));
    L_TRY(hipHostMalloc((void **)&b.h_limits, (size_t)S->capLimits * 4, hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&b.d_results, (size_t)S->capJobs * sizeof(bbmsa_result)));
    L_TRY(hipHostMalloc((void **)&b.h_results, (size_t)S->capJobs * sizeof(bbmsa_result), hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&b.d_counters, 64));
    L_TRY(hipHostMalloc((void **)&b.h_counters, 64, hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&b.d_slow, (size_t)S->capJobs * 4));
    L_TRY(hipHostMalloc((void **)&b.h_slow, (size_t)S->capJobs * 4, hipHostMallocDefault));
    b.state = OPEN;
    return BBMAP_OK;
}

void free_batch(Batch &b) {
    if (b.h_in) (void)hipHostFree(b.h_in);
    if (b.d_in) (void)hipFree(b.d_in);
    if (b.d_planes) (void)hipFree(b.d_planes);
    if (b.h_planes) (void)hipHostFree(b.h_planes);
    if (b.d_limits) (void)hipFree(b.d_limits);
    if (b.h_limits) (void)hipHostFree(b.h_limits);
    if (b.d_results) (void)hipFree(b.d_results);
    if (b.h_results) (void)hipHostFree(b.h_results);
    if (b.d_counters) (void)hipFree(b.d_counters);
    if (b.h_counters) (void)hipHostFree(b.h_counters);
    if (b.d_slow) (void)hipFree(b.d_slow);
    if (b.h_slow) (void)hipHostFree(b.h_slow);
    memset(&b, 0, sizeof b);
}

// One closed batch through the device.  Called by the leader without the lock; everything is queued on the service's stream and
// the single synchronisation is the last line.
int run_batch(bbmsa_ctx *c, Batch &b) {
    bbmsa_legacy *S = c->legacy;
    L_TRY(hipSetDevice(c->device));
    hipStream_t st = S->stream;
    const int n = b.njobs;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point a) { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a).count(); };
    L_TRY(hipMemcpyAsync(b.d_in, b.h_in, (size_t)(S->offBytes + b.usedBytes), hipMemcpyHostToDevice, st));
    L_TRY(hipMemsetAsync(b.d_counters, 0, 64, st));
    const bbmsa_job *d_jobs = (const bbmsa_job *)b.d_in;
    const long long *d_planeOff = (const long long *)(b.d_in + S->offPlaneOff);
    const long long *d_limitsOff = (const long long *)(b.d_in + S->offLimitsOff);
    const uint8_t *d_bytes = b.d_in + S->offBytes;
    const long long *h_planeOff = (const long long *)(b.h_in + S->offPlaneOff);
    const long long *h_limitsOff = (const long long *)(b.h_in + S->offLimitsOff);
    auto download = [&]() -> int {
        L_TRY(hipMemcpyAsync(b.h_planes, b.d_planes, (size_t)b.usedInts * 4, hipMemcpyDeviceToHost, st));
        L_TRY(hipMemcpyAsync(b.h_limits, b.d_limits, (size_t)b.usedLimits * 4, hipMemcpyDeviceToHost, st));
        L_TRY(hipMemcpyAsync(b.h_results, b.d_results, (size_t)n * sizeof(bbmsa_result), hipMemcpyDeviceToHost, st));
        return BBMAP_OK;
    };
    int nslow = n;
    if (S->wave) {
        bbmsa::FillParams fp;
        memset(&fp, 0, sizeof fp);
        fp.jobs = d_jobs; fp.reads = d_bytes; fp.refs = d_bytes; fp.results = b.d_results; fp.match = nullptr;
        fp.njobs = n; fp.njobs_dev = nullptr;
        fp.queue = b.d_counters; fp.dirbuf = S->d_dir; fp.dir_slot_dwords = S->dirSlotDwords;
        fp.list = nullptr; fp.list_count = nullptr; fp.priority = 0;
        fp.slow_list = b.d_slow; fp.slow_count = b.d_counters + 1;
        fp.match_stride = 0; fp.lanesPerJob = 64; fp.fastCols = S->cols; fp.tmpBytes = S->tmpBytes; fp.tableLen = S->tableLen;
        fp.maxRows = c->cfg.maxRows; fp.maxColumns = c->cfg.maxColumns;
        fp.bandwidth = c->cfg.bandwidth; fp.bandwidthRatio = c->cfg.bandwidthRatio;
        fp.planes = b.d_planes; fp.plane_off = d_planeOff; fp.limits = b.d_limits; fp.limits_off = d_limitsOff;
        void *args[] = {&fp};
        const int blocks = n < S->blocks ? n : S->blocks;
        // rows per lane from the longest read of THIS batch, not from maxRows: a 150-base read in a 601-row context runs 3 rows per
        // lane on 50 lanes instead of 10 rows on 15 (the rows of a lane are a serial chain; 2.3x per fill)
        int maxLen = 1;
        for (int i = 0; i < n; i++) if (b.jobs()[i].read_len > maxLen) maxLen = b.jobs()[i].read_len;
        const int R = (maxLen + 63) / 64;
        L_TRY(hipLaunchKernel(bbmsa::fast_kernel_mat_for(R, c->banded), dim3((unsigned)blocks), dim3(64), args, (size_t)S->ldsBytes, st));
        L_TRY(hipMemcpyAsync(b.h_counters, b.d_counters, 64, hipMemcpyDeviceToHost, st));
        L_TRY(hipMemcpyAsync(b.h_slow, b.d_slow, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        { const int rc = download(); if (rc != BBMAP_OK) return rc; }
        L_TRY(hipStreamSynchronize(st));
        nslow = (int)b.h_counters[1];
        S->nsWave += since(t0);
        if (nslow == 0) return BBMAP_OK;             // the usual case: one launch, one synchronisation
    } else {
        for (int i = 0; i < n; i++) b.h_slow[i] = i;
    }
    // hand-overs: the generic kernel, one fill at a time into scratch slot 0, then the touched rectangle into the staging layout
    const auto t1 = std::chrono::steady_clock::now();
    const long long fullPlane = (long long)(c->cfg.maxRows + 1) * (c->cfg.maxColumns + 2);
    for (int k = 0; k < nslow; k++) {
        const int j = b.h_slow[k];
        const bbmsa_job &jb = b.jobs()[j];
        const int rows = jb.read_len, columns = jb.refEndLoc - jb.refStartLoc + 1;
        L_TRY(hipMemsetAsync(b.d_counters + 2, 0, 4, st));
        if (c->scheme == BBMSA_SCHEME_11TS) {   // the scratch matrix keeps cells of earlier fills (as the reference's `packed` does); cells this fill does not visit
            // are handed out as subfloor, like the wavefront kernel's, so that a fill's planes do not depend on what ran before it
            const long long maxGain = (long long)(rows - 1) * bbmsa::P_MATCH2 + bbmsa::P_MATCH;
            const bool lim = (jb.flags & BBMSA_MODE_MASK) == BBMSA_FILL_LIMITED_RAW;
            const long long subfloor = lim ? (long long)jb.minScore * 2048 - maxGain - 5LL * bbmsa::P_MATCH2 : -2 * maxGain;
            for (int s = 0; s < 3; s++)
                L_TRY(hipMemsetD32Async((hipDeviceptr_t)(c->d_matrix + (long long)s * fullPlane), (int)subfloor,
                                        (size_t)(rows + 1) * (size_t)(columns + 2), st));
        }
        bbmsa::GenericParams gp;
        gp.jobs = d_jobs + j; gp.reads = d_bytes; gp.refs = d_bytes; gp.results = b.d_results + j; gp.match = nullptr;
        gp.list = nullptr; gp.list_count = nullptr; gp.njobs = 1; gp.njobs_dev = nullptr;
        gp.matrix = c->d_matrix; gp.limits = c->d_limits; gp.queue = b.d_counters + 2;
        gp.match_stride = 0; gp.maxRows = c->cfg.maxRows; gp.maxColumns = c->cfg.maxColumns;
        gp.bandwidth = c->cfg.bandwidth; gp.bandwidthRatio = c->cfg.bandwidthRatio;
        if (c->scheme == BBMSA_SCHEME_9PACBIO)
            hipLaunchKernelGGL(bbmsa::msa_fill_generic_kernel<bbmsa::Scheme9PacBio>, dim3(1), dim3(1), 0, st, gp);
        else
            hipLaunchKernelGGL(bbmsa::msa_fill_generic_kernel<bbmsa::Scheme11ts>, dim3(1), dim3(1), 0, st, gp);
        L_TRY(hipGetLastError());
        const size_t W = (size_t)columns + 2;                       // the generic kernel's row stride for this fill
        for (int s = 0; s < 3; s++)
            L_TRY(hipMemcpy2DAsync(b.d_planes + h_planeOff[j] + (long long)s * rows * columns, (size_t)columns * 4,
                                   c->d_matrix + (long long)s * fullPlane + W + 1, W * 4, (size_t)columns * 4, (size_t)rows,
                                   hipMemcpyDeviceToDevice, st));
        L_TRY(hipMemcpyAsync(b.d_limits + h_limitsOff[j], c->d_limits, (size_t)(rows + 1) * 4, hipMemcpyDeviceToDevice, st));
        L_TRY(hipMemcpyAsync(b.d_limits + h_limitsOff[j] + rows + 1, c->d_limits + c->cfg.maxRows + 2, (size_t)(columns + 1) * 4,
                             hipMemcpyDeviceToDevice, st));
    }
    S->handed += nslow;
    { const int rc = download(); if (rc != BBMAP_OK) return rc; }
    L_TRY(hipStreamSynchronize(st));
    S->nsHanded += since(t1);
    return BBMAP_OK;
}

// lock held, leaderActive already set by the caller: closes the open batch, runs it, publishes it
void lead(bbmsa_ctx *c, std::unique_lock<std::mutex> &lk) {
    bbmsa_legacy *S = c->legacy;
    // the other buffer becomes the open batch: wait until its last reader has collected (callers keep joining ours meanwhile)
    Batch &other = S->B[S->open ^ 1];
    {
        const auto t0 = std::chrono::steady_clock::now();
        S->cv.wait(lk, [&] { return other.state != DONE || other.readers == 0; });
        S->nsWaitReaders += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    Batch &b = S->B[S->open];
    b.state = RUNNING;
    other.state = OPEN; other.njobs = 0; other.usedBytes = other.usedInts = other.usedLimits = 0; other.gen = ++S->nextGen; other.rc = BBMAP_OK;
    S->open ^= 1;
    S->cv.notify_all();                                    // callers that found the batch full can join the new one
    lk.unlock();
    const int rc = run_batch(c, b);
    lk.lock();
    b.rc = rc;
    if (rc != BBMAP_OK) snprintf(b.err, sizeof b.err, "legacy fill batch failed on the device (%d)", rc);
    b.readers = b.njobs;
    b.state = DONE;
    S->launches++;
    S->leaderActive = false;
    S->cv.notify_all();
}

}  // namespace

int bbmsa_legacy_create(bbmsa_ctx *c) {
    bbmsa_legacy *S = new (std::nothrow) bbmsa_legacy();
    if (!S) return lfail(BBMAP_E_NOMEM, "bbmsa_create: out of host memory");
    c->legacy = S;
    S->open = 0; S->leaderActive = false; S->nextGen = 1; S->calls = S->launches = S->handed = 0; S->nsWave = S->nsHanded = S->nsWaitReaders = 0;
    S->stream = nullptr; S->d_dir = nullptr;
    memset(S->B, 0, sizeof S->B);
    const int maxRows = c->cfg.maxRows, maxCols = c->cfg.maxColumns;
    const long long oneJobInts = 3LL * maxRows * maxCols;
    long long poolInts = ((long long)env_int("BBMSA_LEGACY_POOL_MB", 64) << 20) / 4;
    if (poolInts < oneJobInts) poolInts = oneJobInts;             // the largest fill the context admits always fits
    S->capInts = poolInts;
    S->capJobs = env_int("BBMSA_LEGACY_BATCH", 1024);
    if (S->capJobs < 1) S->capJobs = 1;
    S->capBytes = (long long)S->capJobs * 1024;
    if (S->capBytes < maxRows + maxCols + 16) S->capBytes = maxRows + maxCols + 16;
    S->capLimits = (long long)S->capJobs * 512;
    if (S->capLimits < maxRows + maxCols + 8) S->capLimits = maxRows + maxCols + 8;
    S->offPlaneOff = (long long)S->capJobs * sizeof(bbmsa_job);
    S->offLimitsOff = S->offPlaneOff + (long long)S->capJobs * 8;
    S->offBytes = S->offLimitsOff + (long long)S->capJobs * 8;
    L_TRY(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) { const int rc = alloc_batch(S, S->B[i]); if (rc != BBMAP_OK) return rc; }
    S->B[0].gen = S->nextGen;
    // geometry of the matrix-materialising wavefront launch
    S->wave = false;
    if (c->scheme == BBMSA_SCHEME_11TS) {
        S->R = (maxRows + 63) / 64;
        S->cols = maxCols;
        S->tmpBytes = ((64 * S->R + S->cols + 8) + 3) & ~3;
        const int side = (maxCols > maxRows ? maxCols : maxRows) + 2;
        S->tableLen = (side < 2048 ? side : 2048) + maxRows + 8;
        if (S->tableLen > bbmsa::kTableLen) S->tableLen = bbmsa::kTableLen;
        S->tableLen = (S->tableLen + 3) & ~3;
        const int perJob = bbmsa::lds_job_ints(S->cols, S->tmpBytes);
        S->ldsBytes = (bbmsa::lds_table_ints(S->tableLen) + perJob) * 4;
        c->banded = !(c->cfg.bandwidth < 1 && c->cfg.bandwidthRatio <= 0.0f);
        const void *kfn = bbmsa::fast_kernel_mat_for(S->R, c->banded);
        if (kfn && S->ldsBytes <= 160 * 1024) {
            if (S->ldsBytes > 64 * 1024)
                for (int r = 1; r <= S->R; r++)
                    L_TRY(hipFuncSetAttribute(bbmsa::fast_kernel_mat_for(r, c->banded), hipFuncAttributeMaxDynamicSharedMemorySize, S->ldsBytes));
            int per = 0;
            L_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kfn, 64, S->ldsBytes));
            if (per < 1) per = 1;
            if (per > 8) per = 8;
            S->blocks = c->numCUs * per;
            if (S->blocks > S->capJobs) S->blocks = S->capJobs;
            S->dirSlotDwords = (long long)(((S->cols + 64 - 1) >> 3) + 1) * S->R * 64;
            L_TRY(hipMalloc((void **)&S->d_dir, (size_t)((long long)S->blocks * S->dirSlotDwords * 4)));
            S->wave = true;
        }
    }
    return BBMAP_OK;
}

void bbmsa_legacy_destroy(bbmsa_ctx *c) {
    bbmsa_legacy *S = c->legacy;
    if (!S) return;
    if (S->stream) { (void)hipStreamSynchronize(S->stream); (void)hipStreamDestroy(S->stream); }
    for (int i = 0; i < 2; i++) free_batch(S->B[i]);
    if (S->d_dir) (void)hipFree(S->d_dir);
    delete S;
    c->legacy = nullptr;
}

extern "C" int bbmsa_fill_submit(bbmsa_ctx *c, const uint8_t *read, int32_t read_len, const uint8_t *ref, int32_t ref_len,
                                 int32_t refStartLoc, int32_t refEndLoc, int32_t minScore, int32_t mode,
                                 int32_t *result5, int64_t *iterations, bbmsa_ticket *ticket) {
    if (!c || !read || !ref || !result5 || !ticket) return lfail(BBMAP_E_ARG, "bbmsa_fill_submit: null argument");
    if (!c->legacy) return lfail(BBMAP_E_ARG, "bbmsa_fill_submit: the context was not created with BBMSA_LEGACY_ONLY");
    if (mode != BBMSA_FILL_LIMITED_RAW && mode != BBMSA_FILL_UNLIMITED_RAW) return lfail(BBMAP_E_ARG, "bbmsa_fill_submit: mode must be one of the two raw fills");
    const int rows = read_len;
    const long long columns = (long long)refEndLoc - refStartLoc + 1;
    if (rows < 1 || columns < 1 || rows > c->cfg.maxRows || columns > c->cfg.maxColumns || refStartLoc < 0 || refEndLoc >= ref_len)
        return lfail(BBMAP_E_SHAPE, "bbmsa_fill_submit: problem exceeds the context limits or its reference array");
    bbmsa_legacy *S = c->legacy;
    const long long needBytes = ((long long)rows + columns + 7) & ~7LL;
    const long long needInts = 3LL * rows * columns;
    const long long needLimits = rows + columns + 2;
    std::unique_lock<std::mutex> lk(S->mu);
    S->calls++;
    Batch *b;
    for (;;) {
        b = &S->B[S->open];
        if (b->njobs < S->capJobs && b->usedBytes + needBytes <= S->capBytes && b->usedInts + needInts <= S->capInts &&
            b->usedLimits + needLimits <= S->capLimits) break;
        // full: make sure somebody is taking it to the device, then wait for the next one to open
        if (!S->leaderActive) { S->leaderActive = true; lead(c, lk); }
        else S->cv.wait(lk);
    }
    const int slot = b->njobs++;
    const long long myGen = b->gen;
    uint8_t *bytes = b->h_in + S->offBytes + b->usedBytes;
    memcpy(bytes, read, (size_t)rows);
    memcpy(bytes + rows, ref + refStartLoc, (size_t)columns);          // only the window leaves the caller's array; the job is rebased
    bbmsa_job &jb = b->jobs()[slot];
    jb.read_off = b->usedBytes; jb.ref_off = b->usedBytes + rows; jb.read_len = rows; jb.ref_len = (int32_t)columns;
    jb.refStartLoc = 0; jb.refEndLoc = (int32_t)columns - 1; jb.minScore = minScore; jb.flags = mode;
    ((long long *)(b->h_in + S->offPlaneOff))[slot] = b->usedInts;
    ((long long *)(b->h_in + S->offLimitsOff))[slot] = b->usedLimits;
    b->usedBytes += needBytes; b->usedInts += needInts; b->usedLimits += needLimits;
    // wait for the batch to come back; whoever finds no leader while waiting becomes one
    while (!(b->state == DONE && b->gen == myGen)) {
        if (!S->leaderActive && b->state == OPEN) { S->leaderActive = true; lead(c, lk); }
        else S->cv.wait(lk);
    }
    ticket->batch = (int32_t)(b - S->B); ticket->slot = slot; ticket->gen = myGen;
    ticket->rows = rows; ticket->columns = (int32_t)columns;
    if (b->rc != BBMAP_OK) {
        const int rc = b->rc;
        bbmap_set_error(b->err);
        if (--b->readers == 0) S->cv.notify_all();
        ticket->gen = -1;
        return rc;
    }
    const bbmsa_result &res = b->h_results[slot];
    for (int i = 0; i < 5; i++) result5[i] = res.result[i];
    if (iterations) *iterations += res.iterations;                    // the native code increments, jni/...c:471,:746
    return BBMAP_OK;
}

extern "C" int bbmsa_fill_collect(bbmsa_ctx *c, bbmsa_ticket *ticket, int32_t *packed, int32_t *vertLimit, int32_t *horizLimit) {
    if (!c || !c->legacy || !ticket) return lfail(BBMAP_E_ARG, "bbmsa_fill_collect: null argument");
    if (ticket->gen < 0 || ticket->batch < 0 || ticket->batch > 1) return lfail(BBMAP_E_ARG, "bbmsa_fill_collect: the ticket holds no finished fill");
    bbmsa_legacy *S = c->legacy;
    Batch &b = S->B[ticket->batch];
    // no lock needed to read: the batch cannot be reopened while this reader is counted
    const int rows = ticket->rows, columns = ticket->columns, slot = ticket->slot;
    const long long planeOff = ((const long long *)(b.h_in + S->offPlaneOff))[slot];
    const long long limOff = ((const long long *)(b.h_in + S->offLimitsOff))[slot];
    if (packed) {
        // the Java layout: 3 x (maxRows + 1) x (maxColumns + 1) ints, state-major (MultiStateAligner11tsJNI.java:71-113); row 0 and
        // column 0 belong to the constructor and are never touched by a fill
        const size_t rowInts = (size_t)c->cfg.maxColumns + 1, plane = (size_t)(c->cfg.maxRows + 1) * rowInts;
        for (int s = 0; s < 3; s++) {
            const int *src = b.h_planes + planeOff + (long long)s * rows * columns;
            int32_t *dst = packed + (size_t)s * plane;
            for (int r = 1; r <= rows; r++) memcpy(dst + (size_t)r * rowInts + 1, src + (size_t)(r - 1) * columns, (size_t)columns * 4);
        }
    }
    const bool limited = (b.jobs()[slot].flags & BBMSA_MODE_MASK) == BBMSA_FILL_LIMITED_RAW;
    if (limited && vertLimit) memcpy(vertLimit, b.h_limits + limOff, (size_t)(rows + 1) * 4);
    if (limited && horizLimit) memcpy(horizLimit, b.h_limits + limOff + rows + 1, (size_t)(columns + 1) * 4);
    ticket->gen = -1;
    {
        std::lock_guard<std::mutex> g(S->mu);
        if (--b.readers == 0) S->cv.notify_all();
    }
    return BBMAP_OK;
}

// the two halves together, for callers that own `packed` outright (ctypes, tests)
extern "C" int bbmsa_fill_packed(bbmsa_ctx *c, const uint8_t *read, int32_t read_len, const uint8_t *ref, int32_t ref_len,
                                 int32_t refStartLoc, int32_t refEndLoc, int32_t minScore, int32_t mode,
                                 int32_t *result5, int64_t *iterations, int32_t *packed) {
    if (!packed) return lfail(BBMAP_E_ARG, "bbmsa_fill_packed: null argument");
    bbmsa_ticket t;
    const int rc = bbmsa_fill_submit(c, read, read_len, ref, ref_len, refStartLoc, refEndLoc, minScore, mode, result5, iterations, &t);
    if (rc != BBMAP_OK) return rc;
    return bbmsa_fill_collect(c, &t, packed, nullptr, nullptr);
}

extern "C" int bbmsa_legacy_stats(bbmsa_ctx *c, int64_t *stats6) {
    if (!c || !c->legacy || !stats6) return lfail(BBMAP_E_ARG, "bbmsa_legacy_stats: null argument");
    std::lock_guard<std::mutex> g(c->legacy->mu);
    stats6[0] = c->legacy->calls; stats6[1] = c->legacy->launches; stats6[2] = c->legacy->handed;
    stats6[3] = c->legacy->nsWave; stats6[4] = c->legacy->nsHanded; stats6[5] = c->legacy->nsWaitReaders;
    return BBMAP_OK;
}
