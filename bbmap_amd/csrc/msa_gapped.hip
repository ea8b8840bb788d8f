// Gapped reference windows for the MultiStateAligner11ts entry points.
//
// A SiteScore whose index hit spans a long deletion carries a gap array {start, stop, start, stop, ...}; the
// reference then aligns against a "gapped reference" in which every long gap is shortened to
// GAPBUFFER + (gap % GAPLEN) bases, (gap - GAPBUFFER2) / GAPLEN gap symbols '-' (each standing for GAPLEN = 128
// bases) and GAPBUFFER more bases (MultiStateAligner11tsJNI.makeGref, current/align2/MultiStateAligner11tsJNI.java:
// 668-757), fills that buffer from column 0 to greflimit (fillLimited(..., gaps) :116-128), and translates the
// resulting coordinates back (:759-801).  Here: one small kernel builds the gapped references of a batch into a
// scratch buffer in HBM and derives ordinary jobs that point at them (a job's ref_off is relative to `refs`, so it
// can address the scratch buffer), the normal fill kernels run, and a second small kernel translates score[1..2].
// Gapped sites are rare (a few per thousand reads), so neither helper kernel is tuned.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstring>

#include "msa_common.h"
#include "msa_ctx.h"

void bbmap_set_error(const char *msg);

namespace bbmsa {

constexpr int K_GAPBUFFER = 64, K_GAPBUFFER2 = 128, K_GREF_CUSHION = 128;
constexpr uint8_t K_GAPC = '-';

struct GappedParams {
    const bbmsa_job *jobs;
    const bbmsa_gaps *gaps;
    const uint8_t *refs;
    bbmsa_job *out_jobs;
    uint8_t *gref;            // n x glen bytes
    int *aux;                 // n x 4: {origin, greflimit2, status (0 ok, 2 bad shape), ngaps}
    long long njobs;
    const unsigned int *njobs_dev;
    int glen;                 // maxColumns + 2 (the reference's grefbuffer length, MSA.java:77)
    int maxColumns;
};

// MSA.fillAndScoreLimited's gapped branch (:104-105,:125-131) + makeGref (:668-757): one wavefront per job.  The segment
// bookkeeping is a few scalar steps per gap; the byte copies are done by all 64 lanes, coalesced.
__global__ __launch_bounds__(256) void make_gref_kernel(const GappedParams P) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= job_count(P.njobs, P.njobs_dev)) return;
    bbmsa_job jb = P.jobs[j];
    int *aux = P.aux + 4 * j;
    const int ngaps = P.gaps[j].ngaps;
    if (ngaps <= 0) { if (lane == 0) { aux[0] = 0; aux[1] = 0; aux[2] = 0; aux[3] = ngaps; P.out_jobs[j] = jb; } return; }
    const uint8_t *ref = P.refs + jb.ref_off;
    const int a = max(0, jb.refStartLoc), b = min(jb.ref_len - 1, jb.refEndLoc);
    int g[BBMSA_MAX_GAPS];
    for (int i = 0; i < BBMSA_MAX_GAPS; i++) g[i] = P.gaps[j].gaps[i];
    bool bad = ngaps < 2 || ngaps > BBMSA_MAX_GAPS || (ngaps & 1) || b < a;
    int gpos = 0, origin = 0, greflimit2 = -1;
    uint8_t *gref = P.gref + j * (long long)P.glen;
    if (!bad) {
        g[0] = min(g[0], a);
        g[ngaps - 1] = max(g[ngaps - 1], b);
        origin = g[0];
        // every byte the reference would read must exist (Java would throw on a negative or too large index)
        for (int i = 0; i < ngaps && !bad; i++) if (g[i] < 0 || g[i] >= jb.ref_len) bad = true;
        for (int i = 0; i + 1 < ngaps && !bad; i++) if (g[i + 1] < g[i]) bad = true;
    }
    // copy ref[from .. from+n) to gref[gpos ..), all lanes; fails (bad) when it would run past the buffer
    auto copy = [&](int from, int n) {
        if (n <= 0 || bad) return;
        if (gpos + n > P.glen) { bad = true; return; }
        for (int i = lane; i < n; i += 64) gref[gpos + i] = ref[from + i];
        gpos += n;
    };
    for (int i = 0; i < ngaps && !bad; i += 2) {
        const int x = g[i], y = g[i + 1];
        copy(x, y - x + 1);
        if (i + 2 < ngaps && !bad) {
            const int z = g[i + 2];
            const int gap = z - y - 1;
            if (gap < K_GAPBUFFER2) { bad = true; break; }                        // the reference asserts gap >= MINGAP
            const int rem = gap % kGapLen;
            const int div = (gap - K_GAPBUFFER2) / kGapLen;
            copy(y + 1, K_GAPBUFFER + rem);
            if (!bad) {
                if (gpos + div > P.glen) bad = true;
                else { for (int q = lane; q < div; q += 64) gref[gpos + q] = K_GAPC; gpos += div; }
            }
            copy(z - K_GAPBUFFER, K_GAPBUFFER);
        }
    }
    const int greflimit = gpos;
    if (!bad) {
        const int lim = min(P.glen, greflimit + K_GREF_CUSHION);
        for (int i = greflimit + lane; i < lim; i += 64) { const int r = b + 1 + (i - greflimit); gref[i] = (r < jb.ref_len) ? ref[r] : (uint8_t)'N'; }
        if (lim > greflimit) greflimit2 = lim - 1;
        // translateToGappedCoordinate(a) must be 0 (asserted by the reference, :514): a <= origin always holds here
        // because origin = min(gaps[0], a); the fill covers columns 0..greflimit, which needs that byte to exist
        if (greflimit >= P.glen || greflimit + 1 > P.maxColumns) bad = true;
    }
    bbmsa_job o = jb;
    if (bad) {
        // an impossible shape: the fill kernels answer BBMSA_ST_BAD_SHAPE for it
        o.read_len = 0; o.refStartLoc = 0; o.refEndLoc = -1; o.flags = BBMSA_FILL_LIMITED_RAW;
    } else {
        // translateToGappedCoordinate(b) (:781-801): index of the gapped reference whose original coordinate is b.  The gapped
        // reference was just written by this wave: make the stores visible to its own loads first.
        __threadfence_block();
        int gstop = INT_MIN;
        if (b <= origin) gstop = b - origin;
        else {
            // coordinate of index i = origin + i + (kGapLen - 1) * (gap symbols before i): scan 64 indices per step
            int carry = 0;                                   // gap symbols before the current chunk
            for (int base = 0; base < greflimit2 && gstop == INT_MIN; base += 64) {
                const int i = base + lane;
                const bool in = i < greflimit2;
                const bool isGap = in && gref[i] == K_GAPC;
                const unsigned long long gm = __ballot(isGap);
                const int before = carry + __builtin_popcountll(gm & ((1ull << lane) - 1ull));
                const int q = origin + i + (kGapLen - 1) * before;
                const unsigned long long hit = __ballot(in && q == b);
                if (hit) gstop = base + __builtin_ctzll(hit);
                carry += __builtin_popcountll(gm);
            }
        }
        o.ref_off = (int64_t)(gref - P.refs);
        o.ref_len = gstop;                                   // parked for the score2 tail (BBMSA_INTERNAL_GAPPED)
        o.refStartLoc = 0; o.refEndLoc = greflimit;
        // fillLimited(read, ref, a, b, minScore, gaps) (:116-128), or fillUnlimited(read, ref, a, b, gaps) (:166-176: realign_new's last resort)
        o.flags = ((jb.flags & BBMSA_MODE_MASK) == BBMSA_FILL_UNLIMITED_RAW ? BBMSA_FILL_UNLIMITED_RAW : BBMSA_FILL_LIMITED) | BBMSA_DO_SCORE |
                  (jb.flags & (BBMSA_DO_TRACEBACK | BBMSA_TRACE_KEEP_GAPS)) | BBMSA_INTERNAL_GAPPED;
    }
    if (lane == 0) { aux[0] = origin; aux[1] = greflimit2; aux[2] = bad ? 2 : 0; aux[3] = ngaps; P.out_jobs[j] = o; }
}

// translateFromGappedCoordinate (:759-779) on score[1], score[2]: one wavefront per job
__global__ __launch_bounds__(256) void gref_post_kernel(const GappedParams P, bbmsa_result *results) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= job_count(P.njobs, P.njobs_dev)) return;
    const int *aux = P.aux + 4 * j;
    if (aux[3] <= 0 || aux[2] != 0) return;
    bbmsa_result &r = results[j];
    if (r.score_len <= 0) return;
    const uint8_t *gref = P.gref + j * (long long)P.glen;
    const int origin = aux[0], greflimit2 = aux[1];
    int outv[3] = {0, INT_MIN, INT_MIN};
    for (int w = 1; w <= 2; w++) {
        const int point = r.score[w];
        if (point <= 0) { outv[w] = origin + point; continue; }
        if (point >= greflimit2) continue;                   // the reference's loop ends without a match: INT_MIN
        // coordinate of index `point` = origin + point + (kGapLen - 1) * (gap symbols before it)
        int gaps = 0;
        for (int base = 0; base < point; base += 64) {
            const int i = base + lane;
            gaps += __builtin_popcountll(__ballot(i < point && gref[i] == K_GAPC));
        }
        outv[w] = origin + point + (kGapLen - 1) * gaps;
    }
    if (lane == 0) { r.score[1] = outv[1]; r.score[2] = outv[2]; }
}

}  // namespace bbmsa

static thread_local char g_gerr[256];
#define GHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_gerr, sizeof g_gerr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_gerr); return BBMAP_E_HIP; } } while (0)

int bbmsa_align_impl(bbmsa_ctx *c, void *stream_, int64_t n_jobs, const uint32_t *n_jobs_dev, const bbmsa_job *jobs,
                     const uint8_t *reads, const uint8_t *refs, bbmsa_result *results, uint8_t *match, int32_t match_stride);   // msa_host.hip

static int gapped_impl(bbmsa_ctx *c, void *stream_, int64_t n_jobs, const uint32_t *n_jobs_dev, const bbmsa_job *jobs,
                       const bbmsa_gaps *gaps, const uint8_t *reads, const uint8_t *refs,
                       bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!c) { bbmap_set_error("bbmsa_align_gapped_batch_device: null context"); return BBMAP_E_ARG; }
    if (n_jobs < 0 || n_jobs > 0x7fffffffLL) { bbmap_set_error("bbmsa_align_gapped_batch_device: n_jobs out of range"); return BBMAP_E_ARG; }
    if (n_jobs == 0) return BBMAP_OK;
    if (!jobs || !gaps || !reads || !refs || !results) { bbmap_set_error("bbmsa_align_gapped_batch_device: null buffer"); return BBMAP_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    GHIP(hipSetDevice(c->device));
    const int glen = c->cfg.maxColumns + 2;
    if (n_jobs > c->gappedCap) {
        GHIP(hipStreamSynchronize(stream));
        if (c->d_gref) { (void)hipFree(c->d_gref); c->d_gref = nullptr; }
        if (c->d_gaux) { (void)hipFree(c->d_gaux); c->d_gaux = nullptr; }
        if (c->d_gjobs) { (void)hipFree(c->d_gjobs); c->d_gjobs = nullptr; }
        c->gappedCap = 0;
        GHIP(hipMalloc(&c->d_gref, (size_t)n_jobs * (size_t)glen));
        GHIP(hipMalloc(&c->d_gaux, (size_t)n_jobs * 16));
        GHIP(hipMalloc(&c->d_gjobs, (size_t)n_jobs * sizeof(bbmsa_job)));
        c->gappedCap = n_jobs;
    }
    bbmsa::GappedParams P;
    P.jobs = jobs; P.gaps = gaps; P.refs = refs; P.out_jobs = c->d_gjobs; P.gref = c->d_gref; P.aux = c->d_gaux;
    P.njobs = n_jobs; P.njobs_dev = n_jobs_dev; P.glen = glen; P.maxColumns = c->cfg.maxColumns;
    const unsigned blocks = (unsigned)((n_jobs + 3) / 4);                       // one wavefront per job, 4 per block
    hipLaunchKernelGGL(bbmsa::make_gref_kernel, dim3(blocks), dim3(256), 0, stream, P);
    GHIP(hipGetLastError());
    const int rc = bbmsa_align_impl(c, stream_, n_jobs, n_jobs_dev, c->d_gjobs, reads, refs, results, match, match_stride);
    if (rc != BBMAP_OK) return rc;
    hipLaunchKernelGGL(bbmsa::gref_post_kernel, dim3(blocks), dim3(256), 0, stream, P, results);
    GHIP(hipGetLastError());
    return BBMAP_OK;
}

extern "C" int bbmsa_align_gapped_batch_device(bbmsa_ctx *c, void *stream_, int64_t n_jobs, const bbmsa_job *jobs,
                                               const bbmsa_gaps *gaps, const uint8_t *reads, const uint8_t *refs,
                                               bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    return gapped_impl(c, stream_, n_jobs, nullptr, jobs, gaps, reads, refs, results, match, match_stride);
}

extern "C" int bbmsa_align_gapped_batch_device_indirect(bbmsa_ctx *c, void *stream_, const uint32_t *n_jobs_dev, int64_t max_jobs,
                                                        const bbmsa_job *jobs, const bbmsa_gaps *gaps, const uint8_t *reads,
                                                        const uint8_t *refs, bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!n_jobs_dev) { bbmap_set_error("bbmsa_align_gapped_batch_device_indirect: null counter"); return BBMAP_E_ARG; }
    return gapped_impl(c, stream_, max_jobs, n_jobs_dev, jobs, gaps, reads, refs, results, match, match_stride);
}

extern "C" int bbmsa_align_gapped_batch(bbmsa_ctx *c, int64_t n_jobs, const bbmsa_job *jobs, const bbmsa_gaps *gaps,
                                        const uint8_t *reads, int64_t reads_bytes, const uint8_t *refs, int64_t refs_bytes,
                                        bbmsa_result *results, uint8_t *match, int32_t match_stride) {
    if (!c) { bbmap_set_error("bbmsa_align_gapped_batch: null context"); return BBMAP_E_ARG; }
    if (n_jobs == 0) return BBMAP_OK;
    if (n_jobs < 0 || !jobs || !gaps || !reads || !refs || !results || reads_bytes < 0 || refs_bytes < 0) {
        bbmap_set_error("bbmsa_align_gapped_batch: bad argument"); return BBMAP_E_ARG;
    }
    for (int64_t i = 0; i < n_jobs; i++) {
        const bbmsa_job &j = jobs[i];
        if (j.read_len < 0 || j.read_off < 0 || j.read_off + j.read_len > reads_bytes ||
            j.ref_len < 0 || j.ref_off < 0 || j.ref_off + j.ref_len > refs_bytes) {
            bbmap_set_error("bbmsa_align_gapped_batch: a job lies outside its buffers"); return BBMAP_E_ARG;
        }
        if (gaps[i].ngaps <= 0 && !(j.flags & BBMSA_CLAMP_WINDOW) && (j.refStartLoc < 0 || j.refEndLoc >= j.ref_len)) {
            bbmap_set_error("bbmsa_align_gapped_batch: window outside its reference array"); return BBMAP_E_ARG;
        }
    }
    GHIP(hipSetDevice(c->device));
    bbmsa_job *d_jobs = nullptr; bbmsa_gaps *d_gaps = nullptr; uint8_t *d_reads = nullptr, *d_refs = nullptr, *d_match = nullptr; bbmsa_result *d_res = nullptr;
    int rc = BBMAP_OK;
#define GGO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_gerr, sizeof g_gerr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_gerr); rc = BBMAP_E_HIP; goto done; } } while (0)
    GGO(hipMalloc(&d_jobs, (size_t)n_jobs * sizeof(bbmsa_job)));
    GGO(hipMalloc(&d_gaps, (size_t)n_jobs * sizeof(bbmsa_gaps)));
    GGO(hipMalloc(&d_reads, (size_t)(reads_bytes > 0 ? reads_bytes : 1)));
    GGO(hipMalloc(&d_refs, (size_t)(refs_bytes > 0 ? refs_bytes : 1)));
    GGO(hipMalloc(&d_res, (size_t)n_jobs * sizeof(bbmsa_result)));
    if (match) GGO(hipMalloc(&d_match, (size_t)n_jobs * (size_t)match_stride));
    GGO(hipMemcpy(d_jobs, jobs, (size_t)n_jobs * sizeof(bbmsa_job), hipMemcpyHostToDevice));
    GGO(hipMemcpy(d_gaps, gaps, (size_t)n_jobs * sizeof(bbmsa_gaps), hipMemcpyHostToDevice));
    GGO(hipMemcpy(d_reads, reads, (size_t)reads_bytes, hipMemcpyHostToDevice));
    GGO(hipMemcpy(d_refs, refs, (size_t)refs_bytes, hipMemcpyHostToDevice));
    GGO(hipMemset(d_res, 0xff, (size_t)n_jobs * sizeof(bbmsa_result)));
    rc = bbmsa_align_gapped_batch_device(c, nullptr, n_jobs, d_jobs, d_gaps, d_reads, d_refs, d_res, d_match, match_stride);
    if (rc != BBMAP_OK) goto done;
    GGO(hipStreamSynchronize(nullptr));
    GGO(hipMemcpy(results, d_res, (size_t)n_jobs * sizeof(bbmsa_result), hipMemcpyDeviceToHost));
    if (match) GGO(hipMemcpy(match, d_match, (size_t)n_jobs * (size_t)match_stride, hipMemcpyDeviceToHost));
done:
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_gaps) (void)hipFree(d_gaps);
    if (d_reads) (void)hipFree(d_reads);
    if (d_refs) (void)hipFree(d_refs);
    if (d_res) (void)hipFree(d_res);
    if (d_match) (void)hipFree(d_match);
    return rc;
#undef GGO
}
