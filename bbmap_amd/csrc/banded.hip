// BandedAligner on gfx950: unit-cost edit distance inside a diagonal band, two rolling rows.
//
// One alignment per G-lane group of a wave (G = 16/32/64, chosen from the band width); the lanes
// of a group are the cells of the band row.  A row's left-to-right dependency
// (score = min(up+1, diag+mis, left+1)) is a min-plus prefix scan: with base = min(up+1, diag+mis),
// score[i] = min_j<=i (base[j] + i - j), computed as a DPP/shuffle prefix-min of (base[j] - j).
// The only cell that ignores its left neighbour (the forced-diagonal column at the end of the
// reference) is always the last cell of the row in iteration order, so it is patched afterwards.
// Rolling rows live in LDS; bands wider than G are walked in G-wide chunks with a carry.
//
// Semantics follow jni/BandedAlignerJNI.c:97-585 (variant 0) or
// current/align2/BandedAlignerConcrete.java:100-551 + BandedAligner.java:98-147 (variant 1);
// see oracle/banded_oracle.c for the list of differences.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "bbmap_amd.h"

namespace bbband {

struct Params {
    const bbband_job *jobs;
    const uint8_t *seqs;
    bbband_result *results;
    long long njobs;
    unsigned int *queue;
    int maxWidth;      // max(width,3)|1
    int variant;
    int G;
};

// dna/AminoAcid.java:110-133,:633-645 (baseToComplementExtended); 0xFF = unmapped (-1 in the reference)
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
__device__ inline bool defined_base(int b) {
    const int u = b & ~32;
    return b < 128 && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U');
}

__device__ inline int group_min(int v, int G) {
    for (int d = 1; d < G; d <<= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ inline long long group_min64(long long v, int G) {
    for (int d = 1; d < G; d <<= 1) { const long long o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}

__global__ __launch_bounds__(256) void banded_kernel(const Params p) {
    extern __shared__ int lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = p.G, gl = lane & (G - 1), sub = lane / G, groupsPerWave = 64 / G;
    const int rowInts = p.maxWidth + 2;
    int *rowA = lds + ((wave * groupsPerWave + sub) * 2) * rowInts;
    int *rowB = rowA + rowInts;
    const int big = p.variant ? 99999999 : 999;

    for (;;) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(p.queue, (unsigned)groupsPerWave);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((long long)base >= p.njobs) break;
        const long long j = (long long)base + sub;
        const bool valid = j < p.njobs;
        bbband_job jb;
        if (valid) jb = p.jobs[j];
        else { jb.query_off = jb.ref_off = 0; jb.query_len = jb.ref_len = 0; jb.qstart = jb.rstart = 0; jb.maxEdits = 0; jb.flags = 0; }

        // the reference's swap rules (jni/BandedAlignerJNI.c:141-148,:260-267,:375-382,:491-498)
        int dir = jb.flags & BBBAND_DIR_MASK;
        const bool exact = (jb.flags & BBBAND_EXACT) != 0;
        const uint8_t *query = p.seqs + jb.query_off, *ref = p.seqs + jb.ref_off;
        int qlen = jb.query_len, rlen = jb.ref_len, qstart = jb.qstart, rstart = jb.rstart;
        bool swapped;
        switch (dir) {
            case 0: swapped = qlen - qstart > rlen - rstart; break;
            case 1: swapped = qstart + 1 > rlen - rstart; if (swapped) dir = 3; break;
            case 2: swapped = qstart > rstart; break;
            default: swapped = qlen - qstart > rstart + 1; if (swapped) dir = 1; break;
        }
        if (swapped) {
            const uint8_t *tp = query; query = ref; ref = tp;
            int t = qlen; qlen = rlen; rlen = t;
            t = qstart; qstart = rstart; rstart = t;
        }
        const bool rc = (dir == 1 || dir == 3), fwdRef = (dir == 0 || dir == 1);
        const int qstep = (dir == 0 || dir == 3) ? 1 : -1, rstep = fwdRef ? 1 : -1;
        int width = min(p.maxWidth, jb.maxEdits * 2 + 1);
        if (p.variant) width = min(width, max(qlen, rlen) * 2 + 2) | 1;
        const int halfWidth = width / 2, center = halfWidth + 1;
        int xlines, ylines;
        switch (dir) {
            case 0: xlines = qlen - qstart; ylines = rlen - rstart; break;
            case 1: xlines = qstart + 1; ylines = rlen - rstart; break;
            case 2: xlines = qstart + 1; ylines = rstart + 1; break;
            default: xlines = qlen - qstart; ylines = rstart + 1; break;
        }
        const int len = min(xlines, ylines);
        // index sanity: the reference would read outside its arrays
        const bool shapeOK = valid && qlen >= 0 && rlen >= 0 && width >= 1 &&
                             (len < 1 || (qstart >= 0 && qstart < qlen && rstart >= -halfWidth - 1 && rstart <= rlen + halfWidth));
        const bool run = shapeOK && len >= 1;

        int *cur = rowA, *prev = rowB;
        for (int i = gl; i < rowInts; i += G) { cur[i] = big; prev[i] = big; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        int qloc = qstart, rsloc = rstart - halfWidth, edits = 0, row = 0;
        bool active = run;
        // first pass through the loop body is row 0; `active` drops when the row minimum exceeds maxEdits
        while (__any(active && row < len)) {
            const bool go = active && row < len;
            if (go && row > 0) {
                int *t = cur; cur = prev; prev = t;
                for (int i = gl; i < rowInts; i += G) cur[i] = big;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int rowMin = big;
            if (go) {
                int q = query[qloc];
                if (rc) q = complement_extended(q);
                const bool qdef = defined_base(q);
                const int colStart = max(0, rsloc), colLimit = min(rsloc + width, rlen);
                const int count = colLimit - colStart;
                const int mStart = fwdRef ? 1 + (colStart - rsloc) : 1 + width - (colLimit - rsloc);
                const bool forceDiag = (row == len - 1);
                const int forcedCol = fwdRef ? rlen - 1 : 0;
                int carry = big;                                   // cur[mStart-1]
                for (int cb = 0; cb < count; cb += G) {
                    const int i = cb + gl;
                    const bool in = i < count;
                    const int mloc = mStart + i;
                    const int col = fwdRef ? colStart + i : colLimit - 1 - i;
                    int mis = 0;
                    if (in) {
                        const int r = ref[col];
                        mis = (q == r || (!exact && (!qdef || !defined_base(r)))) ? 0 : 1;
                    }
                    int s;
                    if (row == 0) {
                        s = mis;
                    } else {
                        int diag = big, t = 0x3fffffff;
                        bool forced = false;
                        if (in) {
                            diag = prev[mloc] + mis;
                            t = min(prev[mloc + 1] + 1, diag) - i;
                            forced = forceDiag || col == forcedCol;
                        }
                        for (int d = 1; d < G; d <<= 1) {            // inclusive prefix-min within the group
                            const int o = __shfl_up(t, d, 64);
                            if (gl >= d) t = min(t, o);
                        }
                        s = min(t + i, carry + (i - cb + 1));
                        if (forced) s = diag;
                        const int lastLane = min(G - 1, count - cb - 1);
                        carry = __shfl(s, (lane & ~(G - 1)) + lastLane, 64);   // score of the chunk's last cell
                    }
                    if (in) cur[mloc] = s;
                    if (in) rowMin = min(rowMin, s);
                }
            }
            rowMin = group_min(rowMin, G);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (go) {
                edits = rowMin;
                if (row == 0) {
                    // penalizeOffCenter on the first row (jni/...c:196-198 / BandedAligner.java:131-147)
                    int e = big;
                    for (int m = 1 + gl; m <= width; m += G) {
                        const int i = m > center ? m - center : center - m;
                        int v = cur[m];
                        if (i > 0) v = p.variant ? min(big, max(i, v)) : min(big, v + i);
                        cur[m] = v;
                        e = min(e, v);
                    }
                    edits = group_min(e, G);
                    row++; qloc += qstep; rsloc += rstep;
                } else if (edits > jb.maxEdits) {
                    row++; active = false;                          // qloc / rsloc stay (jni/...c:222-225)
                } else {
                    row++; qloc += qstep; rsloc += rstep;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        bbband_result res;
        res.edits = 0; res.lastQueryLoc = 0; res.lastRefLoc = 0; res.lastRow = -1; res.lastEdits = 0; res.lastOffset = 0;
        res.status = shapeOK ? 0 : 2; res.reserved = 0;
        if (run) {
            // final penalizeOffCenter + lastOffset (first strict minimum in the order centre, +1, -1, +2, -2, ...)
            long long best = (long long)0x7fffffff << 32;
            int e = big;
            for (int m = 1 + gl; m <= width; m += G) {
                const int i = m > center ? m - center : center - m;
                int v = cur[m];
                if (i > 0) v = p.variant ? min(big, max(i, v)) : min(big, v + i);
                e = min(e, v);
                const int rank = (m == center) ? 0 : (m > center ? 2 * i - 1 : 2 * i);
                const long long key = ((long long)v << 32) | (unsigned)((rank << 12) | m);
                best = key < best ? key : best;
            }
            edits = group_min(e, G);
            best = group_min64(best, G);
            const int minLoc = (int)(best & 0xfff);
            const int lastOffset = center - minLoc;
            int lastQueryLoc = qloc - qstep, lastRefLoc;
            if (fwdRef) {
                lastRefLoc = rsloc + halfWidth - lastOffset - 1;
                if (dir == 0) { while (lastRefLoc >= rlen || lastQueryLoc >= qlen) { lastRefLoc--; lastQueryLoc--; } }
                else { while (lastRefLoc >= rlen || lastQueryLoc < 0) { lastRefLoc--; lastQueryLoc++; } }
            } else {
                lastRefLoc = rsloc + halfWidth + lastOffset + 1;
                if (dir == 2) { while (lastRefLoc < 0 || lastQueryLoc < 0) { lastRefLoc++; lastQueryLoc++; } }
                else { while (lastRefLoc < 0 || lastQueryLoc >= qlen) { lastRefLoc++; lastQueryLoc--; } }
            }
            res.edits = edits; res.lastRow = row - 1; res.lastEdits = edits; res.lastOffset = lastOffset;
            res.lastQueryLoc = swapped ? lastRefLoc : lastQueryLoc;
            res.lastRefLoc = swapped ? lastQueryLoc : lastRefLoc;
        }
        if (valid && gl == 0) p.results[j] = res;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace bbband

void bbmap_set_error(const char *msg);   // msa_host.hip
static thread_local char g_berr[256] = "";

struct bbband_ctx {
    int device, maxWidth, variant, G, blocks, ldsBytes;
    unsigned int *d_queue;
};

static int bfail(int code, const char *msg) { bbmap_set_error(msg); return code; }

#define BHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_berr, sizeof g_berr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_berr); return BBMAP_E_HIP; } } while (0)

extern "C" int bbband_create(const bbband_config *cfg, bbband_ctx **out) {
    if (!cfg || !out) return bfail(BBMAP_E_ARG, "bbband_create: null argument");
    *out = nullptr;
    if (cfg->width < 1 || cfg->width > 1023) return bfail(BBMAP_E_ARG, "bbband_create: width must be 1..1023");
    if (cfg->semantics != BBBAND_SEMANTICS_JNI_C && cfg->semantics != BBBAND_SEMANTICS_JAVA)
        return bfail(BBMAP_E_ARG, "bbband_create: unknown semantics");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bfail(BBMAP_E_NODEVICE, "bbband_create: no HIP device (no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return bfail(BBMAP_E_ARG, "bbband_create: bad device ordinal");
    BHIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    BHIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return bfail(BBMAP_E_NODEVICE, "bbband_create: this build targets gfx950 only");
    bbband_ctx *c = new (std::nothrow) bbband_ctx();
    if (!c) return bfail(BBMAP_E_NOMEM, "bbband_create: out of memory");
    c->device = cfg->device;
    c->maxWidth = (cfg->width < 3 ? 3 : cfg->width) | 1;            // BandedAligner.java:13
    c->variant = cfg->semantics;
    c->G = c->maxWidth <= 16 ? 16 : (c->maxWidth <= 32 ? 32 : 64);
    c->ldsBytes = 4 * (64 / c->G) * 2 * (c->maxWidth + 2) * 4;
    c->blocks = prop.multiProcessorCount * 8;
    BHIP(hipMalloc(&c->d_queue, 64));
    *out = c;
    return BBMAP_OK;
}

extern "C" void bbband_destroy(bbband_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_queue) (void)hipFree(c->d_queue);
    delete c;
}

extern "C" int bbband_align_batch_device(bbband_ctx *c, void *stream_, int64_t n, const bbband_job *jobs,
                                         const uint8_t *seqs, bbband_result *results) {
    if (!c) return bfail(BBMAP_E_ARG, "bbband_align_batch_device: null context");
    if (n < 0 || n > 0x7fffffffLL) return bfail(BBMAP_E_ARG, "bbband_align_batch_device: n_jobs out of range");
    if (n == 0) return BBMAP_OK;
    if (!jobs || !seqs || !results) return bfail(BBMAP_E_ARG, "bbband_align_batch_device: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    BHIP(hipSetDevice(c->device));
    BHIP(hipMemsetAsync(c->d_queue, 0, 64, stream));
    bbband::Params p;
    p.jobs = jobs; p.seqs = seqs; p.results = results; p.njobs = n; p.queue = c->d_queue;
    p.maxWidth = c->maxWidth; p.variant = c->variant; p.G = c->G;
    const int perBlock = 4 * (64 / c->G);
    long long blocks = (n + perBlock - 1) / perBlock;
    if (blocks > c->blocks) blocks = c->blocks;
    hipLaunchKernelGGL(bbband::banded_kernel, dim3((unsigned)blocks), dim3(256), (size_t)c->ldsBytes, stream, p);
    BHIP(hipGetLastError());
    return BBMAP_OK;
}

extern "C" int bbband_align_batch(bbband_ctx *c, int64_t n, const bbband_job *jobs,
                                  const uint8_t *seqs, int64_t seq_bytes, bbband_result *results) {
    if (!c) return bfail(BBMAP_E_ARG, "bbband_align_batch: null context");
    if (n == 0) return BBMAP_OK;
    if (n < 0 || !jobs || !seqs || !results || seq_bytes < 0) return bfail(BBMAP_E_ARG, "bbband_align_batch: bad argument");
    for (int64_t i = 0; i < n; i++) {
        const bbband_job &j = jobs[i];
        if (j.query_len < 0 || j.ref_len < 0 || j.query_off < 0 || j.ref_off < 0 ||
            j.query_off + j.query_len > seq_bytes || j.ref_off + j.ref_len > seq_bytes)
            return bfail(BBMAP_E_ARG, "bbband_align_batch: a sequence lies outside the seqs buffer");
    }
    BHIP(hipSetDevice(c->device));
    bbband_job *dj = nullptr; uint8_t *ds = nullptr; bbband_result *dr = nullptr;
    int rc = BBMAP_OK;
#define BGO(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_berr, sizeof g_berr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_berr); rc = BBMAP_E_HIP; goto done; } } while (0)
    BGO(hipMalloc(&dj, (size_t)n * sizeof(bbband_job)));
    BGO(hipMalloc(&ds, (size_t)(seq_bytes > 0 ? seq_bytes : 1)));
    BGO(hipMalloc(&dr, (size_t)n * sizeof(bbband_result)));
    BGO(hipMemcpy(dj, jobs, (size_t)n * sizeof(bbband_job), hipMemcpyHostToDevice));
    BGO(hipMemcpy(ds, seqs, (size_t)seq_bytes, hipMemcpyHostToDevice));
    rc = bbband_align_batch_device(c, nullptr, n, dj, ds, dr);
    if (rc != BBMAP_OK) goto done;
    BGO(hipStreamSynchronize(nullptr));
    BGO(hipMemcpy(results, dr, (size_t)n * sizeof(bbband_result), hipMemcpyDeviceToHost));
done:
    if (dj) (void)hipFree(dj);
    if (ds) (void)hipFree(ds);
    if (dr) (void)hipFree(dr);
    return rc;
#undef BGO
}

// ---------------------------------------------------------------------------------------------------------------------
// BandedAligner's orchestration (current/align2/BandedAligner.java:24-55), batched over pairs: alignQuadruple (forward +
// reverse, then the two reverse-complement directions with the tightened maxEdits), alignQuadrupleProgressive (maxEdits grows
// by 4x until a pair aligns below it) and alignDouble (forward, then forward-RC bounded by the forward result).  Every
// directional alignment runs in banded_kernel; the host only builds job lists and takes minima / maxima.
namespace {
struct PairBatch {
    bbband_ctx *c; int64_t n; const bbband_pair *pairs; bool exact;
    uint8_t *d_seqs = nullptr; bbband_job *d_jobs = nullptr; bbband_result *d_res = nullptr;
    std::vector<bbband_job> jobs; std::vector<bbband_result> res;
    ~PairBatch() { if (d_seqs) (void)hipFree(d_seqs); if (d_jobs) (void)hipFree(d_jobs); if (d_res) (void)hipFree(d_res); }
    int init(const uint8_t *seqs, int64_t seq_bytes) {
        for (int64_t i = 0; i < n; i++) {
            const bbband_pair &p = pairs[i];
            if (p.query_len < 0 || p.ref_len < 0 || p.query_off < 0 || p.ref_off < 0 || p.query_off + p.query_len > seq_bytes || p.ref_off + p.ref_len > seq_bytes)
                return bfail(BBMAP_E_ARG, "bbband pair batch: a sequence lies outside the seqs buffer");
        }
        BHIP(hipSetDevice(c->device));
        BHIP(hipMalloc(&d_seqs, (size_t)(seq_bytes > 0 ? seq_bytes : 1)));
        BHIP(hipMemcpy(d_seqs, seqs, (size_t)seq_bytes, hipMemcpyHostToDevice));
        BHIP(hipMalloc(&d_jobs, (size_t)(2 * n) * sizeof(bbband_job)));
        BHIP(hipMalloc(&d_res, (size_t)(2 * n) * sizeof(bbband_result)));
        jobs.resize((size_t)(2 * n)); res.resize((size_t)(2 * n));
        return BBMAP_OK;
    }
    // one launch over `m` jobs already written to jobs[0..m)
    int run(int64_t m) {
        if (m == 0) return BBMAP_OK;
        BHIP(hipMemcpy(d_jobs, jobs.data(), (size_t)m * sizeof(bbband_job), hipMemcpyHostToDevice));
        const int rc = bbband_align_batch_device(c, nullptr, m, d_jobs, d_seqs, d_res);
        if (rc != BBMAP_OK) return rc;
        BHIP(hipStreamSynchronize(nullptr));
        BHIP(hipMemcpy(res.data(), d_res, (size_t)m * sizeof(bbband_result), hipMemcpyDeviceToHost));
        return BBMAP_OK;
    }
    bbband_job job(int64_t i, int dir, int qstart, int rstart, int maxEdits) const {
        const bbband_pair &p = pairs[i];
        bbband_job j; j.query_off = p.query_off; j.ref_off = p.ref_off; j.query_len = p.query_len; j.ref_len = p.ref_len;
        j.qstart = qstart; j.rstart = rstart; j.maxEdits = maxEdits; j.flags = dir | (exact ? BBBAND_EXACT : 0);
        return j;
    }
    // alignQuadruple for the pairs in `idx` with their maxEdits in `me`; out[k] = result for idx[k]
    int quadruple(const std::vector<int64_t> &idx, const std::vector<int> &me, std::vector<int> &out) {
        const int64_t m = (int64_t)idx.size();
        out.assign((size_t)m, 0);
        for (int64_t k = 0; k < m; k++) {
            const bbband_pair &p = pairs[idx[(size_t)k]];
            jobs[(size_t)(2 * k)] = job(idx[(size_t)k], BBBAND_FORWARD, 0, 0, me[(size_t)k]);
            jobs[(size_t)(2 * k + 1)] = job(idx[(size_t)k], BBBAND_REVERSE, p.query_len - 1, p.ref_len - 1, me[(size_t)k]);
        }
        int rc = run(2 * m);
        if (rc != BBMAP_OK) return rc;
        std::vector<int64_t> todo; std::vector<int> me2s;
        for (int64_t k = 0; k < m; k++) {
            const int a = res[(size_t)(2 * k)].edits, b = res[(size_t)(2 * k + 1)].edits;
            const int mx = a > b ? a : b;
            out[(size_t)k] = mx;
            const int me2 = me[(size_t)k] < mx ? me[(size_t)k] : mx;
            if (me2 == 0) out[(size_t)k] = 0; else { todo.push_back(k); me2s.push_back(me2); }
        }
        for (size_t t = 0; t < todo.size(); t++) {
            const int64_t i = idx[(size_t)todo[t]];
            const bbband_pair &p = pairs[i];
            jobs[2 * t] = job(i, BBBAND_FORWARD_RC, p.query_len - 1, 0, me2s[t]);
            jobs[2 * t + 1] = job(i, BBBAND_REVERSE_RC, 0, p.ref_len - 1, me2s[t]);
        }
        rc = run((int64_t)(2 * todo.size()));
        if (rc != BBMAP_OK) return rc;
        for (size_t t = 0; t < todo.size(); t++) {
            const int cc = res[2 * t].edits, d = res[2 * t + 1].edits;
            const int mcd = cc > d ? cc : d;
            int &o = out[(size_t)todo[t]];
            o = o < mcd ? o : mcd;
        }
        return BBMAP_OK;
    }
};
}  // namespace

extern "C" int bbband_align_quadruple_batch(bbband_ctx *c, int64_t n, const bbband_pair *pairs, const uint8_t *seqs, int64_t seq_bytes,
                                            int32_t maxEdits, int32_t exact, int32_t *edits) {
    if (!c || n < 0 || (n > 0 && (!pairs || !seqs || !edits)) || seq_bytes < 0) return bfail(BBMAP_E_ARG, "bbband_align_quadruple_batch: bad argument");
    if (n == 0) return BBMAP_OK;
    PairBatch B{c, n, pairs, exact != 0};
    int rc = B.init(seqs, seq_bytes);
    if (rc != BBMAP_OK) return rc;
    std::vector<int64_t> idx((size_t)n); std::vector<int> me((size_t)n, maxEdits), out;
    for (int64_t i = 0; i < n; i++) idx[(size_t)i] = i;
    rc = B.quadruple(idx, me, out);
    if (rc != BBMAP_OK) return rc;
    for (int64_t i = 0; i < n; i++) edits[i] = out[(size_t)i];
    return BBMAP_OK;
}

extern "C" int bbband_align_quadruple_progressive_batch(bbband_ctx *c, int64_t n, const bbband_pair *pairs, const uint8_t *seqs, int64_t seq_bytes,
                                                        int32_t minEdits, int32_t maxEdits, int32_t exact, int32_t *edits) {
    if (!c || n < 0 || (n > 0 && (!pairs || !seqs || !edits)) || seq_bytes < 0) return bfail(BBMAP_E_ARG, "bbband_align_quadruple_progressive_batch: bad argument");
    if (n == 0) return BBMAP_OK;
    PairBatch B{c, n, pairs, exact != 0};
    int rc = B.init(seqs, seq_bytes);
    if (rc != BBMAP_OK) return rc;
    // BandedAligner.java:24-37, every pair with its own loop state (i, me); a round aligns the pairs that are still looping
    std::vector<long long> iv((size_t)n); std::vector<long long> mev((size_t)n, -1); std::vector<int> maxE((size_t)n);
    std::vector<char> done((size_t)n, 0);
    for (int64_t p = 0; p < n; p++) {
        const int longer = pairs[p].query_len > pairs[p].ref_len ? pairs[p].query_len : pairs[p].ref_len;
        maxE[(size_t)p] = maxEdits < longer ? maxEdits : longer;
        iv[(size_t)p] = minEdits < maxE[(size_t)p] ? minEdits : maxE[(size_t)p];
        edits[p] = maxE[(size_t)p];
    }
    for (;;) {
        std::vector<int64_t> idx; std::vector<int> me, out;
        for (int64_t p = 0; p < n; p++) {
            if (done[(size_t)p]) continue;
            if (!(mev[(size_t)p] < maxE[(size_t)p])) { done[(size_t)p] = 1; continue; }         // the for-loop's condition me < maxEdits
            long long m = iv[(size_t)p] < maxE[(size_t)p] ? iv[(size_t)p] : maxE[(size_t)p];
            if (m * 2 > maxE[(size_t)p]) m = maxE[(size_t)p];
            mev[(size_t)p] = m;
            idx.push_back(p); me.push_back((int)m);
        }
        if (idx.empty()) break;
        rc = B.quadruple(idx, me, out);
        if (rc != BBMAP_OK) return rc;
        for (size_t t = 0; t < idx.size(); t++) {
            const int64_t p = idx[t];
            if (out[t] < me[t]) { edits[p] = out[t]; done[(size_t)p] = 1; }
            else {
                if (iv[(size_t)p] == 0) { done[(size_t)p] = 1; continue; }        // i = i * 4 stays 0: the reference would loop forever; it never passes minEdits = 0 with maxEdits > 0 unaligned
                iv[(size_t)p] *= 4;
            }
        }
    }
    return BBMAP_OK;
}

extern "C" int bbband_align_double_batch(bbband_ctx *c, int64_t n, const bbband_pair *pairs, const uint8_t *seqs, int64_t seq_bytes,
                                         int32_t maxEdits, int32_t exact, int32_t *edits) {
    if (!c || n < 0 || (n > 0 && (!pairs || !seqs || !edits)) || seq_bytes < 0) return bfail(BBMAP_E_ARG, "bbband_align_double_batch: bad argument");
    if (n == 0) return BBMAP_OK;
    PairBatch B{c, n, pairs, exact != 0};
    int rc = B.init(seqs, seq_bytes);
    if (rc != BBMAP_OK) return rc;
    for (int64_t i = 0; i < n; i++) B.jobs[(size_t)i] = B.job(i, BBBAND_FORWARD, 0, 0, maxEdits);
    rc = B.run(n);
    if (rc != BBMAP_OK) return rc;
    std::vector<int64_t> todo;
    for (int64_t i = 0; i < n; i++) { edits[i] = B.res[(size_t)i].edits; if (edits[i] != 0) todo.push_back(i); }
    for (size_t t = 0; t < todo.size(); t++) B.jobs[t] = B.job(todo[t], BBBAND_FORWARD_RC, pairs[todo[t]].query_len - 1, 0, edits[todo[t]]);
    rc = B.run((int64_t)todo.size());
    if (rc != BBMAP_OK) return rc;
    for (size_t t = 0; t < todo.size(); t++) { const int cc = B.res[t].edits; if (cc < edits[todo[t]]) edits[todo[t]] = cc; }
    return BBMAP_OK;
}
