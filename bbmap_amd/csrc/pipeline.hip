// Batch helpers of the mapper that are wave-cooperative byte scans: reverse complement of a read batch and the paired-read
// rescue scan (AbstractMapThread.quickRescue).  The control flow that decides which sites are aligned lives in mapper.hip.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>

#include "bbmap_amd.h"
#include "wave_prims.h"

void bbmap_set_error(const char *msg);

namespace bbpipe {
using namespace wavep;

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

// AminoAcid.reverseComplementBases over a batch: one wave per read, coalesced
__global__ void revcomp_kernel(const bbidx_read *reads, long long n, const uint8_t *in, uint8_t *out) {
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    const bbidx_read rr = reads[r];
    for (int i = threadIdx.x & 63; i < rr.len; i += 64) out[rr.bases_off + i] = (uint8_t)complement_extended(in[rr.bases_off + rr.len - 1 - i]);
}

// ---------------------------------------------------------------------------------------------------------------
// Paired-read rescue scan: AbstractMapThread.quickRescue (current/align2/AbstractMapThread.java:2300-2391).
// The mate of a mapped read is slid over up to searchDist reference positions; per start the reference counts
// mismatches (giving up once they exceed the best count so far) and the longest completed match run, and keeps the
// best (length - mismatches + run) with the start closest to idealStart as tie-break; a perfect hit narrows the
// remaining search range.  Here: one wavefront per job, 64 consecutive starts per step.  Lane s compares the read
// (LDS, broadcast) with ref[start_s + j] (consecutive lanes -> consecutive bytes: coalesced).  The acceptance rule is
// order-dependent, so the (few) lanes whose count stayed within the cap are replayed in search order with readlanes.
struct RescueParams {
    const bbresc_job *jobs;
    const uint8_t *reads;
    const long long *chromOff;
    const int *chromLen, *chromMin;
    const uint8_t *refs;
    bbresc_result *results;
    long long njobs;
    int pointsMatch, pointsMatch2, useAffine, baseHitScore;
};

constexpr int RESC_WAVES = 4, RESC_MAXLEN = 608;

__global__ __launch_bounds__(64 * RESC_WAVES) void quick_rescue_kernel(const RescueParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t rdbuf[RESC_WAVES][RESC_MAXLEN];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long jx = (long long)blockIdx.x * RESC_WAVES + wave;
    if (jx >= P.njobs) return;                                   // whole wave; the kernel has no block-level barrier
    const bbresc_job jb = P.jobs[jx];
    bbresc_result res;
    res.found = 0; res.start = 0; res.stop = 0; res.score = 0; res.mismatches = 0; res.perfect = 0; res.semiperfect = 0; res.maxContig = 0;
    const int len = jb.read_len;
    if (len < 10 || len > RESC_MAXLEN - 8 || jb.chrom < 1) { if (lane == 0) { res.found = len > RESC_MAXLEN - 8 ? -2 : 0; P.results[jx] = res; } return; }
    const uint8_t *bases = P.reads + jb.read_off;
    const uint8_t *ref = P.refs + P.chromOff[jb.chrom];
    const int reflen = P.chromLen[jb.chrom], minIndex = P.chromMin[jb.chrom];
    uint8_t *rd = rdbuf[wave];
    for (int i = lane; i < len; i += 64) rd[i] = bases[i];
    wsync();
    const bool right = (jb.flags & 1) != 0;
    int lower, upper;
    if (right) { lower = max(minIndex, jb.loc); upper = min(reflen - len, jb.loc + jb.searchDist); }
    else { lower = max(minIndex, jb.loc - jb.searchDist); upper = min(reflen - len, jb.loc); }
    int minMM = jb.maxAllowedMismatches + 1, maxContig = 0, bestScore = 0, bestStart = -1, bestAbsdif = INT_MAX;
    bool finished = false;
    for (int base = 0; !finished; base += 64) {
        const int start = right ? lower + base + lane : upper - base - lane;
        const bool valid = start >= lower && start <= upper;
        if (!__ballot(valid)) break;
        const int cap = minMM;
        int mm = 0, contig = 0, cur = 0;
        // four bases per step: one LDS dword of the read against one (unaligned) dword of the reference; a lane that has passed
        // its mismatch cap may run up to three bases further than the byte-wise loop would, but its counts are discarded anyway
        const int len4 = len & ~3;
        int j = 0;
        for (; j < len4; j += 4) {
            const bool on = valid && mm <= cap;
            if (!__ballot(on)) break;
            const unsigned c4 = *(const unsigned *)(rd + j);
            if (on) {
                unsigned r4;
                __builtin_memcpy(&r4, ref + start + j, 4);
                unsigned x = c4 ^ r4;                                                  // a zero byte = equal bases
                const unsigned isN = c4 ^ 0x4e4e4e4eu;                                   // a zero byte = the read base is 'N'
                const unsigned nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;            // 0x80 per mismatching byte
                const unsigned nn = ~((((isN & 0x7f7f7f7fu) + 0x7f7f7f7fu) | isN)) & 0x80808080u;     // 0x80 per 'N' of the read
                const unsigned bad = nz | nn;
                if (bad == 0) cur += 4;
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (bad & (0x80u << (8 * q))) { mm++; contig = max(contig, cur); cur = 0; } else cur++;
                    }
                }
            }
        }
        for (; j < len; j++) {
            const bool on = valid && mm <= cap;
            if (!__ballot(on)) break;
            const int c = rd[j];
            if (on) {
                const int r = ref[start + j];
                if (c != r || c == 'N') { mm++; contig = max(contig, cur); cur = 0; } else cur++;
            }
        }
        for (u64 m = __ballot(valid && mm <= cap); m; m &= m - 1) {
            const int s = __builtin_ctzll(m);
            const int st = rl(start, s), ms = rl(mm, s), ct = rl(contig, s);
            if (right ? st > upper : st < lower) { finished = true; break; }      // a perfect hit narrowed the range
            if (ms > minMM) continue;
            const int score = (len - ms) + ct;
            const int ad = st > jb.idealStart ? st - jb.idealStart : jb.idealStart - st;
            if (score > bestScore || (score == bestScore && ad < bestAbsdif)) {
                bestStart = st; minMM = ms; maxContig = ct; bestScore = score; bestAbsdif = ad;
                if (ms == 0) { if (right) upper = min(upper, jb.idealStart + ad); else lower = max(lower, jb.idealStart - ad); }
            }
        }
    }
    if (bestStart >= 0) {
        res.found = 1; res.start = bestStart; res.stop = bestStart + len - 1; res.mismatches = minMM; res.maxContig = maxContig;
        res.score = P.useAffine ? P.pointsMatch + P.pointsMatch2 * (len - 1 - minMM) : maxContig + P.baseHitScore * (len - minMM);
        // SiteScore.setPerfect (current/stream/SiteScore.java:239-292), order-independent form (see index_probe_wave.hip)
        bool perfect = true;
        int refloc = res.start, readloc = 0, N = 0;
        const int mx = min(res.stop, reflen - 1), nlimit = len / 2;
        if (res.start < 0) { N -= res.start; readloc -= res.start; refloc -= res.start; perfect = false; }
        if (res.stop >= reflen) { N += (res.stop - reflen + 1); perfect = false; }
        bool anyHard = false, anyCN = false, anyBad = false;
        if (N <= nlimit) {
            for (int j0 = 0; refloc + j0 <= mx; j0 += 64) {
                const int j = j0 + lane;
                bool bad = false, hard = false, cn = false;
                if (refloc + j <= mx) {
                    const int c = rd[readloc + j], r = ref[refloc + j];
                    bad = (c != r || c == 'N'); hard = bad && r != 'N'; cn = bad && c == 'N';
                }
                const u64 badM = __ballot(bad);
                if (badM) {
                    anyBad = true;
                    if (__ballot(hard)) { anyHard = true; break; }
                    if (__ballot(cn)) anyCN = true;
                    N += popc(badM);
                    if (N > nlimit) break;
                }
            }
            if (!anyHard && N <= nlimit) { res.semiperfect = anyCN ? 0 : 1; res.perfect = (perfect && !anyBad && !anyCN && N == 0) ? 1 : 0; }
        }
    }
    if (lane == 0) P.results[jx] = res;
}

}  // namespace bbpipe

static thread_local char g_perr[256];
#define PHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_perr, sizeof g_perr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_perr); return BBMAP_E_HIP; } } while (0)

extern "C" int bbpipe_revcomp_device(void *stream_, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases_in, uint8_t *bases_out) {
    if (n_reads < 0) { bbmap_set_error("bbpipe_revcomp_device: bad size"); return BBMAP_E_ARG; }
    if (n_reads == 0) return BBMAP_OK;
    if (!reads || !bases_in || !bases_out) { bbmap_set_error("bbpipe_revcomp_device: null buffer"); return BBMAP_E_ARG; }
    const long long blocks = (n_reads + 3) / 4;
    hipLaunchKernelGGL(bbpipe::revcomp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, reads, (long long)n_reads, bases_in, bases_out);
    PHIP(hipGetLastError());
    return BBMAP_OK;
}

extern "C" int bbpipe_quick_rescue_device(void *stream_, int64_t n_jobs, const bbresc_job *jobs, const uint8_t *reads,
                                          const int64_t *chrom_off, const int32_t *chrom_len, const int32_t *chrom_min_index,
                                          const uint8_t *refs, bbresc_result *results,
                                          int32_t points_match, int32_t points_match2, int32_t use_affine, int32_t base_hit_score) {
    if (n_jobs < 0) { bbmap_set_error("bbpipe_quick_rescue_device: bad size"); return BBMAP_E_ARG; }
    if (n_jobs == 0) return BBMAP_OK;
    if (!jobs || !reads || !chrom_off || !chrom_len || !chrom_min_index || !refs || !results) {
        bbmap_set_error("bbpipe_quick_rescue_device: null buffer"); return BBMAP_E_ARG;
    }
    bbpipe::RescueParams P;
    P.jobs = jobs; P.reads = reads; P.chromOff = (const long long *)chrom_off; P.chromLen = chrom_len; P.chromMin = chrom_min_index;
    P.refs = refs; P.results = results; P.njobs = n_jobs;
    P.pointsMatch = points_match; P.pointsMatch2 = points_match2; P.useAffine = use_affine; P.baseHitScore = base_hit_score;
    const long long blocks = (n_jobs + bbpipe::RESC_WAVES - 1) / bbpipe::RESC_WAVES;
    hipLaunchKernelGGL(bbpipe::quick_rescue_kernel, dim3((unsigned)blocks), dim3(64 * bbpipe::RESC_WAVES), 0, (hipStream_t)stream_, P);
    PHIP(hipGetLastError());
    return BBMAP_OK;
}
