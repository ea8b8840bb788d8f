// Device-side glue between the index probe and the DP: which probe sites need a slow alignment.
//
// In the reference this decision is host Java between the two hot kernels
// (current/align2/BBMapThread.java:389-470 processRead -> AbstractMapThread.scoreNoIndels :762-856 ->
// BBMapThread.scoreSlow :252-386).  Doing it on the device keeps the read batch in HBM from probe to DP:
//   1. every probe site gets the ungapped score MSA.scoreNoIndels (MultiStateAligner11tsJNI.java:1034-1089);
//      a perfect site keeps maxQuality (AbstractMapThread.java:788-797);
//   2. a read with at least one "near perfect" site (ungapped score >= maxImperfectScore) is finished without DP
//      (BBMapThread.java:457: `if(numNearPerfectScores<1) scoreSlow`), unless a site whose index score was near
//      perfect scored lower without indels (forceSlow, AbstractMapThread.java:836-838);
//   3. otherwise every site with ungapped score < maxImperfectScore and not semiperfect becomes one DP job:
//      window = site +- SLOW_ALIGN_PADDING, minScore = max(ungapped score, minMsaLimit) (BBMapThread.java:289-309).
// Not carried over (host-side policies of the mapper, out of scope here): trimList, findTipDeletions, the
// stop-anchored retry of scoreNoIndels (:808-815), the second wider fill after pad hints (scoreSlow :312-335) and
// sites that carry a gap array (they need makeGref; such sites are counted and skipped).
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bbmap_amd.h"

void bbmap_set_error(const char *msg);

namespace bbpipe {

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

// MSA.scoreNoIndels(read, ref, refStart) without base scores, MultiStateAligner11tsJNI.java:1034-1089
__device__ int score_no_indels(const uint8_t *read, int len, const uint8_t *ref, int reflen, int refStart) {
    int score = 0, mode = -1, timeInMode = 0, readStart = 0, readStop = len;
    if (refStart < 0) readStart = -refStart;                  // POINTS_NOREF is 0
    if (refStart + len > reflen) readStop -= (refStart + len - reflen);
    for (int i = readStart; i < readStop; i++) {
        const int c = read[i], r = ref[refStart + i];
        if (c == r && c != 'N') {
            if (mode == 0) { timeInMode++; score += 100; } else { timeInMode = 0; score += 70; }
            mode = 0;
        } else if (c >= 128 || c == 'N') {
        } else if (r >= 128 || r == 'N') {
        } else {
            if (mode == 3) timeInMode++; else timeInMode = 0;
            const int t = timeInMode + 1;
            score += t > 5 ? -25 : (t > 1 ? -51 : -127);          // POINTS_SUB_ARRAY[timeInMode+1]
            mode = 3;
        }
    }
    return score;
}

struct SelectParams {
    const bbidx_read *reads;
    const uint8_t *bases;          // plus-strand reads; the reverse complements sit minus_delta bytes further
    long long minus_delta;
    const int *nsites;
    bbidx_site *sites;
    int maxSites;
    const long long *chromOff;     // [nchroms+1] offsets of the chromosome arrays inside `refs`
    const int *chromLen;
    const uint8_t *refs;
    long long nreads;
    int pad, maxColumns;
    float minRatio;
    bbmsa_job *jobs;
    int *jobSrc;                   // read * maxSites + site for each job
    unsigned int *counters;        // [0] jobs, [1] reads finished without DP, [2] sites skipped (gap arrays), [3] reads with no site
    int *noIndelScore;             // optional: per (read, site) ungapped score
};

__global__ void select_jobs_kernel(const SelectParams P) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.nreads) return;
    const int ns = P.nsites[r];
    if (ns <= 0) { if (ns == 0) atomicAdd(&P.counters[3], 1u); return; }
    const bbidx_read rr = P.reads[r];
    const int len = rr.len;
    const int maxSw = 70 + (len - 1) * 100;                        // msa.maxQuality(len)
    const int maxImperfect = maxSw + (-472 < -395 - 100 ? -472 : -395 - 100);   // msa.maxImperfectScore
    bbidx_site *ss = P.sites + r * (long long)P.maxSites;
    int near = 0; bool forceSlow = false;
    for (int s = 0; s < ns; s++) {
        const uint8_t *bases = P.bases + rr.bases_off + (ss[s].strand ? P.minus_delta : 0);
        const int oldScore = ss[s].score;
        int sw;
        if (ss[s].perfect) { near++; sw = maxSw; ss[s].ngaps = 0; }
        else {
            const uint8_t *ref = P.refs + P.chromOff[ss[s].chrom];
            sw = score_no_indels(bases, len, ref, P.chromLen[ss[s].chrom], ss[s].start);
            if (sw >= maxImperfect) {
                near++;
                ss[s].stop = ss[s].start + len - 1; ss[s].ngaps = 0;
                if (sw >= maxSw) ss[s].perfect = ss[s].semiperfect = 1;
            } else if (oldScore >= maxImperfect) forceSlow = true;
        }
        ss[s].score = sw;                                             // ss.setScore(slowScoreNoIndel)
        if (P.noIndelScore) P.noIndelScore[r * (long long)P.maxSites + s] = sw;
    }
    const int numNear = forceSlow ? -near : near;
    if (numNear >= 1) { atomicAdd(&P.counters[1], 1u); return; }
    const int minMsaLimit = -258 + (int)__fmul_rn(P.minRatio, (float)maxSw);     // -CLEARZONE1e + (int)(ratio*maxSwScore)
    for (int s = 0; s < ns; s++) {
        const int sw = ss[s].score;
        if (!(sw < maxImperfect && !ss[s].semiperfect)) continue;
        if (ss[s].ngaps > 0) { atomicAdd(&P.counters[2], 1u); continue; }
        int start = ss[s].start, stop = ss[s].stop;
        if (stop - start + 1 + 2 * P.pad > P.maxColumns) stop = start + P.maxColumns - 2 * P.pad - 1;
        const unsigned k = atomicAdd(&P.counters[0], 1u);
        bbmsa_job j;
        j.read_off = rr.bases_off + (ss[s].strand ? P.minus_delta : 0);
        j.ref_off = P.chromOff[ss[s].chrom];
        j.read_len = len; j.ref_len = P.chromLen[ss[s].chrom];
        j.refStartLoc = start - P.pad; j.refEndLoc = stop + P.pad;
        j.minScore = max(sw, minMsaLimit);
        j.flags = BBMSA_FILL_AND_SCORE_LIMITED | BBMSA_DO_TRACEBACK;
        P.jobs[k] = j;
        P.jobSrc[k] = (int)(r * (long long)P.maxSites + s);
    }
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

extern "C" int bbpipe_select_jobs_device(void *stream_, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases,
                                         int64_t minus_delta, const int32_t *nsites, bbidx_site *sites, int32_t max_sites,
                                         const int64_t *chrom_off, const int32_t *chrom_len, const uint8_t *refs,
                                         int32_t pad, int32_t max_columns, float min_ratio,
                                         bbmsa_job *jobs, int32_t *job_src, uint32_t *counters, int32_t *no_indel_score) {
    if (n_reads < 0 || max_sites < 1 || pad < 0 || max_columns < 1) { bbmap_set_error("bbpipe_select_jobs_device: bad size"); return BBMAP_E_ARG; }
    if (n_reads == 0) return BBMAP_OK;
    if (!reads || !bases || !nsites || !sites || !chrom_off || !chrom_len || !refs || !jobs || !job_src || !counters) {
        bbmap_set_error("bbpipe_select_jobs_device: null buffer"); return BBMAP_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    PHIP(hipMemsetAsync(counters, 0, 16, stream));
    bbpipe::SelectParams P;
    P.reads = reads; P.bases = bases; P.minus_delta = minus_delta; P.nsites = nsites; P.sites = sites; P.maxSites = max_sites;
    P.chromOff = (const long long *)chrom_off; P.chromLen = chrom_len; P.refs = refs; P.nreads = n_reads;
    P.pad = pad; P.maxColumns = max_columns; P.minRatio = min_ratio; P.jobs = jobs; P.jobSrc = job_src; P.counters = counters;
    P.noIndelScore = no_indel_score;
    const long long blocks = (n_reads + 255) / 256;
    hipLaunchKernelGGL(bbpipe::select_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, P);
    PHIP(hipGetLastError());
    return BBMAP_OK;
}
