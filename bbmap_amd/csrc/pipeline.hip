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
// Not carried over (host-side policies of the mapper, out of scope here): trimList, findTipDeletions, the second wider
// fill after pad hints (scoreSlow :312-335), the expected-length cap (:296-303), fixXY / clipTipIndels on the result, and
// the coupling BETWEEN the sites of one read: scoreSlow raises minMsaLimit to (slowScore - CLEARZONE3) after every site
// (:375), so a read's later sites can get a higher minScore than its first.  Here every site of a read gets the initial
// minMsaLimit: identical for reads with one DP candidate (99.99 % of the bench workload), a looser bound -- never a lost
// alignment -- for the second and later candidates of a read.  A host that needs the exact per-read sequence submits the
// candidates of a read in rounds through bbmsa_align_batch_device.
// Sites that carry a gap array go to a second job list (bbmsa_align_gapped_batch_device builds their gapped reference).
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

// MSA.scoreNoIndels(read, ref, refStart) without base scores, MultiStateAligner11tsJNI.java:1034-1089, by one
// wavefront, 64 bases per step.  The reference's running state is (mode, timeInMode) where undefined bases ('N',
// bytes >= 128) leave both untouched; in event terms:
//   match: +100 if the previous event was a match, else +70
//   substitution: POINTS_SUB_ARRAY[t], t = length of the run of substitution events ending here
// Both come from the match / substitution ballots with bit arithmetic; the carry between steps is (last event was a
// match, length of the trailing substitution run).
__device__ int score_no_indels_wave(const uint8_t *read, int len, const uint8_t *ref, int reflen, int refStart, int lane) {
    int readStart = 0, readStop = len;
    if (refStart < 0) readStart = -refStart;                  // POINTS_NOREF is 0
    if (refStart + len > reflen) readStop -= (refStart + len - reflen);
    int score = 0, carrySub = 0;
    bool carryMatch = false;
    for (int base = readStart; base < readStop; base += 64) {
        const int i = base + lane;
        const bool valid = i < readStop;
        const int c = valid ? read[i] : 'N', r = valid ? ref[refStart + i] : 'N';
        const bool m = valid && c == r && c != 'N';
        const bool sub = valid && !m && !(c >= 128 || c == 'N') && !(r >= 128 || r == 'N');
        const u64 Mm = __ballot(m), Sm = __ballot(sub), ev = Mm | Sm, lt = lt_mask(lane);
        int contrib = 0;
        if (m) {
            const u64 pe = ev & lt;
            contrib = (pe ? (bool)((Mm >> hibit(pe)) & 1) : carryMatch) ? 100 : 70;
        } else if (sub) {
            const u64 pm = Mm & lt;
            const int t = pm ? popc(Sm & lt & gt_mask(hibit(pm))) + 1 : popc(Sm & lt) + 1 + carrySub;
            contrib = t > 5 ? -25 : (t > 1 ? -51 : -127);      // POINTS_SUB_ARRAY[t]
        }
        score += contrib;                                      // per lane; one reduction after the last step
        if (ev) {
            const int le = hibit(ev);
            if ((Mm >> le) & 1) { carryMatch = true; carrySub = 0; }
            else {
                const u64 upto = lt_mask(le) | (1ull << le), pm = Mm & upto;
                carrySub = pm ? popc(Sm & upto & gt_mask(hibit(pm))) : popc(Sm & upto) + carrySub;
                carryMatch = false;
            }
        }
    }
    return wsum(score);
}

// SiteScore.setPerfect (current/stream/SiteScore.java:239-292) by one wavefront, order-independent form (see
// index_probe_wave.hip): perfect = every base equal and called; semiperfect tolerates reference N for up to len/2 bases.
__device__ void set_perfect_wave(const uint8_t *read, int len, const uint8_t *ref, int reflen, int start, int stop, int lane,
                                 int &perfectOut, int &semiOut) {
    perfectOut = 0; semiOut = 0;
    if (len != stop - start + 1) return;
    bool perfect = true;
    int refloc = start, readloc = 0, N = 0;
    const int mx = min(stop, reflen - 1), nlimit = len / 2;
    if (start < 0) { N -= start; readloc -= start; refloc -= start; perfect = false; }
    if (stop >= reflen) { N += (stop - reflen + 1); perfect = false; }
    if (N > nlimit) return;
    bool anyHard = false, anyCN = false, anyBad = false;
    for (int j0 = 0; refloc + j0 <= mx; j0 += 64) {
        const int j = j0 + lane;
        bool bad = false, hard = false, cn = false;
        if (refloc + j <= mx) {
            const int c = read[readloc + j], r = ref[refloc + j];
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
    if (anyHard || N > nlimit) return;
    semiOut = anyCN ? 0 : 1;
    perfectOut = (perfect && !anyBad && !anyCN && N == 0) ? 1 : 0;
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
    unsigned int *counters;        // [0] jobs, [1] reads finished without DP, [2] gapped jobs (sites with gap arrays), [3] reads with no site
    int *noIndelScore;             // optional: per (read, site) ungapped score
    int extraFlags;                // OR-ed into every job's flags (e.g. BBMSA_NO_ITERATIONS)
    int gappedCap;                 // capacity of the gapped list
    uint8_t *ungMatch;             // optional: match strings of the reads finished without DP (see match_no_indels_kernel)
    int ungStride;
    int *ungLen;
    int *readState;                // optional, per read: -1 no site, (s << 2) | 1 finished without DP with best site s, 2 sent to DP
    bbmsa_job *gjobs;              // optional second list: jobs for sites that carry a gap array (need makeGref)
    bbmsa_gaps *ggaps;
    int *gjobSrc;
};

constexpr int SEL_WAVES = 4, SEL_READS_PER_WAVE = 16, SEL_CAP = 128;

// One wavefront takes SEL_READS_PER_WAVE consecutive reads; the DP jobs it selects are parked in LDS as
// (read * maxSites + site, minScore) and written out behind ONE reservation on the global job counter per flush.
#ifndef BBPIPE_SEL_OCC
#define BBPIPE_SEL_OCC 6
#endif
__global__ __launch_bounds__(64 * SEL_WAVES, BBPIPE_SEL_OCC) void select_jobs_kernel(const SelectParams P) {
    __shared__ int pendSrc[SEL_WAVES][SEL_CAP], pendMin[SEL_WAVES][SEL_CAP];
    __shared__ unsigned blockCnt[2], waveCnt[SEL_WAVES], blockBase;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < 2) blockCnt[threadIdx.x] = 0;
    __syncthreads();
    int npend = 0;
    unsigned cDone = 0, cNoSite = 0;
    auto write_jobs = [&](unsigned base) {
        for (int j = lane; j < npend; j += 64) {
            const int src = pendSrc[wave][j];
            const long long r = src / P.maxSites;
            const bbidx_read rr = P.reads[r];
            const bbidx_site ssj = P.sites[src];
            int start = ssj.start, stop = ssj.stop;
            if (stop - start + 1 + 2 * P.pad > P.maxColumns) stop = start + P.maxColumns - 2 * P.pad - 1;
            bbmsa_job j_;
            j_.read_off = rr.bases_off + (ssj.strand ? P.minus_delta : 0);
            j_.ref_off = P.chromOff[ssj.chrom];
            j_.read_len = rr.len; j_.ref_len = P.chromLen[ssj.chrom];
            j_.refStartLoc = start - P.pad; j_.refEndLoc = stop + P.pad;
            j_.minScore = pendMin[wave][j];
            j_.flags = BBMSA_FILL_AND_SCORE_LIMITED | BBMSA_DO_TRACEBACK | P.extraFlags;
            P.jobs[base + j] = j_;
            P.jobSrc[base + j] = src;
        }
        wsync();
        npend = 0;
    };
    auto flush = [&]() {                                  // mid-run overflow of the parking area: reserve for this wave alone
        if (npend == 0) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&P.counters[0], (unsigned)npend);
        write_jobs(__builtin_amdgcn_readfirstlane(base));
    };
    const long long r0 = ((long long)blockIdx.x * SEL_WAVES + wave) * SEL_READS_PER_WAVE;
    for (int q = 0; q < SEL_READS_PER_WAVE; q++) {
        const long long r = r0 + q;
        if (r >= P.nreads) break;
        const int ns = P.nsites[r];
        if (ns <= 0) { if (ns == 0) cNoSite++; if (lane == 0) { if (P.readState) P.readState[r] = -1; if (P.ungLen) P.ungLen[r] = 0; } continue; }
        const bbidx_read rr = P.reads[r];
        const int len = rr.len;
        const int maxSw = 70 + (len - 1) * 100;                        // msa.maxQuality(len)
        const int maxImperfect = maxSw + (-472 < -395 - 100 ? -472 : -395 - 100);   // msa.maxImperfectScore
        bbidx_site *ss = P.sites + r * (long long)P.maxSites;
        int near = 0; bool forceSlow = false;
        // lane s keeps what the second pass needs of site s (s < 64; further sites are re-read)
        int mySw = 0, mySemi = 0, myGaps = 0;
        int bestSw = INT_MIN, bestSite = 0;                            // first site with the highest ungapped score
        for (int s = 0; s < ns; s++) {
            const int strand = ss[s].strand, chrom = ss[s].chrom, oldScore = ss[s].score;
            int perfect = ss[s].perfect, semi = ss[s].semiperfect, ngaps = ss[s].ngaps, start = ss[s].start, stop = ss[s].stop;
            bool newStart = false;
            int sw;
            if (perfect) { near++; sw = maxSw; ngaps = 0; }
            else {
                const uint8_t *bases = P.bases + rr.bases_off + (strand ? P.minus_delta : 0);
                const uint8_t *cref = P.refs + P.chromOff[chrom];
                const int clen = P.chromLen[chrom];
                sw = score_no_indels_wave(bases, len, cref, clen, start, lane);
                // the read may belong at the site's stop rather than its start (AbstractMapThread.java:808-815)
                if (sw < oldScore && oldScore >= maxImperfect && stop - start + 1 != len) {
                    const int sw2 = score_no_indels_wave(bases, len, cref, clen, stop - len + 1, lane);
                    if (sw2 >= maxImperfect) {
                        sw = sw2; start = stop - len + 1; newStart = true;
                        set_perfect_wave(bases, len, cref, clen, start, stop, lane, perfect, semi);
                    }
                }
                if (sw >= maxImperfect) {
                    near++;
                    stop = start + len - 1; ngaps = 0;
                    if (sw >= maxSw) perfect = semi = 1;
                    else set_perfect_wave(bases, len, cref, clen, start, stop, lane, perfect, semi);      // :833-837
                } else if (oldScore >= maxImperfect) forceSlow = true;
            }
            if (lane == 0) {
                ss[s].score = sw; ss[s].stop = stop; ss[s].ngaps = ngaps; ss[s].perfect = perfect; ss[s].semiperfect = semi;
                if (newStart) ss[s].start = start;
                if (P.noIndelScore) P.noIndelScore[r * (long long)P.maxSites + s] = sw;
            }
            if (lane == (s & 63)) { mySw = sw; mySemi = semi; myGaps = ngaps; }
            if (sw > bestSw) { bestSw = sw; bestSite = s; }
        }
        __threadfence_block();                                        // lane 0's site updates before any lane re-reads them
        const int numNear = forceSlow ? -near : near;
        if (numNear >= 1) {
            cDone++;
            if (P.readState && lane == 0) P.readState[r] = (bestSite << 2) | 1;
            if (P.ungMatch) {
                // MSA.scoreNoIndelsAndMakeMatchString at the best site (MultiStateAligner11tsJNI.java:1244-1318): the read
                // and the reference bytes were just scored, so they come from cache
                __threadfence_block();
                const bbidx_site sb = ss[bestSite];
                const int reflen = P.chromLen[sb.chrom];
                int mlen = len;
                if (sb.start < 0 || sb.start + len > reflen || len > P.ungStride) mlen = -1;      // the reference returns -99999
                else {
                    const uint8_t *bases = P.bases + rr.bases_off + (sb.strand ? P.minus_delta : 0);
                    const uint8_t *cref = P.refs + P.chromOff[sb.chrom] + sb.start;
                    uint8_t *outm = P.ungMatch + r * (long long)P.ungStride;
                    for (int i = lane; i < len; i += 64) {
                        const int c = bases[i], q2 = cref[i];
                        outm[i] = (c == q2 && c != 'N') ? 'm' : ((c >= 128 || c == 'N' || q2 >= 128 || q2 == 'N') ? 'N' : 'S');
                    }
                }
                if (lane == 0) P.ungLen[r] = mlen;
            }
            continue;
        }
        if (lane == 0) { if (P.readState) P.readState[r] = 2; if (P.ungLen) P.ungLen[r] = 0; }
        const int minMsaLimit = -258 + (int)__fmul_rn(P.minRatio, (float)maxSw);     // -CLEARZONE1e + (int)(ratio*maxSwScore)
        for (int s0 = 0; s0 < ns; s0 += 64) {
            const int s = s0 + lane;
            int sw = mySw, semi = mySemi, gaps = myGaps;
            if (s0 > 0 && s < ns) { sw = ss[s].score; semi = ss[s].semiperfect; gaps = ss[s].ngaps; }   // written by lane 0 above
            const bool cand = s < ns && sw < maxImperfect && !semi;
            const bool want = cand && gaps == 0;
            if (cand && gaps > 0) {                                   // rare: straight to the gapped list
                const unsigned k = atomicAdd(&P.counters[2], 1u);
                if (P.gjobs && k < (unsigned)P.gappedCap) {
                    const bbidx_site sg = ss[s];
                    bbmsa_job j_;
                    j_.read_off = rr.bases_off + (sg.strand ? P.minus_delta : 0);
                    j_.ref_off = P.chromOff[sg.chrom];
                    j_.read_len = len; j_.ref_len = P.chromLen[sg.chrom];
                    j_.refStartLoc = sg.start - P.pad; j_.refEndLoc = sg.stop + P.pad;
                    j_.minScore = max(sw, minMsaLimit);
                    j_.flags = BBMSA_FILL_AND_SCORE_LIMITED | BBMSA_DO_TRACEBACK;
                    P.gjobs[k] = j_;
                    bbmsa_gaps gg;
                    gg.ngaps = sg.ngaps;
                    for (int q = 0; q < BBMSA_MAX_GAPS; q++) gg.gaps[q] = q < sg.ngaps ? sg.gaps[q] : 0;
                    P.ggaps[k] = gg;
                    P.gjobSrc[k] = (int)(r * (long long)P.maxSites + s);
                }
            }
            const u64 W = __ballot(want);
            const int nw = popc(W);
            if (npend + nw > SEL_CAP) flush();
            if (want) { const int slot = npend + popc(W & lt_mask(lane)); pendSrc[wave][slot] = (int)(r * (long long)P.maxSites + s); pendMin[wave][slot] = max(sw, minMsaLimit); }
            npend += nw;
            wsync();
        }
    }
    // one reservation on the global job counter per BLOCK (a single word takes ~88 returning atomics per microsecond:
    // per-wave reservations alone cost more than the scoring), then every wave writes its parked jobs
    __threadfence_block();
    if (lane == 0) { waveCnt[wave] = (unsigned)npend; atomicAdd(&blockCnt[0], cDone); atomicAdd(&blockCnt[1], cNoSite); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < SEL_WAVES; w++) total += waveCnt[w];
        blockBase = total ? atomicAdd(&P.counters[0], total) : 0u;
    }
    __syncthreads();
    {
        unsigned base = blockBase;
        for (int w = 0; w < wave; w++) base += waveCnt[w];
        write_jobs(base);
    }
    if (threadIdx.x < 2 && blockCnt[threadIdx.x]) atomicAdd(&P.counters[threadIdx.x == 0 ? 1 : 3], blockCnt[threadIdx.x]);
}

// MSA.scoreNoIndelsAndMakeMatchString(read, ref, refStart, matchReturn) (MultiStateAligner11tsJNI.java:1244-1318) for the
// reads the site filter finished without DP: their best site needs no alignment, its match string is one symbol per base
// ('m' equal and called, 'N' when the read or the reference base is undefined, 'S' otherwise).  One wavefront per read.
struct MatchParams {
    const bbidx_read *reads;
    const uint8_t *bases;
    long long minus_delta;
    const bbidx_site *sites;
    int maxSites;
    const int *readState;
    const long long *chromOff;
    const int *chromLen;
    const uint8_t *refs;
    long long nreads;
    uint8_t *match;
    int stride;
    int *matchLen;
};

__global__ __launch_bounds__(256) void match_no_indels_kernel(const MatchParams P) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= P.nreads) return;
    const int st = P.readState[r];
    if ((st & 3) != 1 || st < 0) { if (lane == 0) P.matchLen[r] = 0; return; }
    const bbidx_read rr = P.reads[r];
    const bbidx_site ss = P.sites[r * (long long)P.maxSites + (st >> 2)];
    const int len = rr.len, reflen = P.chromLen[ss.chrom];
    if (ss.start < 0 || ss.start + len > reflen || len > P.stride) { if (lane == 0) P.matchLen[r] = -1; return; }   // the reference returns -99999
    const uint8_t *bases = P.bases + rr.bases_off + (ss.strand ? P.minus_delta : 0);
    const uint8_t *ref = P.refs + P.chromOff[ss.chrom] + ss.start;
    uint8_t *out = P.match + r * (long long)P.stride;
    for (int i = lane; i < len; i += 64) {
        const int c = bases[i], q = ref[i];
        out[i] = (c == q && c != 'N') ? 'm' : ((c >= 128 || c == 'N' || q >= 128 || q == 'N') ? 'N' : 'S');
    }
    if (lane == 0) P.matchLen[r] = len;
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
    __shared__ uint8_t rdbuf[RESC_WAVES][RESC_MAXLEN];
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
        for (int j = 0; j < len; j++) {
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

extern "C" int bbpipe_select_jobs_device(void *stream_, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases,
                                         int64_t minus_delta, const int32_t *nsites, bbidx_site *sites, int32_t max_sites,
                                         const int64_t *chrom_off, const int32_t *chrom_len, const uint8_t *refs,
                                         int32_t pad, int32_t max_columns, float min_ratio,
                                         bbmsa_job *jobs, int32_t *job_src, uint32_t *counters, int32_t *no_indel_score,
                                         bbmsa_job *gapped_jobs, bbmsa_gaps *gapped_gaps, int32_t *gapped_src, int32_t gapped_cap,
                                         int32_t extra_job_flags, int32_t *read_state,
                                         uint8_t *ungapped_match, int32_t ungapped_stride, int32_t *ungapped_len) {
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
    if ((gapped_jobs != nullptr) != (gapped_gaps != nullptr) || (gapped_jobs != nullptr) != (gapped_src != nullptr)) {
        bbmap_set_error("bbpipe_select_jobs_device: the three gapped-list buffers go together"); return BBMAP_E_ARG;
    }
    P.gjobs = gapped_jobs; P.ggaps = gapped_gaps; P.gjobSrc = gapped_src;
    P.extraFlags = extra_job_flags & BBMSA_NO_ITERATIONS;
    P.gappedCap = gapped_cap;
    P.readState = read_state;
    if ((ungapped_match != nullptr) != (ungapped_len != nullptr) || (ungapped_match && ungapped_stride < 1)) {
        bbmap_set_error("bbpipe_select_jobs_device: ungapped_match, ungapped_stride and ungapped_len go together"); return BBMAP_E_ARG;
    }
    P.ungMatch = ungapped_match; P.ungStride = ungapped_stride; P.ungLen = ungapped_len;
    const long long per_block = bbpipe::SEL_WAVES * bbpipe::SEL_READS_PER_WAVE;
    const long long blocks = (n_reads + per_block - 1) / per_block;
    hipLaunchKernelGGL(bbpipe::select_jobs_kernel, dim3((unsigned)blocks), dim3(64 * bbpipe::SEL_WAVES), 0, stream, P);
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

extern "C" int bbpipe_match_no_indels_device(void *stream_, int64_t n_reads, const bbidx_read *reads, const uint8_t *bases,
                                             int64_t minus_delta, const bbidx_site *sites, int32_t max_sites, const int32_t *read_state,
                                             const int64_t *chrom_off, const int32_t *chrom_len, const uint8_t *refs,
                                             uint8_t *match, int32_t match_stride, int32_t *match_len) {
    if (n_reads < 0 || max_sites < 1 || match_stride < 1) { bbmap_set_error("bbpipe_match_no_indels_device: bad size"); return BBMAP_E_ARG; }
    if (n_reads == 0) return BBMAP_OK;
    if (!reads || !bases || !sites || !read_state || !chrom_off || !chrom_len || !refs || !match || !match_len) {
        bbmap_set_error("bbpipe_match_no_indels_device: null buffer"); return BBMAP_E_ARG;
    }
    bbpipe::MatchParams P;
    P.reads = reads; P.bases = bases; P.minus_delta = minus_delta; P.sites = sites; P.maxSites = max_sites; P.readState = read_state;
    P.chromOff = (const long long *)chrom_off; P.chromLen = chrom_len; P.refs = refs; P.nreads = n_reads;
    P.match = match; P.stride = match_stride; P.matchLen = match_len;
    hipLaunchKernelGGL(bbpipe::match_no_indels_kernel, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, P);
    PHIP(hipGetLastError());
    return BBMAP_OK;
}
