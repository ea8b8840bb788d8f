// Strip-tiled wavefront DP for long reads (gfx950): the MultiStateAligner9PacBio parameter set, reads of up to 6,100 bases
// against windows of up to 8,192 columns (mapPacBio: ALIGN_ROWS 6020, ALIGN_COLUMNS 7600, current/align2/BBMapThreadPacBio.java:27-28;
// recurrence current/align2/MultiStateAligner9PacBio.java:128-560, constants :2359-2439 -- the same skeleton as
// jni/MultiStateAligner11tsJNI.c:100-704 with other constants, 9 time bits and barriers of 1).
//
// One alignment per wavefront.  The read is cut into horizontal STRIPS of 64 x R rows; inside a strip lane l owns R consecutive
// rows and sweeps them left to right one column behind lane l-1 (the anti-diagonal wavefront of msa_fill_fast.hip: the three
// values a row needs from the row above arrive by DPP, nothing else leaves the lane's registers).  Between strips only the
// boundary row travels: the last row's three planes (one int per column and plane) go to HBM as the last lane produces them and
// come back 64 columns at a time, coalesced, to feed lane 0 of the next strip through a readlane; the first / last "good" column
// of that row travel as two scalars.  A 6,000 x 7,600 matrix is 549 MB in the reference (3 planes of int32) and never exists
// here; what is kept per cell is the 4-bit traceback record (what traceback2 / score2 would decide at that cell), 23 MB per job
// in HBM, and the walk reads it back cooperatively, 64 diagonal cells per step, across strip borders.
// The score-pruned window of fillLimitedX is reproduced exactly as in msa_fill_fast.hip: cells outside
// a row's window read as `subfloor`, over-computing provably-pruned cells changes nothing, and the visited-cell count is
// recovered from the rows' first / last good columns.  Banded fills are handed to the one-job-per-thread kernel
// (msa_fill_generic.hip); windows narrower than the read are taken here.
#include "msa_common.h"
#include "msa_cell.h"

namespace bbmsa {

struct StripParams {
    const bbmsa_job *jobs;
    const uint8_t *reads;
    const uint8_t *refs;
    bbmsa_result *results;
    uint8_t *match;
    long long njobs;
    const unsigned int *njobs_dev;
    unsigned int *queue;          // work-queue head (zeroed before launch)
    unsigned int *dirbuf;         // per resident wave: strips x dir_strip_dwords
    long long dir_slot_dwords, dir_strip_dwords;
    int *boundary;                // per resident wave: 2 x 3 x (maxColumns + 2) ints (ping-pong boundary rows)
    uint8_t *tmpbuf;              // per resident wave: maxRows + maxColumns + 8 bytes (reversed match string)
    int *slow_list;               // jobs handed to the generic kernel
    unsigned int *slow_count;
    int match_stride;
    int maxRows, maxColumns;
    int bandwidth;
    float bandwidthRatio;
    // pipelined form (PIPE): the strips of ONE job run in pipeK wavefronts (blocks slot * pipeK + w), strip w a few dozen columns
    // behind strip w - 1; boundary rows per strip and the hand-shake words live in these two buffers
    int pipeK, pipeSlots;
    int *pipeBoundary;            // per slot: pipeK x 3 x (maxColumns + 2) ints
    int *pipeSync;                // per slot: PIPE_SYNC_INTS(pipeK) ints, zeroed before the launch
    int pipeSpinLimit;            // polls of a hand-shake word before a wave gives up (BBMSA_PIPE_SPIN_LIMIT; ~3 s by default)
};

namespace {

__device__ __forceinline__ int lane_up(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ unsigned load_coherent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// what this wave wrote earlier (boundary rows, the reversed match string) is read back past the vector L1, which may still hold
// the lines as they were before the writes
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint8_t ld_agent_u8(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// hand-shake words of a slot: [0] "go": jobs ALL of whose pipeK waves are through with, [1] arrivals (every wave adds one per job when
// it is through with it; the last one to arrive zeroes the per-strip words and raises [0]), [3] the slot is dead (a wave gave up waiting:
// its jobs go to the one-thread kernel), [6] jobs claimed (a job's result is written, or the job handed on, by whoever raises this from
// its index: exactly once); then per strip: columns of its last row published, first good column of that row so far (-1: none), last
// good column (final), state (0 running, 1 done, 2 done and dead: the row has no good cell, 3 failed), first row not entered, visited
// cells (2 ints).
// Why so careful: the waves of a slot are only ASSUMED co-resident (the grid is sized from an occupancy query).  If one of them starts
// late, its partners time out -- and must neither reuse the hand-shake words while the straggler may still write them (no job starts
// before all pipeK waves have arrived from the one before) nor lose a job (a dead slot's remaining jobs are claimed one by one and
// handed to the one-thread kernel by whichever wave gets there).
__host__ __device__ constexpr int pipe_sync_ints(int K) { return 8 + 8 * K; }
constexpr int PIPE_SPIN_LIMIT = 1 << 21;      // default of StripParams.pipeSpinLimit: ~3 s of polling

template <class S, int R, bool PIPE>
__global__ __launch_bounds__(64) void msa_fill_strip_kernel(const StripParams p) {
    extern __shared__ int lds[];
    int *colHl = lds;                                                // [c] = horizLimit[c] + ONE
    uint8_t *colRef = reinterpret_cast<uint8_t *>(lds + p.maxColumns + 2);   // [c] = reference byte of column c (5 bytes of LDS per column)
    constexpr int ONE = 1 << S::OFF;                                 // one score unit: a packed cell p = score|time satisfies
                                                                     // score <= L  <=>  p < L + ONE for bounds that are multiples of ONE
    constexpr int STRIP = 64 * R;
    const int lane = threadIdx.x;
    const int K = PIPE ? p.pipeK : 1;
    const long long slot = PIPE ? blockIdx.x / K : blockIdx.x;
    const int w = PIPE ? (int)(blockIdx.x % K) : 0;                  // this wave's strip
    unsigned *dirSlot = p.dirbuf + slot * p.dir_slot_dwords;
    int *bnd = PIPE ? p.pipeBoundary + slot * (long long)K * 3 * (p.maxColumns + 2) : p.boundary + slot * 6LL * (p.maxColumns + 2);
    uint8_t *tmp = p.tmpbuf + slot * (long long)(p.maxRows + p.maxColumns + 8);
    const long long total = job_count(p.njobs, p.njobs_dev);
    int *sync = PIPE ? p.pipeSync + slot * (long long)pipe_sync_ints(K) : nullptr;
    int *syProg = nullptr, *syFirst = nullptr, *syLast = nullptr, *syState = nullptr, *syNoEnter = nullptr, *syIters = nullptr;
    if (PIPE) {
        syProg = sync + 8; syFirst = sync + 8 + K; syLast = sync + 8 + 2 * K; syState = sync + 8 + 3 * K; syNoEnter = sync + 8 + 4 * K;
        syIters = sync + 8 + 5 * K;                                  // (two ints per strip)
    }
    // polls a hand-shake word until it reaches `need`; false after pipeSpinLimit polls or as soon as the slot is dead
    auto wait_ge = [&](const int *word, int need) -> bool {
        for (int spin = 0; spin < p.pipeSpinLimit; spin++) {
            if (ld_agent(word) >= need) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return true; }
            if (ld_agent(sync + 3)) return false;
            __builtin_amdgcn_s_sleep(8);
        }
        return false;
    };
    // raises the slot's claim word from q to q + 1: true for exactly one caller per job
    auto claim = [&](long long q) -> bool {
        int got = 0;
        if (lane == 0) {
            int expected = (int)q;
            got = __hip_atomic_compare_exchange_strong(sync + 6, &expected, (int)q + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        }
        return __builtin_amdgcn_readfirstlane(got) != 0;
    };
    // a wave that gave up waiting: the slot is dead from here on; every job of it that nobody has claimed yet goes to the one-thread kernel
    auto drain = [&]() {
        if (lane == 0) __hip_atomic_store(sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const long long q = (long long)__builtin_amdgcn_readfirstlane(ld_agent(sync + 6));
            const long long jq = slot + q * p.pipeSlots;
            if (jq >= total) break;
            if (claim(q) && lane == 0) { const unsigned k = atomicAdd(p.slow_count, 1u); p.slow_list[k] = (int)jq; }
        }
    };

    for (long long kjob = 0;; kjob++) {
        long long j;
        if (PIPE) {
            // (a dead slot is drained before anything else: the job that failed may have been the slot's last)
            if (__builtin_amdgcn_readfirstlane(ld_agent(sync + 3))) { drain(); break; }
            j = slot + kjob * p.pipeSlots;
            if (j >= total) break;
            // no wave starts a job before ALL the slot's waves are through with the one before (a straggler still writing the hand-shake
            // words of job kjob - 1 would corrupt job kjob); a wave that waits in vain declares the slot dead and hands its jobs on
            if (!wait_ge(sync, (int)kjob)) { drain(); break; }
        } else {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(p.queue, 1u);
            j = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (j >= total) break;
        }
        // PIPE: this wave is through with job kjob (finished, declined or failed).  The last of the slot's K waves to arrive puts the
        // per-strip words back to zero and lets everybody go on to the next job.
        auto arrive = [&]() {
            if (PIPE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                int last = 0;
                if (lane == 0) last = (__hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == K * ((int)kjob + 1)) ? 1 : 0;
                if (__builtin_amdgcn_readfirstlane(last)) {
                    for (int i = lane; i < 8 * K; i += 64) __hip_atomic_store(sync + 8 + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0) __hip_atomic_store(sync, (int)kjob + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };

        // ------------------------------------------------------------------ job setup (as msa_fill_fast.hip)
        const bbmsa_job jb = p.jobs[j];
        const int rows = jb.read_len;
        int a = jb.refStartLoc, b = jb.refEndLoc;
        const int mode = jb.flags & BBMSA_MODE_MASK;
        if (jb.flags & BBMSA_CLAMP_WINDOW) {
            a = max(0, a);
            b = min(jb.ref_len - 1, b);
            if (b - a >= p.maxColumns) b = min(jb.ref_len - 1, a + p.maxColumns - 1);
        }
        const int columns = b - a + 1;
        const bool shapeOK = rows >= 1 && columns >= 1 && rows <= p.maxRows && columns <= p.maxColumns;
        int halfband = 0;
        if (!(p.bandwidth < 1 && p.bandwidthRatio <= 0.0f)) {
            const int bwA = p.bandwidth < 1 ? 9999999 : p.bandwidth;
            const int bwB = p.bandwidthRatio <= 0.0f ? 9999999 : 8 + (int)__fmul_rn((float)rows, p.bandwidthRatio);
            halfband = max(min(bwA, bwB), columns - rows + 8) / 2;
        }
        int minScore = jb.minScore;
        bool limited;
        if (mode == BBMSA_FILL_UNLIMITED_RAW) limited = false;
        else if (mode == BBMSA_FILL_LIMITED_RAW) limited = true;
        else {
            if (minScore < 1 || (columns + rows < 90) ||
                ((halfband < 1 || halfband * 3 > columns) && (columns > rows + min(170, rows + 20)))) limited = false;
            else { limited = true; minScore -= 120; }
        }
        const bool banded = limited && halfband > 0;
        if (!shapeOK) {
            if (w == 0 && (!PIPE || claim(kjob)) && lane == 0) {
                bbmsa_result r;
                for (int i = 0; i < 5; i++) r.result[i] = 0;
                r.status = BBMSA_ST_BAD_SHAPE; r.iterations = 0;
                for (int i = 0; i < 8; i++) r.score[i] = 0;
                r.score_len = 0; r.match_len = 0; r.fill_kind = 0; r.columns = columns;
                p.results[j] = r;
            }
            arrive();
            continue;
        }
        // (Windows narrower than the read need no special case here: the per-plane priority of the "still needed" penalties --
        // deletions first in the match and insertion planes, insertions first in the deletion plane -- is spelled out per cell below.
        // Round 2 handed them to the one-thread kernel as the 11ts wavefront kernel does, whose single penalty table cannot hold
        // both: a 6,000-base piece with a few more inserted than deleted bases then cost 12 s on one thread.)
        if (banded) {                                               // the generic kernel takes these
            if (w == 0 && (!PIPE || claim(kjob)) && lane == 0) { const unsigned k = atomicAdd(p.slow_count, 1u); p.slow_list[k] = (int)j; }
            arrive();
            continue;
        }

        const uint8_t *rd = p.reads + jb.read_off;
        const uint8_t *rf = p.refs + jb.ref_off + a;                 // rf[c-1] is the reference byte of column c
        const int maxGain = (rows - 1) * S::MATCH2 + S::MATCH;
        const int minScoreOff = minScore * ONE;
        const int floorv = limited ? minScoreOff - maxGain : kNegInf;
        const int subfloor = limited ? floorv - 5 * S::MATCH2 : 0 - 2 * maxGain;
        const int floorP = floorv + ONE;
        const bool notLimited = !limited;
        const int nstrips = (rows + STRIP - 1) / STRIP;

        // column info: reference bytes by the whole wave, horizLimit by lane 0 (jni/...c:427-438)
        for (int c = lane + 1; c <= columns; c += 64) colRef[c] = rf[c - 1];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            int h = minScoreOff;
            bool prevDef = false;
            for (int i = columns - 1; i >= 0; i--) {
                colHl[i + 1] = (limited ? h : kNegInf) + ONE;
                const int cb = colRef[i + 1];
                const bool def = fully_defined(cb);
                const int cost = def ? (prevDef ? S::MATCH2 : S::MATCH) : ((prevDef && cb == '-') ? S::DEL : 0);
                h = max(h - cost, floorv);
                prevDef = def;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // vertLimit needs, per row, the gain still available from the bases after it: cost of base i = defined ?
        // (next defined ? MATCH2 : MATCH) : 0 (jni/...c:413-425).  suffixAfter = that sum over the strips below the current one.
        auto base_cost = [&](int i) -> int {                          // i = 0-based base index
            if (i >= rows || !fully_defined(rd[i])) return 0;
            return (i + 1 < rows && fully_defined(rd[i + 1])) ? S::MATCH2 : S::MATCH;
        };
        // first / last good column of the row above the current strip, visited-cell bookkeeping, last-row argmax
        int bMin = 1, bMax = columns;                                 // row 0: the reference starts with minGoodCol 1, maxGoodCol columns
        long long iters = 0;
        int noEnterRow = INT_MAX;                                     // first row the limited fill does not enter
        int lastColStart = 1, lastHasGood = 0;
        int bestM = 0, bestD = 0, bestI = 0, bestMc = -1, bestDc = -1, bestIc = -1;

        bool failed = false;                                          // PIPE: a hand-shake timed out
        bool deadAbove = false;                                       // PIPE: the strip above ended without a good cell in its last row
        if (PIPE && w >= nstrips) { arrive(); continue; }             // fewer strips than waves: nothing to do for this job
        for (int s = PIPE ? w : 0; s < (PIPE ? w + 1 : nstrips); s++) {
            const int rowBase = s * STRIP;                            // rows rowBase+1 .. rowBase+STRIP
            const int r0 = rowBase + lane * R + 1;
            unsigned *dir = dirSlot + (long long)s * p.dir_strip_dwords;
            const int W = p.maxColumns + 2;
            // boundary row written by strip s-1 / for strip s+1: ping-pong in the sequential form, one row set per strip when pipelined
            const int *bIn = PIPE ? bnd + (long long)(s > 0 ? s - 1 : 0) * 3 * W : bnd + (s & 1) * 3 * W;
            int *bOut = PIPE ? bnd + (long long)s * 3 * W : bnd + ((s + 1) & 1) * 3 * W;
            int fgPrev = -1;                                          // PIPE: first good column of the row above, as far as it is published
            if (PIPE) {
                if (lane == 63) { const int v = S::col0(min(rowBase + STRIP, rows)); bOut[0] = v; bOut[W] = v; bOut[2 * W] = v; }
                if (lane == 0) __hip_atomic_store(syFirst + s, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s > 0 && !wait_ge(syProg + s - 1, min(64, columns))) { failed = true; break; }
            }

            int call1[R], vlimP[R], delForce[R], insHiForce[R], mPrev[R];
            bool rowValid[R];
            int call0First = '?';
            {
                if (r0 >= 2 && r0 - 2 < rows) call0First = rd[r0 - 2];
                int cst[R], laneSum = 0;
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const int row = r0 + k;
                    rowValid[k] = row <= rows;
                    call1[k] = rowValid[k] ? rd[row - 1] : 0;
                    cst[k] = base_cost(row);                          // base index `row` = the first base after this row
                    laneSum += cst[k];
                }
                int inc = laneSum;                                    // inclusive suffix sum over the lanes
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(inc, d, 64); if (lane + d < 64) inc += o; }
                // the bases of the strips below: summed by the whole wave once per strip (a few passes over the read)
                int below = 0;
                for (int i = rowBase + STRIP + 1 + lane; i < rows; i += 64) below += base_cost(i);
                for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d, 64);
                // (cost indices: row r's limit counts bases r .. rows-1; lane sums cover r0 .. rowBase+STRIP, `below` the rest)
                int suffix = inc - laneSum + below;
#pragma unroll
                for (int k = R - 1; k >= 0; k--) {
                    suffix += cst[k];
                    const int v = limited ? max(minScoreOff - suffix, floorv) : kNegInf;
                    vlimP[k] = rowValid[k] ? v + ONE : (1 << 30);
                }
            }
            int pM[R], pD[R], pI[R], minGood[R], maxGood[R];
            unsigned dacc[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                const int row = r0 + k;
                const int c0v = S::col0(min(row, rows));
                pM[k] = c0v; pD[k] = c0v; pI[k] = c0v;
                minGood[k] = -1; maxGood[k] = -2; dacc[k] = 0;
                mPrev[k] = (call1[k] == '!') ? 8 : 0;
                delForce[k] = (row < S::BAR_D1 || row > rows - S::BAR_D1) ? INT_MAX : INT_MIN;
                insHiForce[k] = (row > rows - S::BAR_I1) ? INT_MAX : INT_MIN;
            }
            const bool lead = lane == 0;
            const bool rowOneLow = (r0 < S::BAR_I1);                  // slot 0 of lane 0 in strip 0 when BAR_I1 > 1 (never for 9PacBio)
            const int rowAbove0 = S::col0(r0 - 1);                    // column 0 of the row above my first row (0 for row 0)
            int svM = rowBase == 0 && lead ? 0 : rowAbove0, svD = svM, svI = svM;
            if (lead && s > 0) { svM = ld_agent(bIn); svD = ld_agent(bIn + W); svI = ld_agent(bIn + 2 * W); }
            int lastRef = '!';
            const int lastSlot = (rows - 1) % R;
            const bool ownerLaneFlag = (rows - 1) / STRIP == s && lane == ((rows - 1) % STRIP) / R;
            const bool lastLane = lane == 63;
            const int steps = columns + 63;
            int bufM = 0, bufD = 0, bufI = 0;                         // 64 columns of the boundary row above, one per lane

            for (int t = 1; t <= steps; t++) {
                if (s > 0 && ((t - 1) & 63) == 0) {                   // columns t .. t+63 of the previous strip's last row
                    if (PIPE) {
                        if (t <= columns && !wait_ge(syProg + s - 1, min(t + 63, columns))) { failed = true; break; }
                        fgPrev = ld_agent(syFirst + s - 1);
                        const int stAbove = ld_agent(syState + s - 1);
                        if (stAbove == 3) { failed = true; break; }
                        if (stAbove == 2 && limited) { deadAbove = true; break; }      // nothing below that row is ever entered
                    }
                    const int cc = min(t + lane, columns);
                    bufM = ld_agent(bIn + cc); bufD = ld_agent(bIn + W + cc); bufI = ld_agent(bIn + 2 * W + cc);
                }
                const int c = t - lane;
                if (c >= 1) {
                    const bool inRange = c <= columns;
                    const int cc = min(c, columns);
                    const int hlP = colHl[cc], ref1 = colRef[cc];
                    const int ref0 = c < 2 ? '!' : lastRef;
                    const bool gap = ref1 == '-', refN = ref1 == 'N';
                    const int refPen = refN ? S::DEL_REF_N : (gap ? S::GAP : 0);
                    const int insNeededBase = (columns - c) + 1;
                    const bool cGt1 = c > 1;
                    const int cLtLastForce = (c < columns - 1) ? INT_MAX : INT_MIN;
                    const unsigned sh = (unsigned)(t & 7) * 4u;

                    int upM = lane_up(pM[R - 1], 0), upD = lane_up(pD[R - 1], 0), upI = lane_up(pI[R - 1], 0);
                    int upMin = lane_up(minGood[R - 1], 1);
                    if (lead) {
                        if (s == 0) { upM = 0; upD = 0; upI = 0; upMin = 1; }                // row 0 is all zero
                        else {
                            // lane 0 is at column t: entry (t - 1) & 63 of the buffered boundary columns
                            const int src = (t - 1) & 63;
                            upM = __builtin_amdgcn_readlane(bufM, src); upD = __builtin_amdgcn_readlane(bufD, src); upI = __builtin_amdgcn_readlane(bufI, src);
                            if (PIPE) upMin = (fgPrev >= 0 && c >= fgPrev) ? fgPrev : -1;
                            else upMin = (bMin >= 0 && c >= bMin) ? bMin : -1;
                        }
                    }
                    int dgM = svM, dgD = svD, dgI = svI;
                    svM = upM; svD = upD; svI = upI;
                    bool started = upMin >= 0;
                    int pm8 = ((call0First == ref0) & (ref0 != 'N')) ? 8 : 0;
                    const SpelledPen<S> pen;
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const int row = r0 + k;
                        CellIn ci;                                             // the cell itself: msa_cell.h, shared with the wavefront kernel
                        ci.row = row; ci.c = c; ci.rows = rows; ci.insNeededBase = insNeededBase;
                        ci.cl1 = call1[k]; ci.ref1 = ref1; ci.refN = refN; ci.gap = gap; ci.match = (call1[k] == ref1) & !refN;
                        ci.act = inRange & (started | notLimited);
                        ci.refPen = refPen; ci.limitP = max(vlimP[k], hlP); ci.floorP = floorP; ci.subfloor = subfloor;
                        ci.dgM = dgM; ci.dgD = dgD; ci.dgI = dgI; ci.lM = pM[k]; ci.lD = pD[k]; ci.upM = upM; ci.upI = upI;
                        ci.delForce = delForce[k];
                        ci.insForce = max((k == 0 && rowOneLow && cGt1) ? INT_MAX : INT_MIN, min(insHiForce[k], cLtLastForce));
                        ci.pm8 = pm8;
                        const CellOut co = cell_update<S, true>(pen, ci);
                        dacc[k] |= co.nib << sh;

                        const bool good = co.goodM | co.goodD | co.goodI;
                        minGood[k] = (good & (minGood[k] < 0)) ? c : minGood[k];
                        maxGood[k] = good ? c : maxGood[k];
                        dgM = pM[k]; dgD = pD[k]; dgI = pI[k];
                        pM[k] = co.nM; pD[k] = co.nD; pI[k] = co.nI;
                        upM = co.nM; upI = co.nI;
                        started = minGood[k] >= 0;
                        pm8 = mPrev[k];
                        mPrev[k] = co.mb8;
                    }
                    lastRef = ref1;

                    if (lastLane && inRange) {                        // the strip's last row: next strip's boundary
                        bOut[c] = pM[R - 1]; bOut[W + c] = pD[R - 1]; bOut[2 * W + c] = pI[R - 1];
                    }
                    if (PIPE && s + 1 < nstrips) {                    // publish the row 64 columns at a time (and its end)
                        const int c63 = t - 63;
                        if (c63 >= 1 && c63 <= columns && ((c63 & 63) == 0 || c63 == columns)) {
                            const int fg = __builtin_amdgcn_readlane(minGood[R - 1], 63);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (lane == 0) __hip_atomic_store(syFirst + s, fg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                            if (lane == 0) __hip_atomic_store(syProg + s, c63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    {   // last row of the read: first strict maximum per plane, ascending column
                        int lm = pM[0], ld = pD[0], li = pI[0];
#pragma unroll
                        for (int k = 1; k < R; k++) { lm = (lastSlot == k) ? pM[k] : lm; ld = (lastSlot == k) ? pD[k] : ld; li = (lastSlot == k) ? pI[k] : li; }
                        const bool track = ownerLaneFlag & inRange;
                        const bool um = track & ((bestMc < 0) | ((lm & S::SMASK) > (bestM & S::SMASK)));
                        const bool ud = track & ((bestDc < 0) | ((ld & S::SMASK) > (bestD & S::SMASK)));
                        const bool ui = track & ((bestIc < 0) | ((li & S::SMASK) > (bestI & S::SMASK)));
                        bestM = um ? lm : bestM; bestMc = um ? c : bestMc;
                        bestD = ud ? ld : bestD; bestDc = ud ? c : bestDc;
                        bestI = ui ? li : bestI; bestIc = ui ? c : bestIc;
                    }
                }
                if ((t & 7) == 7) {
                    const long long o = (long long)(t >> 3) * R * 64 + lane;
#pragma unroll
                    for (int k = 0; k < R; k++) { dir[o + (long long)k * 64] = dacc[k]; dacc[k] = 0; }
                }
            }
            if ((steps & 7) != 7) {
                const long long o = (long long)(steps >> 3) * R * 64 + lane;
#pragma unroll
                for (int k = 0; k < R; k++) dir[o + (long long)k * 64] = dacc[k];
            }
            if (!PIPE && lastLane) { const int v = S::col0(min(rowBase + STRIP, rows)); bOut[0] = v; bOut[W] = v; bOut[2 * W] = v; }
            if (PIPE && failed) break;
            if (PIPE && deadAbove) {                                  // not one row of this strip is entered
#pragma unroll
                for (int k = 0; k < R; k++) { minGood[k] = -1; maxGood[k] = -2; }
            }
            if (PIPE) {                                               // the row above, final: first / last good column
                if (s > 0) {
                    if (!wait_ge(syState + s - 1, 1)) { failed = true; break; }
                    if (ld_agent(syState + s - 1) == 3) { failed = true; break; }
                    bMin = ld_agent(syFirst + s - 1); bMax = ld_agent(syLast + s - 1);
                } else { bMin = 1; bMax = columns; }
            }

            // ---- row extents of this strip -> iterations, first row not entered (jni/...c:441-449, :660-661)
            {
                int pMin = lane_up(minGood[R - 1], 1), pMax = lane_up(maxGood[R - 1], columns);
                if (lead) { pMin = bMin; pMax = bMax; }
                long long it = 0;
                int firstNoEnter = INT_MAX;
                long long itersRow[R];
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const int row = r0 + k;
                    itersRow[k] = 0;
                    if (rowValid[k]) {
                        const int hasGood = minGood[k] >= 0;
                        const int colStart = pMin, colStop = pMax;
                        const bool enter = pMin >= 0 && colStart >= 0 && colStop >= colStart;
                        if (!enter && firstNoEnter == INT_MAX) firstNoEnter = row;
                        const int endc = min(columns, max(colStop, hasGood ? maxGood[k] : -2) + 1);
                        itersRow[k] = (long long)(endc - colStart + 1);
                        if (row == rows) { lastColStart = colStart; lastHasGood = hasGood; }
                    }
                    pMin = minGood[k]; pMax = maxGood[k];
                }
                int g = firstNoEnter;
                for (int d = 1; d < 64; d <<= 1) g = min(g, __shfl_xor(g, d, 64));
                noEnterRow = min(noEnterRow, g);
#pragma unroll
                for (int k = 0; k < R; k++) if (rowValid[k] && r0 + k < noEnterRow) it += itersRow[k];
                for (int d = 1; d < 64; d <<= 1) it += __shfl_xor(it, d, 64);
                iters += it;
            }
            // the last row's extents for the next strip
            bMin = __builtin_amdgcn_readlane(minGood[R - 1], 63);
            bMax = __builtin_amdgcn_readlane(maxGood[R - 1], 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the boundary row is in L2 before the next strip reads it
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __builtin_amdgcn_wave_barrier();
            if (PIPE) {                                               // this strip's share of the bookkeeping, then "done"
                if (lane == 0) {
                    __hip_atomic_store(syFirst + s, bMin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(syLast + s, bMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(syNoEnter + s, noEnterRow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(syIters + 2 * s, (int)(iters & 0xffffffffLL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(syIters + 2 * s + 1, (int)(iters >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) {
                    __hip_atomic_store(syProg + s, columns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(syState + s, (limited && bMin < 0) ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!PIPE && limited && noEnterRow <= rows) break;       // the fill died inside this strip: no row below is entered
        }
        if (PIPE) {
            if (failed) {
                // a hand-shake timed out (or the slot died under us): tell the waves below, declare the slot dead -- its unclaimed jobs,
                // this one included, go to the one-thread kernel (drain, at the top of the loop) -- and arrive
                if (lane == 0) {
                    __hip_atomic_store(syState + w, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                arrive();
                continue;
            }
            if (w != nstrips - 1) { arrive(); continue; }             // only the wave of the last strip goes on to the result
            {
                // every strip is done (a strip finishes only after the one above it): first row not entered, visited cells
                int g = INT_MAX; long long its = 0;
                if (lane < nstrips) {
                    g = ld_agent(syNoEnter + lane);
                    its = (long long)(unsigned)ld_agent(syIters + 2 * lane) | ((long long)ld_agent(syIters + 2 * lane + 1) << 32);
                }
                const unsigned long long finite = __ballot(g != INT_MAX);
                const int sstar = finite ? __builtin_ctzll(finite) : 64;     // strips below the first one with an unentered row add nothing
                if (lane > sstar) its = 0;
                for (int d = 32; d >= 1; d >>= 1) { g = min(g, __shfl_xor(g, d, 64)); its += __shfl_xor(its, d, 64); }
                noEnterRow = g; iters = its;
            }
            // the job's result is this wave's to write -- unless the slot died meanwhile and a draining wave has claimed the job
            if (!claim(kjob)) { arrive(); continue; }
        }
        if (!limited) iters = (long long)rows * columns;

        // ------------------------------------------------------------------ result[] (as msa_fill_fast.hip)
        const int ownerLane = ((rows - 1) % STRIP) / R;
        int bScore, bCol, bState, bPacked;
        {
            int sx = bestMc >= 0 ? (bestM & S::SMASK) : INT_MIN, cbest = bestMc, st = 0, pk = bestM;
            if (bestDc >= 0 && (bestD & S::SMASK) > sx) { sx = bestD & S::SMASK; cbest = bestDc; st = 1; pk = bestD; }
            if (bestIc >= 0 && (bestI & S::SMASK) > sx) { sx = bestI & S::SMASK; cbest = bestIc; st = 2; pk = bestI; }
            bScore = __shfl(sx, ownerLane, 64); bCol = __shfl(cbest, ownerLane, 64); bState = __shfl(st, ownerLane, 64); bPacked = __shfl(pk, ownerLane, 64);
            lastColStart = __shfl(lastColStart, ownerLane, 64); lastHasGood = __shfl(lastHasGood, ownerLane, 64);
        }
        int res1, res2, res3, res4 = 0;
        bool fillNull = false;
        if (!limited) { res1 = bCol; res2 = bState; res3 = bScore >> S::OFF; }
        else if (noEnterRow <= rows) { res1 = 1; res2 = 0; res3 = S::BADOFF; res4 = 1; fillNull = true; }
        else if (!lastHasGood) { res1 = max(1, lastColStart - 1); res2 = 0; res3 = subfloor; res4 = 1; fillNull = true; }
        else if (bScore < minScoreOff) { res1 = bCol; res2 = bState; res3 = bScore; res4 = 1; fillNull = true; }
        else { res1 = bCol; res2 = bState; res3 = bScore >> S::OFF; }

        // ------------------------------------------------------------------ score2 + traceback2 on the records
        const bool wantScore = !fillNull && (jb.flags & BBMSA_DO_SCORE);
        const bool wantTrace = !fillNull && (jb.flags & BBMSA_DO_TRACEBACK) && p.match != nullptr;
        int sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int scoreLen = 0, matchLen = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        auto nibble = [&](int row, int col) -> unsigned {
            const int s = (row - 1) / STRIP, lr = (row - 1) - s * STRIP;
            const int ol = lr / R, ok = lr - ol * R, ot = col + ol;
            const unsigned dw = load_coherent(dirSlot + (long long)s * p.dir_strip_dwords + ((long long)(ot >> 3) * R + ok) * 64 + ol);
            return (dw >> ((ot & 7) * 4)) & 15u;
        };
        if (wantScore || wantTrace) {
            int row = rows, col = res1, state = res2, n = 0, gapSyms = 0, stateTime = 0;
            while (row > 0 && col > 0) {
                if (state == 0) {
                    const int rr = row - lane, cq = col - lane;       // diagonal run: lane l looks at cell (row-l, col-l)
                    const bool inside = rr >= 1 && cq >= 1;
                    const unsigned nibv = inside ? nibble(rr, cq) : 0u;
                    const bool brk = !inside || (nibv & 3u) != 0u;
                    const unsigned long long bal = __ballot(brk);
                    const int fb = bal ? __builtin_ctzll(bal) : 64;
                    const int fbInside = __shfl((int)inside, min(fb, 63), 64);
                    const int fbPrev = __shfl((int)(nibv & 3u), min(fb, 63), 64);
                    const int consumed = (fb < 64 && fbInside) ? fb + 1 : fb;
                    if (wantTrace && lane < consumed) {
                        const int cb = rd[rr - 1], rb = colRef[cq];
                        tmp[n + lane] = (cb == rb) ? 'm' : ((!fully_defined(cb) || !fully_defined(rb)) ? 'N' : 'S');
                    }
                    stateTime += fb;
                    if (fb < 64 && fbInside) { stateTime = 0; state = fbPrev; }
                    row -= consumed; col -= consumed; n += consumed;
                } else {
                    const unsigned nibv = nibble(row, col);
                    int prev;
                    if (state == 1) {
                        prev = (nibv & 4u) ? 1 : 0;
                        const int rb = colRef[col];
                        if (wantTrace && lane == 0) tmp[n] = (rb == '-') ? '-' : 'D';
                        if (rb == '-') gapSyms++;
                        col--;
                    } else {
                        prev = (nibv & 8u) ? 2 : 0;
                        if (wantTrace && lane == 0) tmp[n] = (col >= columns) ? 'Y' : 'I';
                        row--;
                    }
                    n++;
                    if (prev == state) stateTime++; else stateTime = 0;
                    state = prev;
                }
            }
            int colS = col;
            if (row > colS) colS -= row;
            const int bestRefStart = a + colS, bestRefStop = a + res1 - 1;
            int padLeft = 0, padRight = 0;
            if (bestRefStart < a) padLeft = max(0, a - bestRefStart);
            else if (bestRefStart == a && state == 2) padLeft = stateTime;
            const int bW = (jb.flags & BBMSA_INTERNAL_GAPPED) ? jb.ref_len : b;
            if (bestRefStop > bW) padRight = max(0, bestRefStop - bW);
            else if (bestRefStop == bW && res2 == 2) padRight = bPacked & S::TMASK;
            if (wantScore) {
                sc[0] = bScore >> S::OFF; sc[1] = bestRefStart; sc[2] = bestRefStop; sc[3] = rows; sc[4] = res1; sc[5] = res2;
                sc[6] = padLeft; sc[7] = padRight;
                scoreLen = (padLeft > 0 || padRight > 0) ? 8 : 6;
                if (scoreLen == 6) { sc[6] = 0; sc[7] = 0; }
            }
            if (wantTrace) {
                const int xs = (col != row) ? row : 0;                // leftover read bases become 'X'
                for (int i = lane; i < xs; i += 64) tmp[n + i] = 'X';
                n += xs;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                __builtin_amdgcn_wave_barrier();
                uint8_t *out = p.match + j * (long long)p.match_stride;
                const bool keepGaps = (jb.flags & BBMSA_TRACE_KEEP_GAPS) != 0;          // leave each '-' in the string (the caller expands)
                const int totalLen = keepGaps ? n : n + gapSyms * (kGapLen - 1);
                if (totalLen > p.match_stride) matchLen = -1;
                else if (gapSyms == 0 || keepGaps) {
                    for (int i = lane; i < n; i += 64) out[i] = ld_agent_u8(tmp + n - 1 - i);
                    matchLen = n;
                } else {
                    if (lane == 0) {
                        int o = 0;
                        for (int i = n - 1; i >= 0; i--) {
                            const uint8_t ch = ld_agent_u8(tmp + i);
                            if (ch != '-') out[o++] = ch;
                            else for (int q = 0; q < kGapLen; q++) out[o++] = 'D';
                        }
                    }
                    matchLen = totalLen;
                }
            }
        }
        if (lane == 0) {
            bbmsa_result r;
            r.result[0] = rows; r.result[1] = res1; r.result[2] = res2; r.result[3] = res3; r.result[4] = res4;
            r.status = (fillNull && mode == BBMSA_FILL_LIMITED) ? BBMSA_ST_NULL : BBMSA_ST_OK;
            r.iterations = iters;
            for (int i = 0; i < 8; i++) r.score[i] = sc[i];
            r.score_len = scoreLen; r.match_len = matchLen;
            r.fill_kind = limited ? 0 : 1; r.columns = columns;
            p.results[j] = r;
        }
        __builtin_amdgcn_wave_barrier();
        arrive();
    }
}

constexpr int kStripR = 8;           // rows per lane: strips of 512 rows
template __global__ void msa_fill_strip_kernel<Scheme9PacBio, kStripR, false>(const StripParams);
template __global__ void msa_fill_strip_kernel<Scheme9PacBio, kStripR, true>(const StripParams);
// (Round 3: a 2-rows-per-lane instantiation for launches with few jobs was measured and dropped: a lone 6,000 x 6,100 fill takes
// 370 ms at R = 8 and 411 ms at R = 2 -- per step about 1 us of fixed cost plus 0.5 us per row of the lane's serial chain, so fewer
// rows per lane only multiply the steps.  A lone fill is bound by its dependent instruction chain (one wavefront issues an
// instruction every ~8 cycles); only spreading the strips of one job over several wavefronts would shorten it.)
int strip_rows_per_lane() { return kStripR; }
const void *strip_kernel_pacbio() { return (const void *)msa_fill_strip_kernel<Scheme9PacBio, kStripR, false>; }
const void *strip_kernel_pacbio_pipelined() { return (const void *)msa_fill_strip_kernel<Scheme9PacBio, kStripR, true>; }
int strip_pipe_sync_ints(int K) { return pipe_sync_ints(K); }


}  // namespace bbmsa
