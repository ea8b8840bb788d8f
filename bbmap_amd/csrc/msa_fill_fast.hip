// Fast path of the MultiStateAligner11ts DP for gfx950 (MI355X).
//
// One alignment job is spread over G = 16/32/64 lanes of a 64-wide wavefront (64/G jobs per
// wave).  Lane `gl` of a job owns R consecutive DP rows (rows gl*R+1 .. gl*R+R) and sweeps
// them left to right; lane gl is one column behind lane gl-1, so every step of the wave
// processes one anti-diagonal band of the matrix ("wavefront scan").  The three cell values a
// row needs from the row above arrive with one DPP wave_shr:1 per plane; everything else a
// lane needs is in its own registers.  No DP cell ever touches HBM: the only per-cell output
// is a 4-bit traceback direction record (time>1 / predecessor compare results), streamed to a
// per-job scratch slot with coalesced dword stores, 8 columns per store.
//
// What is reproduced bit for bit (see DESIGN.md for the argument):
//  * fillUnlimited  (jni/MultiStateAligner11tsJNI.c:100-314)
//  * fillLimitedX without a band (jni/...c:361-704): the score-pruned column window is
//    re-derived from per-row first/last "good" columns; cells past a row's true end are
//    provably `subfloor`, so over-computing them changes no value, and the visited-cell count
//    (`iterations`) is recovered exactly from the good-column extents.
//  * traceback2 / score2 (current/align2/MultiStateAligner11tsJNI.java:376-495, :537-658),
//    run on the direction records instead of the `packed` matrix.
// Banded fills (MSA.bandwidth / bandwidthRatio > 0) and shapes beyond this kernel's limits are
// handed to the generic kernel (msa_fill_generic.hip) through `slow_list`.
#include "msa_common.h"
#include "msa_cell.h"

namespace bbmsa {

// value of lane-1 (0x138 = DPP wave_shr:1); lanes with no source keep `fill`
__device__ __forceinline__ int lane_up(int x, int fill) {
    return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int clamp_time(int t) { return t > kMaxTime ? kMaxTime - 3 : t; }

// jni/...c:229-233: per-step deletion extension cost as a function of the streak
__device__ __forceinline__ int del_extend(int streak) {
    int c = (streak & 3) == 0 ? P_DEL5 : 0;
    c = streak < 80 ? P_DEL4 : c;
    c = streak < 20 ? P_DEL3 : c;
    c = streak < 5 ? P_DEL2 : c;
    c = streak == 0 ? P_DEL : c;
    return c;
}
// POINTSoff_INS_ARRAY[streak+1], MultiStateAligner11tsJNI.java:1582-1598
__device__ __forceinline__ int ins_extend(int streak) {
    int c = P_INS4;
    c = streak < 20 ? P_INS3 : c;
    c = streak < 5 ? P_INS2 : c;
    c = streak == 0 ? P_INS : c;
    return c;
}
// POINTSoff_SUB_ARRAY[streak+1], :1608-1621
__device__ __forceinline__ int sub_extend(int streak) {
    int c = P_SUB3;
    c = streak < 5 ? P_SUB2 : c;
    c = streak == 0 ? P_SUB : c;
    return c;
}

__device__ __forceinline__ unsigned load_coherent(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// waves per SIMD the register allocator must leave room for (measured on MI355X at R=5: 3 waves beat 2 by 27 %; with the
// context-sized LDS tables a 4th block fits a CU and 4 waves beat 3 by 7 %, 5 waves lose it again to spills)
#ifndef BBMSA_MIN_WAVES
#define BBMSA_MIN_WAVES(R) ((R) <= 5 ? 4 : ((R) <= 7 ? 2 : 1))
#endif

// MAT: also write every computed cell's three planes (12 bytes per cell) and the two limit vectors -- what the reference's native
// code leaves in the Java class's `packed` matrix for score2 / traceback2 to walk (jni/MultiStateAligner11tsJNI.c:124-127, :707-812).
// A cell that is visited but not "good" is stored as subfloor | time, as the reference stores it (:556-562); cells this schedule
// computes beyond a row's end are stored as subfloor -- the reference leaves those untouched, and never reads them (its sentinels).
template <int R, bool BANDED, bool MAT = false>
__global__ __launch_bounds__(256, BBMSA_MIN_WAVES(R)) void msa_fill_fast_kernel(const FillParams p) {
    extern __shared__ int lds[];
    int *delC = lds;
    const int TL = p.tableLen;
    int *insC = lds + TL;
    int *delExt = lds + 2 * TL;       // jni/...c:229-233 as a table of the streak (index <= 83)
    int *insExt = delExt + 128;               // POINTSoff_INS_ARRAY[streak+1], index min(streak,20)
    // The match/substitution plane's three case distinctions as ONE 16-byte lookup: index = min(streak, 5) | match << 3 | prevMatch << 4,
    // entry = {points of staying in the plane (jni/...c:497-531), points of entering it from D / I (MATCH or SUB), what the prune
    // test subtracts from the limit (MATCH2 or SUB3, :486), 0}.  Five selects and two compares per cell less than spelling it out.
    int4 *mTab = reinterpret_cast<int4 *>(delExt + 192);
    LdsPen pen; pen.delC = delC; pen.insC = insC; pen.delExt = delExt; pen.insExt = insExt; pen.mTab = mTab;
    for (int i = threadIdx.x; i < TL; i += blockDim.x) {
        delC[i] = calc_del_off(i);
        insC[i] = calc_ins_cum_off(i);
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) delExt[i] = del_extend(i);
    if (threadIdx.x < 32) insExt[threadIdx.x] = ins_extend(threadIdx.x);
    if (threadIdx.x < 32) {
        const int st = threadIdx.x & 7, mt = (threadIdx.x >> 3) & 1, pv = threadIdx.x >> 4;
        int4 e;
        if (mt) { e.x = pv ? P_MATCH2 : P_MATCH; e.y = P_MATCH; e.z = P_MATCH2; }
        else { e.x = pv ? (st <= 1 ? P_SUBR : P_SUB) : sub_extend(min(st, 5)); e.y = P_SUB; e.z = P_SUB3; }
        e.w = 0;
        mTab[threadIdx.x] = e;
    }
    __syncthreads();

    if (p.priority == 3) __builtin_amdgcn_s_setprio(3);
    else if (p.priority == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.priority == 1) __builtin_amdgcn_s_setprio(1);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = p.lanesPerJob;
    const int gl = lane & (G - 1);
    const int sub = lane / G;
    const int jobsPerWave = 64 / G;
    const int groupBase = sub * G;           // first wave lane of my job group

    const int perJobLds = lds_job_ints(p.fastCols, p.tmpBytes);
    int *myLds = lds + lds_table_ints(TL) + (wave * jobsPerWave + sub) * perJobLds;
    int *colHl = myLds;                                                   // [c] = horizLimit[c] + 2048
    uint8_t *colRef = reinterpret_cast<uint8_t *>(myLds + (p.fastCols + 2));     // [c] = reference byte of column c
    uint8_t *tmp = reinterpret_cast<uint8_t *>(myLds + (p.fastCols + 2) + ((p.fastCols + 2 + 3) >> 2));

    const long long slot = ((long long)blockIdx.x * (blockDim.x >> 6) + wave) * jobsPerWave + sub;
    unsigned *dir = p.dirbuf + slot * p.dir_slot_dwords;

    const long long total = p.list ? (long long)load_coherent(p.list_count) : job_count(p.njobs, p.njobs_dev);
    for (;;) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(p.queue, (unsigned)jobsPerWave);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((long long)base >= total) break;
        const long long q = (long long)base + sub;
        const bool valid = q < total;
        const long long j = (p.list && valid) ? (long long)p.list[q] : q;

        // ------------------------------------------------------------------ job setup
        bbmsa_job jb;
        if (valid) jb = p.jobs[j];
        else { jb.read_off = 0; jb.ref_off = 0; jb.read_len = 0; jb.ref_len = 0; jb.refStartLoc = 0; jb.refEndLoc = -1; jb.minScore = 0; jb.flags = 0; }
        const int rows = jb.read_len;
        int a = jb.refStartLoc, b = jb.refEndLoc;
        const int mode = jb.flags & BBMSA_MODE_MASK;
        if (jb.flags & BBMSA_CLAMP_WINDOW) {              // MSA.java:104-105, :118-121
            a = max(0, a);
            b = min(jb.ref_len - 1, b);
            if (b - a >= p.maxColumns) b = min(jb.ref_len - 1, a + p.maxColumns - 1);
        }
        const int columns = b - a + 1;
        const bool shapeOK = valid && rows >= 1 && columns >= 1 && rows <= p.maxRows && columns <= p.maxColumns;

        // halfband: jni/...c:392-393 ; Java gate: MultiStateAligner11tsJNI.java:137-144
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
        const bool fits = rows <= G * R && columns <= p.fastCols && columns >= rows - 2;   // (narrower windows: see the cell)
        const bool needGeneric = shapeOK && !fits;
        const bool run = shapeOK && !needGeneric;
        // Banded fill (jni/...c:441-442): a row may only extend one column past the last good column of the
        // row above, which this schedule learns one column late.  The kernel assumes a row's good columns have
        // no hole of two or more columns; a job that breaks the assumption is flagged and redone exactly by
        // the generic kernel.
        const bool banded = limited && halfband > 0;
        int bandViolation = 0;

        if (gl == 0 && valid) {
            if (!shapeOK) {
                bbmsa_result r;
                for (int i = 0; i < 5; i++) r.result[i] = 0;
                r.status = BBMSA_ST_BAD_SHAPE; r.iterations = 0;
                for (int i = 0; i < 8; i++) r.score[i] = 0;
                r.score_len = 0; r.match_len = 0; r.fill_kind = 0; r.columns = columns;
                p.results[j] = r;
            } else if (needGeneric) {
                const unsigned k = atomicAdd(p.slow_count, 1u);
                p.slow_list[k] = (int)j;
            }
        }

        const uint8_t *rd = p.reads + jb.read_off;
        const uint8_t *rf = p.refs + jb.ref_off + a;       // rf[c-1] is the reference byte of column c
        int *planeM = nullptr; long long planeInts = 0; int planeW = 0;
        if (MAT && valid) { planeW = columns; planeInts = (long long)rows * planeW; planeM = p.planes + p.plane_off[j]; }      // cell (row, c) at (row - 1) * columns + c - 1

        const int maxGain = (rows - 1) * P_MATCH2 + P_MATCH;
        const int minScoreOff = minScore * 2048;
        const int floorv = limited ? minScoreOff - maxGain : kNegInf;
        const int subfloor = limited ? floorv - 5 * P_MATCH2 : 0 - 2 * maxGain;

        // rows owned by this lane: r0 .. r0+R-1
        const int r0 = gl * R + 1;
        int call1[R], vlim[R];
        bool rowValid[R];
        int call0First = '?';
        {
            if (run && r0 >= 2 && r0 - 2 < rows) call0First = rd[r0 - 2];
            // vertLimit: jni/...c:413-425.  cost of base i = defined ? (next defined ? MATCH2 : MATCH) : 0
            int cst[R];
            int laneSum = 0;
            int nextByte = (run && r0 + R < rows) ? rd[r0 + R] : 0;     // base index r0+R (after my last cost index)
            int idxByte[R + 1];
#pragma unroll
            for (int k = 0; k < R; k++) {
                const int row = r0 + k;
                rowValid[k] = run && row <= rows;
                call1[k] = rowValid[k] ? rd[row - 1] : 0;
                idxByte[k] = (run && row < rows) ? rd[row] : 0;          // base index `row` = next row's base
            }
            idxByte[R] = nextByte;
#pragma unroll
            for (int k = 0; k < R; k++) {
                const int i = r0 + k;                                    // base index
                int c = 0;
                if (run && i < rows && fully_defined(idxByte[k]))
                    c = (i + 1 < rows && fully_defined(idxByte[k + 1])) ? P_MATCH2 : P_MATCH;
                cst[k] = c;
                laneSum += c;
            }
            int inc = laneSum;                                           // inclusive suffix sum over the group
            for (int d = 1; d < G; d <<= 1) {
                const int o = __shfl_down(inc, d, 64);
                if (gl + d < G) inc += o;
            }
            int suffix = inc - laneSum;                                  // lanes after me
#pragma unroll
            for (int k = R - 1; k >= 0; k--) {
                suffix += cst[k];
                vlim[k] = limited ? max(minScoreOff - suffix, floorv) : kNegInf;
            }
            if (MAT && run && limited) {                                 // vertLimit[0..rows], jni/...c:413-425
                int *vout = p.limits + p.limits_off[j];
#pragma unroll
                for (int k = 0; k < R; k++) if (r0 + k < rows) vout[r0 + k] = vlim[k];
                if (gl == 0) {
                    vout[rows] = minScoreOff;
                    const int c0 = fully_defined(call1[0]) ? ((rows > 1 && fully_defined(idxByte[0])) ? P_MATCH2 : P_MATCH) : 0;
                    vout[0] = rows > 1 ? max(vlim[0] - c0, floorv) : max(minScoreOff - c0, floorv);
                }
            }
        }

        // column info into LDS: reference bytes by the whole group, horizLimit by its first lane
        if (run) {
            for (int c = gl + 1; c <= columns; c += G) colRef[c] = rf[c - 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (run && gl == 0) {                                            // jni/...c:427-438
            int h = minScoreOff;
            bool prevDef = false;
            for (int i = columns - 1; i >= 0; i--) {
                colHl[i + 1] = (limited ? h : kNegInf) + 2048;      // horizLimit[i+1] + 2048 (see the fill loop)
                const int cb = colRef[i + 1];                         // ref[refStartLoc+i]
                const bool def = fully_defined(cb);
                const int cost = def ? (prevDef ? P_MATCH2 : P_MATCH) : ((prevDef && cb == '-') ? P_DEL : 0);
                h = max(h - cost, floorv);
                prevDef = def;
                if (MAT && limited) p.limits[p.limits_off[j] + rows + 1 + i + 1] = colHl[i + 1] - 2048;      // horizLimit[i + 1]
            }
            if (MAT && limited) p.limits[p.limits_off[j] + rows + 1] = h;                                        // horizLimit[0]
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------------ fill
        // The loop body is written branch-free (selects only) so that the R rows of a lane form one basic block
        // the scheduler can interleave.  Tricks that keep it exact:
        //  * a packed cell p = score|time (score a multiple of 2048, time < 2048) satisfies
        //    score <= L  <=>  p < L + 2048 for any multiple-of-2048 bound L, so prune tests run on packed values
        //    against "limit + 2048" (suffix P below) without masking;
        //  * prevMatch(row, col) is match(row-1, col-1): the bit computed one step earlier for the row above;
        //  * the predecessor traceback2 would pick for a DEL/INS cell is exactly "the extension won the fill's
        //    own comparison" (extension costs are never below the opening cost), so those two record bits are free.
        const int nl = (rows + R - 1) / R;                               // lanes in use
        int steps = run ? columns + nl - 1 : 0;
        for (int d = 32; d >= 1; d >>= 1) steps = max(steps, __shfl_xor(steps, d, 64));

        int pM[R], pD[R], pI[R];         // my rows' cells at the previous column
        int minGood[R], maxGood[R];      // first / last good column of each row (-1 / -2 = none)
        int vlimP[R];                    // vertLimit + 2048; INT_MAX for rows beyond the read (prunes everything)
        int delForce[R];                 // INT_MAX where the deletion barrier applies (row<3 || row>rows-3), else INT_MIN
        int insHiForce[R];               // INT_MAX where row > rows-2 (insertion barrier near the end), else INT_MIN
        unsigned dacc[R];
        int mPrev[R];                    // match(row k, previous column) as 8/0: bit 3 of the mTab index (a VGPR: SGPR masks are scarce)
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int row = r0 + k;
            const int c0v = insC[row];                                   // column 0: cumulative insertion cost, time 0
            pM[k] = c0v; pD[k] = c0v; pI[k] = c0v;
            minGood[k] = -1; maxGood[k] = -2; dacc[k] = 0;
            vlimP[k] = rowValid[k] ? vlim[k] + 2048 : (1 << 30);     // above every reachable packed value
            mPrev[k] = (call1[k] == '!') ? 8 : 0;                        // ref0 of column 1 is '!' (jni/...c:463)
            delForce[k] = (row < 3 || row > rows - 3) ? INT_MAX : INT_MIN;
            insHiForce[k] = (row > rows - 2) ? INT_MAX : INT_MIN;
        }
        const bool rowOne = (r0 == 1);                                   // slot 0 of lane 0 is DP row 1
        const bool notLimited = !limited;
        const bool groupLead = (gl == 0);
        const int floorP = floorv + 2048;
        // row above at the previous column (diag for my first row); starts as its column-0 value
        int svM = insC[r0 - 1], svD = svM, svI = svM;
        int lastRef = '!';
        // last-row argmax (first strict maximum per state); kept by the lane/slot that owns row `rows`
        const int lastSlot = (max(rows, 1) - 1) % R;
        const bool ownerLaneFlag = gl == (max(rows, 1) - 1) / R;
        int bestM = 0, bestD = 0, bestI = 0, bestMc = -1, bestDc = -1, bestIc = -1;

        for (int t = 1; t <= steps; t++) {
            const int c = t - gl;
            // Lanes that have not reached column 1 yet sit this step out under EXEC (their registers keep the
            // column-0 state); lanes past the last column keep running with everything pruned, because the
            // lane below still has to read their final cells through DPP.
            if (c >= 1 && run) {
            const bool inRange = c <= columns;
            const int cc = min(c, columns);
            const int hlP = colHl[cc], ref1 = colRef[cc];                // (horizLimit + 2048)
            const int ref0 = c < 2 ? '!' : lastRef;
            const bool gap = ref1 == '-';
            const bool refN = ref1 == 'N';
            const int refPen = refN ? P_DEL_REF_N : (gap ? P_GAP : 0);
            const int insNeededBase = (columns - c) + 1;                 // insNeeded = (rows-row) - this
            const bool cGt1 = c > 1;
            const int cLtLastForce = (c < columns - 1) ? INT_MAX : INT_MIN;
            const unsigned sh = (unsigned)(t & 7) * 4u;

            // row above, same column (lane gl-1 finished it one step ago); row 0 is all zero
            int upM = lane_up(pM[R - 1], 0);
            int upD = lane_up(pD[R - 1], 0);
            int upI = lane_up(pI[R - 1], 0);
            int upMin = lane_up(minGood[R - 1], 1);                      // >=0 once the row above has a good cell
            upM = groupLead ? 0 : upM; upD = groupLead ? 0 : upD; upI = groupLead ? 0 : upI;
            upMin = groupLead ? 1 : upMin;
            int upMaxG = 0;
            if (BANDED) {
                upMaxG = lane_up(maxGood[R - 1], 0);                     // last good column of the row above so far
                upMaxG = groupLead ? min(columns, 2 * halfband) : upMaxG;
            }
            int dgM = svM, dgD = svD, dgI = svI;
            svM = upM; svD = upD; svI = upI;
            bool started = upMin >= 0;
            int pm8 = ((call0First == ref0) & (ref0 != 'N')) ? 8 : 0;      // prevMatch of my first row, as mPrev holds it

#pragma unroll
            for (int k = 0; k < R; k++) {
                const int row = r0 + k;
                bool act = inRange & (started | notLimited);              // rows beyond the read prune through vlimP
                if (BANDED) act = act && (!banded || (c >= row - halfband && c <= upMaxG + 1));
                CellIn ci_;
                ci_.row = row; ci_.c = c; ci_.rows = rows; ci_.insNeededBase = insNeededBase;
                ci_.cl1 = call1[k]; ci_.ref1 = ref1; ci_.refN = refN; ci_.gap = gap; ci_.match = (call1[k] == ref1) & !refN; ci_.act = act;
                ci_.refPen = refPen; ci_.limitP = max(vlimP[k], hlP); ci_.floorP = floorP; ci_.subfloor = subfloor;
                ci_.dgM = dgM; ci_.dgD = dgD; ci_.dgI = dgI; ci_.lM = pM[k]; ci_.lD = pD[k]; ci_.upM = upM; ci_.upI = upI;
                ci_.delForce = delForce[k];
                ci_.insForce = (k == 0) ? ((rowOne && cGt1) ? INT_MAX : min(insHiForce[k], cLtLastForce)) : min(insHiForce[k], cLtLastForce);
                ci_.pm8 = pm8;
                const CellOut co = cell_update<Scheme11ts, false>(pen, ci_);            // msa_cell.h
                const int nM = co.nM, nD = co.nD, nI = co.nI;
                const bool goodM = co.goodM, goodD = co.goodD, goodI = co.goodI;
                const int mb8 = co.mb8;
                if (MAT) {
                    if (inRange & rowValid[k]) {
                        int *cellp = planeM + (long long)(row - 1) * planeW + (c - 1);
                        // (the reference clamps the stored time at MAX_TIME - MASK5 in every plane, :563, :618, :659)
                        // skipped by the prune test: plain subfloor (:486, :567, :620); computed but below the limit: subfloor | time
                        const int sM = co.pruneM ? subfloor : (goodM ? nM : (subfloor | clamp_cell_time<Scheme11ts>(co.timeM)));
                        const int sD = co.pruneD ? subfloor : (goodD ? nD : (subfloor | clamp_cell_time<Scheme11ts>(co.timeD)));
                        const int sI = co.pruneI ? subfloor : (goodI ? nI : (subfloor | clamp_cell_time<Scheme11ts>(co.timeI)));
                        cellp[0] = sM; cellp[planeInts] = sD; cellp[2 * planeInts] = sI;
                    }
                }
                dacc[k] |= co.nib << sh;

                // ---- bookkeeping
                const bool good = goodM | goodD | goodI;
                if (BANDED) {
                    if (good && banded && minGood[k] >= 0 && c - maxGood[k] >= 3 && row < rows) bandViolation = 1;
                }
                minGood[k] = (good & (minGood[k] < 0)) ? c : minGood[k];
                maxGood[k] = good ? c : maxGood[k];
                // next row of this lane: diag = my previous-column cell, up = my new cell
                dgM = pM[k]; dgD = pD[k]; dgI = pI[k];
                pM[k] = nM; pD[k] = nD; pI[k] = nI;
                upM = nM; upI = nI;
                started = minGood[k] >= 0;
                if (BANDED) upMaxG = maxGood[k];
                pm8 = mPrev[k];
                mPrev[k] = mb8;
            }
            lastRef = ref1;

            // last row: first strict maximum per plane, ascending column
            {
                int lm = pM[0], ld = pD[0], li = pI[0];
#pragma unroll
                for (int k = 1; k < R; k++) {
                    lm = (lastSlot == k) ? pM[k] : lm; ld = (lastSlot == k) ? pD[k] : ld; li = (lastSlot == k) ? pI[k] : li;
                }
                const bool track = ownerLaneFlag & inRange;
                const bool um = track & ((bestMc < 0) | ((lm & kScoreMask) > (bestM & kScoreMask)));
                const bool ud = track & ((bestDc < 0) | ((ld & kScoreMask) > (bestD & kScoreMask)));
                const bool ui = track & ((bestIc < 0) | ((li & kScoreMask) > (bestI & kScoreMask)));
                bestM = um ? lm : bestM; bestMc = um ? c : bestMc;
                bestD = ud ? ld : bestD; bestDc = ud ? c : bestDc;
                bestI = ui ? li : bestI; bestIc = ui ? c : bestIc;
            }
            }   // c >= 1

            if ((t & 7) == 7) {
                const long long o = (long long)(t >> 3) * R * G + gl;
#pragma unroll
                for (int k = 0; k < R; k++) { dir[o + (long long)k * G] = dacc[k]; dacc[k] = 0; }
            }
        }
        if ((steps & 7) != 7) {                                          // partial last dword
            const long long o = (long long)(steps >> 3) * R * G + gl;
#pragma unroll
            for (int k = 0; k < R; k++) dir[o + (long long)k * G] = dacc[k];
        }

        // ------------------------------------------------------------------ row extents -> iterations, result[]
        // jni/...c:441-449, :660-661.  Row r is entered iff row r-1 had a good cell (and, with a band, its window is
        // not empty); it starts at colStart(r) and ends at the first non-good column past the last good column of
        // rows r-1 and r (no band) or one past the last good column of row r-1 (band).
        int firstNoEnter = INT_MAX;
        int lastColStart = 1, lastHasGood = 0;
        long long itersAll[R];
        {
            int pMin = lane_up(minGood[R - 1], 1), pMax = lane_up(maxGood[R - 1], columns);
            if (gl == 0) { pMin = 1; pMax = columns; }
#pragma unroll
            for (int k = 0; k < R; k++) {
                const int row = r0 + k;
                itersAll[k] = 0;
                if (rowValid[k]) {
                    const int hasGood = minGood[k] >= 0;
                    const int colStart = banded ? max(pMin, row - halfband) : pMin;
                    const int colStop = banded ? min(pMax, row + 2 * halfband - 1) : pMax;
                    const bool enter = pMin >= 0 && colStart >= 0 && colStop >= colStart;
                    if (!enter && firstNoEnter == INT_MAX) firstNoEnter = row;
                    const int endc = banded ? min(columns, colStop + 1)
                                            : min(columns, max(colStop, hasGood ? maxGood[k] : -2) + 1);
                    itersAll[k] = (long long)(endc - colStart + 1);
                    if (row == rows) { lastColStart = colStart; lastHasGood = hasGood; }
                }
                pMin = minGood[k]; pMax = maxGood[k];
            }
        }
        int groupNoEnter = firstNoEnter;
        for (int d = 1; d < G; d <<= 1) groupNoEnter = min(groupNoEnter, __shfl_xor(groupNoEnter, d, 64));
        long long iters = 0;
#pragma unroll
        for (int k = 0; k < R; k++) if (rowValid[k] && r0 + k < groupNoEnter) iters += itersAll[k];
        for (int d = 1; d < G; d <<= 1) iters += __shfl_xor(iters, d, 64);
        if (!limited) iters = (long long)rows * columns;
        for (int d = 1; d < G; d <<= 1) bandViolation |= __shfl_xor(bandViolation, d, 64);

        // the lane that owns the last row knows the argmax; broadcast it through the group
        const int ownerLane = groupBase + (max(rows, 1) - 1) / R;
        int bScore, bCol, bState, bPacked;
        {
            int s = bestMc >= 0 ? (bestM & kScoreMask) : INT_MIN, cbest = bestMc, st = 0, pk = bestM;
            if (bestDc >= 0 && (bestD & kScoreMask) > s) { s = bestD & kScoreMask; cbest = bestDc; st = 1; pk = bestD; }
            if (bestIc >= 0 && (bestI & kScoreMask) > s) { s = bestI & kScoreMask; cbest = bestIc; st = 2; pk = bestI; }
            bScore = __shfl(s, ownerLane, 64);
            bCol = __shfl(cbest, ownerLane, 64);
            bState = __shfl(st, ownerLane, 64);
            bPacked = __shfl(pk, ownerLane, 64);
            lastColStart = __shfl(lastColStart, ownerLane, 64);
            lastHasGood = __shfl(lastHasGood, ownerLane, 64);
        }

        int res0 = rows, res1, res2, res3, res4 = 0;
        bool fillNull = false;           // result[4]==1
        if (!limited) {
            res1 = bCol; res2 = bState; res3 = bScore >> kScoreOffset;
        } else if (groupNoEnter <= rows) {   // the fill stopped before the last row: it still holds BADoff everywhere
            res1 = 1; res2 = 0; res3 = kBadOff; res4 = 1; fillNull = true;
        } else if (!lastHasGood) {       // last row visited, nothing good: first subfloor cell in scan order
            res1 = max(1, lastColStart - 1); res2 = 0; res3 = subfloor; res4 = 1; fillNull = true;
        } else if (bScore < minScoreOff) {
            res1 = bCol; res2 = bState; res3 = bScore; res4 = 1; fillNull = true;
        } else {
            res1 = bCol; res2 = bState; res3 = bScore >> kScoreOffset;
        }

        // ------------------------------------------------------------------ score2 + traceback2 on the records
        const bool wantScore = run && !fillNull && !bandViolation && (jb.flags & BBMSA_DO_SCORE);
        const bool wantTrace = run && !fillNull && !bandViolation && (jb.flags & BBMSA_DO_TRACEBACK) && p.match != nullptr;
        int sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int scoreLen = 0, matchLen = 0;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // direction records are in L2 before anyone reads them
        __builtin_amdgcn_wave_barrier();

        if (__any(wantScore || wantTrace)) {
            const bool walk = wantScore || wantTrace;
            int row = walk ? rows : 0, col = walk ? res1 : 0, state = res2;
            int n = 0, gapSyms = 0, stateTime = 0;
            while (__any(row > 0 && col > 0)) {
                const bool go = row > 0 && col > 0;
                if (go && state == 0) {
                    // diagonal run: lane gl looks at cell (row-gl, col-gl)
                    const int rr = row - gl, cq = col - gl;
                    const bool inside = rr >= 1 && cq >= 1;
                    unsigned nibv = 0;
                    if (inside) {
                        const int ol = (rr - 1) / R, ok = (rr - 1) - ol * R, ot = cq + ol;
                        const unsigned dw = load_coherent(dir + ((long long)(ot >> 3) * R + ok) * G + ol);
                        nibv = (dw >> ((ot & 7) * 4)) & 15u;
                    }
                    const bool brk = !inside || (nibv & 3u) != 0u;
                    unsigned long long bal = __ballot(brk);
                    unsigned long long mine = (G == 64) ? bal : ((bal >> groupBase) & ((1ull << G) - 1ull));
                    const int fb = mine ? __builtin_ctzll(mine) : G;            // first break in my group
                    const int fbInside = __shfl((int)inside, groupBase + min(fb, G - 1), 64);
                    const int fbPrev = __shfl((int)(nibv & 3u), groupBase + min(fb, G - 1), 64);
                    const int consumed = (fb < G && fbInside) ? fb + 1 : fb;
                    if (wantTrace && gl < consumed) {
                        const int cb = rd[rr - 1], rb = colRef[cq];
                        tmp[n + gl] = (cb == rb) ? 'm' : ((!fully_defined(cb) || !fully_defined(rb)) ? 'N' : 'S');
                    }
                    stateTime += fb;
                    if (fb < G && fbInside) { stateTime = 0; state = fbPrev; }
                    row -= consumed; col -= consumed; n += consumed;
                } else if (go) {
                    const int ol = (row - 1) / R, ok = (row - 1) - ol * R, ot = col + ol;
                    const unsigned dw = load_coherent(dir + ((long long)(ot >> 3) * R + ok) * G + ol);
                    const unsigned nibv = (dw >> ((ot & 7) * 4)) & 15u;
                    int prev;
                    if (state == 1) {
                        prev = (nibv & 4u) ? 1 : 0;
                        const int rb = colRef[col];
                        if (wantTrace && gl == 0) tmp[n] = (rb == '-') ? '-' : 'D';
                        if (rb == '-') gapSyms++;
                        col--;
                    } else {
                        prev = (nibv & 8u) ? 2 : 0;
                        if (wantTrace && gl == 0) tmp[n] = (col >= columns) ? 'Y' : 'I';
                        row--;
                    }
                    n++;
                    if (prev == state) stateTime++; else stateTime = 0;
                    state = prev;
                }
            }
            if (walk) {
                // score2 tail: MultiStateAligner11tsJNI.java:625-657
                int colS = col;
                if (row > colS) colS -= row;
                const int bestRefStart = a + colS;
                const int bestRefStop = a + res1 - 1;
                int padLeft = 0, padRight = 0;
                if (bestRefStart < a) padLeft = max(0, a - bestRefStart);
                else if (bestRefStart == a && state == 2) padLeft = stateTime;
                // score(..., gapped=true) walks with refEndLoc = translateToGappedCoordinate(b), which the library
                // parked in ref_len (MultiStateAligner11tsJNI.java:507-516)
                const int bW = (jb.flags & BBMSA_INTERNAL_GAPPED) ? jb.ref_len : b;
                if (bestRefStop > bW) padRight = max(0, bestRefStop - bW);
                else if (bestRefStop == bW && res2 == 2) padRight = bPacked & kTimeMask;
                if (wantScore) {
                    sc[0] = bScore >> kScoreOffset; sc[1] = bestRefStart; sc[2] = bestRefStop;
                    sc[3] = rows; sc[4] = res1; sc[5] = res2; sc[6] = padLeft; sc[7] = padRight;
                    scoreLen = (padLeft > 0 || padRight > 0) ? 8 : 6;
                    if (scoreLen == 6) { sc[6] = 0; sc[7] = 0; }
                }
                if (wantTrace) {
                    // traceback2 tail (:460-471): leftover read bases become 'X'
                    int xs = (col != row) ? row : 0;
                    for (int i = gl; i < xs; i += G) tmp[n + i] = 'X';
                    n += xs;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint8_t *out = p.match + j * (long long)p.match_stride;
                    const bool keepGaps = (jb.flags & BBMSA_TRACE_KEEP_GAPS) != 0;      // leave each '-' in the string (the caller expands)
                    const int total = keepGaps ? n : n + gapSyms * (kGapLen - 1);
                    if (total > p.match_stride) matchLen = -1;
                    else if (gapSyms == 0 || keepGaps) {
                        for (int i = gl; i < n; i += G) out[i] = tmp[n - 1 - i];
                        matchLen = n;
                    } else {
                        if (gl == 0) {                       // rare: expand each '-' to 128 'D' (:481-493)
                            int o = 0;
                            for (int i = n - 1; i >= 0; i--) {
                                const uint8_t ch = tmp[i];
                                if (ch != '-') out[o++] = ch;
                                else for (int q = 0; q < kGapLen; q++) out[o++] = 'D';
                            }
                        }
                        matchLen = total;
                    }
                }
            }
        }

        if (run && gl == 0 && bandViolation) {
            const unsigned k = atomicAdd(p.slow_count, 1u);
            p.slow_list[k] = (int)j;
        }
        if (run && gl == 0 && !bandViolation) {
            bbmsa_result r;
            r.result[0] = res0; r.result[1] = res1; r.result[2] = res2; r.result[3] = res3; r.result[4] = res4;
            r.status = (fillNull && mode == BBMSA_FILL_LIMITED) ? BBMSA_ST_NULL : BBMSA_ST_OK;
            r.iterations = iters;
#pragma unroll
            for (int i = 0; i < 8; i++) r.score[i] = sc[i];
            r.score_len = scoreLen; r.match_len = matchLen;
            r.fill_kind = limited ? 0 : 1; r.columns = columns;
            p.results[j] = r;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// explicit instantiations used by the host side (msa_host.hip)
#define BBMSA_INST(R)                                                               \
    template __global__ void msa_fill_fast_kernel<R, false>(const FillParams);      \
    template __global__ void msa_fill_fast_kernel<R, true>(const FillParams);
BBMSA_INST(1) BBMSA_INST(2) BBMSA_INST(3) BBMSA_INST(4) BBMSA_INST(5)
BBMSA_INST(6) BBMSA_INST(7) BBMSA_INST(8) BBMSA_INST(9) BBMSA_INST(10)

#define BBMSA_CASE_MAT(R) case R: return banded ? (const void *)msa_fill_fast_kernel<R, true, true> : (const void *)msa_fill_fast_kernel<R, false, true>;
const void *fast_kernel_mat_for(int R, bool banded) {
    switch (R) {
        BBMSA_CASE_MAT(1) BBMSA_CASE_MAT(2) BBMSA_CASE_MAT(3) BBMSA_CASE_MAT(4) BBMSA_CASE_MAT(5)
        BBMSA_CASE_MAT(6) BBMSA_CASE_MAT(7) BBMSA_CASE_MAT(8) BBMSA_CASE_MAT(9) BBMSA_CASE_MAT(10)
    }
    return nullptr;
}
#define BBMSA_CASE(R) case R: return banded ? (const void *)msa_fill_fast_kernel<R, true> : (const void *)msa_fill_fast_kernel<R, false>;
const void *fast_kernel_for(int R, bool banded) {
    switch (R) {
        BBMSA_CASE(1) BBMSA_CASE(2) BBMSA_CASE(3) BBMSA_CASE(4) BBMSA_CASE(5)
        BBMSA_CASE(6) BBMSA_CASE(7) BBMSA_CASE(8) BBMSA_CASE(9) BBMSA_CASE(10)
    }
    return nullptr;
}

}  // namespace bbmsa
