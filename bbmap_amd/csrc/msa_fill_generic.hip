// Generic path of the MultiStateAligner11ts DP for gfx950: one alignment per thread with the
// three score planes kept in a per-thread HBM scratch matrix, exactly the data layout the
// reference's native code works on (state-major planes, jni/MultiStateAligner11tsJNI.c:124-127).
//
// It exists for the jobs the wavefront kernel (msa_fill_fast.hip) cannot take: windows wider than
// its LDS column buffer, and banded fills whose rows have a hole in their "good" columns.  It is
// slow (a serial chain per thread, every cell in HBM/L2) but handles every shape and every band
// setting, and it is the device-side statement of the full semantics:
//   fillUnlimited  jni/MultiStateAligner11tsJNI.c:100-314
//   fillLimitedX   jni/MultiStateAligner11tsJNI.c:361-704 (band :392-393, :441-442)
//   traceback2 / score2  current/align2/MultiStateAligner11tsJNI.java:376-495, :537-658
#include "msa_common.h"

namespace bbmsa {

namespace {

struct Planes {
    int *M, *D, *I;
    int W;     // row stride (columns + 2)
};

template <class S> __device__ inline int ctime(int t) { return t > S::MAXT ? S::MAXT - 3 : t; }

template <class S> __device__ inline int del_step(int streak) {
    if (streak == 0) return S::DEL;
    if (streak < 5) return S::DEL2;
    if (streak < 20) return S::DEL3;
    if (streak < 80) return S::DEL4;
    return (streak & 3) == 0 ? S::DEL5 : 0;
}
template <class S> __device__ inline int ins_step(int streak) {   // POINTSoff_INS_ARRAY[streak+1] / the tiered form of 9PacBio
    if (streak == 0) return S::INS;
    if (streak < 5) return S::INS2;
    if (streak < 20) return S::INS3;
    return S::INS4;
}
template <class S> __device__ inline int sub_step(int streak) {   // POINTSoff_SUB_ARRAY[streak+1] / the tiered form of 9PacBio
    if (streak == 0) return S::SUB;
    if (streak < 5) return S::SUB2;
    return S::SUB3;
}

// one DP cell; `limited` adds the prune tests.  Returns true if any plane is "good".
template <class S> __device__ inline bool dp_cell(const Planes &pl, bool limited, int row, int col, int rows, int columns,
                               int call0, int call1, int ref0, int ref1,
                               int vlimit, int hlimit, int floorv, int subfloor) {
    const int up = (row - 1) * pl.W, cur = row * pl.W;
    const bool gap = ref1 == '-';
    const bool match = (call1 == ref1) && ref1 != 'N';
    const bool prevMatch = (call0 == ref0) && ref0 != 'N';
    const int limit = limited ? max(vlimit, hlimit) : kNegInf;
    const int limit3 = limited ? max(floorv, match ? limit - S::MATCH2 : limit - S::SUB3) : kNegInf;
    const int delNeeded = max(0, row - col - 1);
    const int insNeeded = max(0, (rows - row) - (columns - col) - 1);
    const int delPen = S::del_off(delNeeded);
    const int insPen = S::ins_cum_off(insNeeded);
    bool anyGood = false;

    const int dmP = pl.M[up + col - 1];
    const int dm = dmP & S::SMASK, dd = pl.D[up + col - 1] & S::SMASK, di = pl.I[up + col - 1] & S::SMASK;
    const int lm = pl.M[cur + col - 1] & S::SMASK;
    const int ldP = pl.D[cur + col - 1];
    const int ld = ldP & S::SMASK;
    const int um = pl.M[up + col] & S::SMASK;
    const int uiP = pl.I[up + col];
    const int ui = uiP & S::SMASK;

    if (gap || (limited && dm <= limit3 && dd <= limit3 && di <= limit3)) {
        pl.M[cur + col] = subfloor;
    } else {
        const int streak = dmP & S::TMASK;
        int a, bonus, tA;
        if (match) {
            a = dm + (prevMatch ? S::MATCH2 : S::MATCH); bonus = S::MATCH; tA = prevMatch ? streak + 1 : 1;
        } else {
            if (ref1 != 'N' && call1 != 'N') a = dm + (prevMatch ? (streak <= 1 ? S::SUBR : S::SUB) : sub_step<S>(streak));
            else a = dm;
            bonus = S::SUB; tA = prevMatch ? 1 : streak + 1;
        }
        const int bb = dd + bonus, cc = di + bonus;
        int score, time;
        if (a >= bb && a >= cc) { score = a; time = tA; }
        else if (bb >= cc) { score = bb; time = 1; }
        else { score = cc; time = 1; }
        if (limited) {
            const int limit2 = delNeeded > 0 ? limit - delPen : (insNeeded > 0 ? limit - insPen : limit);
            if (score >= limit2) anyGood = true; else score = subfloor;
        }
        pl.M[cur + col] = score | ctime<S>(time);
    }

    if ((limited && lm <= limit && ld <= limit) || row < S::BAR_D1 || row > rows - S::BAR_D1) {
        pl.D[cur + col] = subfloor;
    } else {
        const int streak = ldP & S::TMASK;
        int a = lm + S::DEL, bsc = ld + del_step<S>(streak);
        if (ref1 == 'N') { a += S::DEL_REF_N; bsc += S::DEL_REF_N; }
        else if (gap) { a += S::GAP; bsc += S::GAP; }
        int score, time;
        if (a >= bsc) { score = a; time = 1; } else { score = bsc; time = streak + 1; }
        if (limited) {
            int limit2 = limit;
            if (insNeeded > 0) limit2 = limit - insPen;
            else if (delNeeded > 0) limit2 = limit - S::del_off(time + delNeeded) + S::del_off(time);
            if (score >= limit2) anyGood = true; else score = subfloor;
        }
        pl.D[cur + col] = score | ctime<S>(time);
    }

    if (gap || (limited && um <= limit && ui <= limit) || (row < S::BAR_I1 && col > 1) || (row > rows - S::BAR_I1 && col < columns - 1)) {
        pl.I[cur + col] = subfloor;
    } else {
        const int streak = uiP & S::TMASK;
        const int a = um + S::INS, bsc = ui + ins_step<S>(streak);
        int score, time;
        if (a >= bsc) { score = a; time = 1; } else { score = bsc; time = streak + 1; }
        if (limited) {
            int limit2 = limit;
            if (delNeeded > 0) limit2 = limit - delPen;
            else if (insNeeded > 0) limit2 = limit - S::ins_cum_off(time + insNeeded) + S::ins_cum_off(time);
            if (score >= limit2) anyGood = true; else score = subfloor;
        }
        pl.I[cur + col] = score | ctime<S>(time);
    }
    return anyGood;
}

__device__ inline int plane_at(const Planes &pl, int state, int row, int col) {
    const int *q = state == 0 ? pl.M : (state == 1 ? pl.D : pl.I);
    return q[row * pl.W + col];
}
// predecessor rule shared by traceback2 and score2 (MultiStateAligner11tsJNI.java:389-443)
template <class S> __device__ inline int walk_prev(const Planes &pl, int state, int row, int col) {
    const int time = plane_at(pl, state, row, col) & S::TMASK;
    if (time > 1) return state;
    if (state == 0) {
        const int a = plane_at(pl, 0, row - 1, col - 1) & S::SMASK;
        const int b = plane_at(pl, 1, row - 1, col - 1) & S::SMASK;
        const int c = plane_at(pl, 2, row - 1, col - 1) & S::SMASK;
        if (a >= b && a >= c) return 0;
        return b >= c ? 1 : 2;
    }
    if (state == 1) {
        const int a = plane_at(pl, 0, row, col - 1) & S::SMASK, b = plane_at(pl, 1, row, col - 1) & S::SMASK;
        return a >= b ? 0 : 1;
    }
    const int a = plane_at(pl, 0, row - 1, col) & S::SMASK, b = plane_at(pl, 2, row - 1, col) & S::SMASK;
    return a >= b ? 0 : 2;
}

}  // namespace

template <class S> __global__ __launch_bounds__(64) void msa_fill_generic_kernel(const GenericParams p) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long planeInts = (long long)(p.maxRows + 1) * (p.maxColumns + 2);
    int *base = p.matrix + tid * 3 * planeInts;
    int *vertLimit = p.limits + tid * (long long)(p.maxRows + p.maxColumns + 4);
    int *horizLimit = vertLimit + p.maxRows + 2;
    const long long total = p.list ? (long long)*p.list_count : job_count(p.njobs, p.njobs_dev);

    for (;;) {
        const long long q = (long long)atomicAdd(p.queue, 1u);
        if (q >= total) break;
        const long long j = p.list ? (long long)p.list[q] : q;
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
        bbmsa_result r;
        for (int i = 0; i < 5; i++) r.result[i] = 0;
        for (int i = 0; i < 8; i++) r.score[i] = 0;
        r.status = BBMSA_ST_OK; r.iterations = 0; r.score_len = 0; r.match_len = 0; r.fill_kind = 0; r.columns = columns;
        if (rows < 1 || columns < 1 || rows > p.maxRows || columns > p.maxColumns) {
            r.status = BBMSA_ST_BAD_SHAPE;
            p.results[j] = r;
            continue;
        }
        const uint8_t *rd = p.reads + jb.read_off;
        const uint8_t *rf = p.refs + jb.ref_off;

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

        Planes pl;
        pl.W = columns + 2;
        pl.M = base; pl.D = base + planeInts; pl.I = base + 2 * planeInts;
        // row 0 is zero, column 0 is the cumulative insertion cost (MultiStateAligner11tsJNI.java:101-112);
        // everything else is whatever an earlier job left behind, as in the reference.
        for (int c = 0; c <= columns + 1; c++) { pl.M[c] = 0; pl.D[c] = 0; pl.I[c] = 0; }
        for (int i = 1; i <= rows; i++) {
            const int v = S::col0(i);
            pl.M[i * pl.W] = v; pl.D[i * pl.W] = v; pl.I[i * pl.W] = v;
        }

        const int maxGain = (rows - 1) * S::MATCH2 + S::MATCH;
        const int minScoreOff = minScore * (1 << S::OFF);
        long long iters = 0;
        int bestScore, bestCol, bestState;
        bool fillNull = false;

        if (!limited) {
            const int subfloor = 0 - 2 * maxGain;
            for (int row = 1; row <= rows; row++) {
                const int call0 = row < 2 ? '?' : rd[row - 2], call1 = rd[row - 1];
                for (int col = 1; col <= columns; col++) {
                    const int ref0 = col < 2 ? '!' : rf[a + col - 2], ref1 = rf[a + col - 1];
                    dp_cell<S>(pl, false, row, col, rows, columns, call0, call1, ref0, ref1, 0, 0, kNegInf, subfloor);
                }
            }
            iters = (long long)rows * columns;
        } else {
            const int floorv = minScoreOff - maxGain;
            const int subfloor = floorv - 5 * S::MATCH2;
            for (int c = 1; c <= columns; c++) {
                pl.M[rows * pl.W + c] = S::BADOFF; pl.D[rows * pl.W + c] = S::BADOFF; pl.I[rows * pl.W + c] = S::BADOFF;
            }
            vertLimit[rows] = minScoreOff;
            bool prevDef = false;
            for (int i = rows - 1; i >= 0; i--) {
                const bool def = fully_defined(rd[i]);
                vertLimit[i] = max(vertLimit[i + 1] - (def ? (prevDef ? S::MATCH2 : S::MATCH) : 0), floorv);
                prevDef = def;
            }
            horizLimit[columns] = minScoreOff;
            prevDef = false;
            for (int i = columns - 1; i >= 0; i--) {
                const int cb = rf[a + i];
                const bool def = fully_defined(cb);
                horizLimit[i] = max(horizLimit[i + 1] - (def ? (prevDef ? S::MATCH2 : S::MATCH)
                                                             : ((prevDef && cb == '-') ? S::DEL : 0)), floorv);
                prevDef = def;
            }
            int minGoodCol = 1, maxGoodCol = columns;
            for (int row = 1; row <= rows; row++) {
                const int colStart = halfband < 1 ? minGoodCol : max(minGoodCol, row - halfband);
                const int colStop = halfband < 1 ? maxGoodCol : min(maxGoodCol, row + halfband * 2 - 1);
                minGoodCol = -1; maxGoodCol = -2;
                if (colStart < 0 || colStop < colStart) break;
                const int up = (row - 1) * pl.W, cur = row * pl.W;
                if (colStart > 1) { pl.M[cur + colStart - 1] = subfloor; pl.I[cur + colStart - 1] = subfloor; pl.D[cur + colStart - 1] = subfloor; }
                const int call0 = row < 2 ? '?' : rd[row - 2], call1 = rd[row - 1];
                const int vlimit = vertLimit[row];
                for (int col = colStart; col <= columns; col++) {
                    const int ref0 = col < 2 ? '!' : rf[a + col - 2], ref1 = rf[a + col - 1];
                    iters++;
                    if (dp_cell<S>(pl, true, row, col, rows, columns, call0, call1, ref0, ref1, vlimit, horizLimit[col], floorv, subfloor)) {
                        maxGoodCol = col; if (minGoodCol < 0) minGoodCol = col;
                    }
                    if (col >= colStop) {
                        if (col > colStop && (maxGoodCol < col || halfband > 0)) break;
                        if (row > 1) { pl.M[up + col + 1] = subfloor; pl.I[up + col + 1] = subfloor; pl.D[up + col + 1] = subfloor; }
                    }
                }
            }
        }
        // first strict maximum over the last row, state-major (jni/...c:672-686)
        bestScore = INT_MIN; bestCol = -1; bestState = -1;
        for (int s = 0; s < 3; s++) {
            for (int c = 1; c <= columns; c++) {
                const int x = plane_at(pl, s, rows, c) & S::SMASK;
                if (x > bestScore) { bestScore = x; bestCol = c; bestState = s; }
            }
        }
        r.result[0] = rows; r.result[1] = bestCol; r.result[2] = bestState;
        if (limited && bestScore < minScoreOff) { r.result[3] = bestScore; r.result[4] = 1; fillNull = true; }
        else { r.result[3] = bestScore >> S::OFF; r.result[4] = 0; }
        r.iterations = iters;
        r.fill_kind = limited ? 0 : 1;
        if (fillNull && mode == BBMSA_FILL_LIMITED) r.status = BBMSA_ST_NULL;

        if (!fillNull && (jb.flags & (BBMSA_DO_SCORE | BBMSA_DO_TRACEBACK))) {
            const bool wantTrace = (jb.flags & BBMSA_DO_TRACEBACK) && p.match != nullptr;
            uint8_t *out = wantTrace ? p.match + j * (long long)p.match_stride : nullptr;
            int row = rows, col = bestCol, state = bestState, n = 0, gaps = 0, stateTime = 0;
            bool overflow = false;
            while (row > 0 && col > 0) {
                const int prev = walk_prev<S>(pl, state, row, col);
                uint8_t sym;
                if (state == 0) {
                    const int cb = rd[row - 1], rb = rf[a + col - 1];
                    sym = (cb == rb) ? 'm' : ((!fully_defined(cb) || !fully_defined(rb)) ? 'N' : 'S');
                    row--; col--;
                } else if (state == 1) {
                    const int rb = rf[a + col - 1];
                    if (rb == '-') { sym = '-'; gaps++; } else sym = 'D';
                    col--;
                } else {
                    sym = (col >= columns) ? 'Y' : 'I';
                    row--;
                }
                if (wantTrace) { if (n < p.match_stride) out[n] = sym; else overflow = true; }
                n++;
                if (state == prev) stateTime++; else stateTime = 0;
                state = prev;
            }
            if (jb.flags & BBMSA_DO_SCORE) {
                int colS = col;
                if (row > colS) colS -= row;
                const int bestRefStart = a + colS, bestRefStop = a + bestCol - 1;
                int padLeft = 0, padRight = 0;
                if (bestRefStart < a) padLeft = max(0, a - bestRefStart);
                else if (bestRefStart == a && state == 2) padLeft = stateTime;
                const int bW = (jb.flags & BBMSA_INTERNAL_GAPPED) ? jb.ref_len : b;      // see msa_fill_fast.hip
                if (bestRefStop > bW) padRight = max(0, bestRefStop - bW);
                else if (bestRefStop == bW && bestState == 2) padRight = plane_at(pl, bestState, rows, bestCol) & S::TMASK;
                r.score[0] = bestScore >> S::OFF; r.score[1] = bestRefStart; r.score[2] = bestRefStop;
                r.score[3] = rows; r.score[4] = bestCol; r.score[5] = bestState;
                if (padLeft > 0 || padRight > 0) { r.score[6] = padLeft; r.score[7] = padRight; r.score_len = 8; }
                else r.score_len = 6;
            }
            if (wantTrace) {
                if (col != row) { while (row > 0) { if (n < p.match_stride) out[n] = 'X'; else overflow = true; n++; row--; col--; } }
                const bool keepGaps = (jb.flags & BBMSA_TRACE_KEEP_GAPS) != 0;      // leave each '-' in the string (the caller expands)
                    const int totalLen = keepGaps ? n : n + gaps * (kGapLen - 1);
                if (overflow || totalLen > p.match_stride) r.match_len = -1;
                else {
                    for (int i = 0, k = n - 1; i < k; i++, k--) { const uint8_t t0 = out[i]; out[i] = out[k]; out[k] = t0; }
                    if (gaps > 0 && !keepGaps) {           // expand '-' to 128 'D' from the back so nothing is overwritten early
                        int w = totalLen - 1;
                        for (int i = n - 1; i >= 0; i--) {
                            const uint8_t ch = out[i];
                            if (ch != '-') out[w--] = ch;
                            else for (int g = 0; g < kGapLen; g++) out[w--] = 'D';
                        }
                    }
                    r.match_len = totalLen;
                }
            }
        }
        p.results[j] = r;
    }
}

template __global__ void msa_fill_generic_kernel<Scheme11ts>(const GenericParams p);
template __global__ void msa_fill_generic_kernel<Scheme9PacBio>(const GenericParams p);

}  // namespace bbmsa
