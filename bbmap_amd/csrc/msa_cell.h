// ONE cell of fillLimitedX / fillUnlimited -- the three planes, their prune tests and the 4-bit traceback record -- shared by the
// wavefront kernel (msa_fill_fast.hip, MultiStateAligner11ts) and the strip-tiled kernel (msa_fill_strip.hip,
// MultiStateAligner9PacBio).  Reference: jni/MultiStateAligner11tsJNI.c:458-658 (limited) / :168-300 (unlimited);
// current/align2/MultiStateAligner9PacBio.java:377-560.  The two kernels differ in their schedule (who owns which rows, where the row
// above comes from) and in HOW the streak- and length-dependent penalties are looked up, not in what a cell is:
//   * S   = the scheme (points, bit layout of a packed cell: msa_common.h Scheme11ts / Scheme9PacBio);
//   * Pen = the lookup policy: LdsPen reads the per-block tables the wavefront kernel keeps in LDS (one 16-byte entry answers the
//     match plane's case distinctions, one table serves the "still needed" terms of all three planes), SpelledPen evaluates the
//     closed forms (a 6,000-row strip would need 50 KB of tables per block).
// Everything is branch-free (selects only) so that the R rows of a lane form one basic block the scheduler can interleave.
// Tricks that keep it exact:
//   * a packed cell p = score|time (score a multiple of ONE = TMASK + 1, time <= TMASK) satisfies score <= L  <=>  p < L + ONE for
//     any multiple-of-ONE bound L, so prune tests run on packed values against "limit + ONE" (suffix P) without masking;
//   * prevMatch(row, col) is match(row-1, col-1): the bit computed one step earlier for the row above (carried as 8 / 0, a bit of
//     the match plane's table index);
//   * the predecessor traceback2 would pick for a DEL / INS cell is exactly "the extension won the fill's own comparison"
//     (extension costs are never below the opening cost), so those two record bits are free.
#pragma once
#include "msa_common.h"

namespace bbmsa {

struct CellIn {
    int row, c, rows, insNeededBase;         // insNeededBase = (columns - c) + 1: insNeeded = (rows - row) - insNeededBase
    int cl1, ref1;                           // read base of the row, reference base of the column
    bool refN, gap, match, act;              // ref1 == 'N', ref1 == '-', cl1 == ref1 && !refN, "the row's window has reached this column"
    int refPen;                              // DEL_REF_N / GAP / 0 of the column
    int limitP, floorP, subfloor;            // max(vertLimit, horizLimit) + ONE, floor + ONE, what a pruned cell holds
    int dgM, dgD, dgI, lM, lD, upM, upI;     // (row-1, c-1), (row, c-1), (row-1, c): packed
    int delForce, insForce;                  // INT_MAX where the barrier rows / columns forbid the plane, else INT_MIN
    int pm8;                                 // prevMatch as 8 / 0
};
struct CellOut {
    int nM, nD, nI;                          // the new packed cells
    int timeM, timeD, timeI;                 // unclamped times (the matrix-materialising mode stores subfloor | time for a computed bad cell)
    bool pruneM, pruneD, pruneI, goodM, goodD, goodI;
    unsigned nib;                            // the traceback record
    int mb8;                                 // match as 8 / 0: the next row's / next column's pm8
};

// (-DBBMSA_LDS_PINS pins every looked-up value to a VGPR at the place it was loaded, asm volatile("" : "+v"(x)).  Rounds 2-3 needed
// that: the compiler hoisted the table loads of all the rows of a lane to the top of the step and spilled what they displaced.
// With the cell as one inlined function it no longer does, and each pin is an s_waitcnt lgkmcnt(0) right behind its ds_read -- six
// exposed LDS round trips per cell: without them 19 instead of 26 spilled VGPRs and the step 224.5 -> 221.5 ms.)
#ifdef BBMSA_LDS_PINS
#define BBMSA_PIN(x) asm volatile("" : "+v"(x))
#else
#define BBMSA_PIN(x) do { } while (0)
#endif

// ---- lookup policies
struct MEntry { int addA, bonus, t3sub; };   // points of staying in the match plane, of entering it from D / I, what its prune test subtracts
struct NeedPen { bool needDel, needIns; int need; const int *X; int pen0, penDel, penIns; };   // pen0: the match plane's; penDel / penIns: SpelledPen only

// the wavefront kernel's LDS tables (filled by the kernel's prologue; see msa_fill_fast.hip)
struct LdsPen {
    const int *delC, *insC, *delExt, *insExt;
    const int4 *mTab;                        // index = min(streak, 5) | match << 3 | prevMatch << 4
    __device__ __forceinline__ MEntry m_entry(int streakM, int mb8, int pm8, bool, bool) const {
        int4 mt = mTab[min(streakM, 5) | mb8 | (pm8 << 1)];
        BBMSA_PIN(mt.x);
        MEntry e; e.addA = mt.x; e.bonus = mt.y; e.t3sub = mt.z;
        return e;
    }
    // A cell needs deletions (left of the corridor) or insertions (right of it), never both: jobs whose window is more than two
    // columns narrower than the read, the only shape where both can hold, are handed on at job setup.  So one table serves the cell:
    // X = delC or insC, pen0 = X[needed], and the "still needed after this streak" term X[time + needed] - X[time] of the plane that
    // continues such a run.
    __device__ __forceinline__ NeedPen need(int delNeeded, int insNeeded) const {
        NeedPen n; n.needDel = delNeeded > 0; n.needIns = insNeeded > 0; n.need = delNeeded + insNeeded;
        n.X = n.needDel ? delC : insC;
        int p0 = n.X[n.need];
        BBMSA_PIN(p0);
        n.pen0 = p0; n.penDel = 0; n.penIns = 0;
        return n;
    }
    __device__ __forceinline__ int del_ext(int streakD) const { int d = delExt[min(streakD, 80 | (streakD & 3))]; BBMSA_PIN(d); return d; }
    __device__ __forceinline__ int ins_ext(int streakI) const { int i = insExt[min(streakI, 20)]; BBMSA_PIN(i); return i; }
    __device__ __forceinline__ void rest(const NeedPen &n, int, int, int timeD, int timeI, int &penD, int &penI) const {
        const int timeX = n.needDel ? timeD : timeI;
        int x2 = n.X[timeX + n.need] - n.X[timeX];                     // 0 when nothing is still needed
        BBMSA_PIN(x2);
        penD = n.needIns ? n.pen0 : x2;
        penI = n.needDel ? n.pen0 : x2;
    }
};

// closed forms of the scheme (the strip kernel: reads of 6,000 bases against 7,600 columns)
template <class S> struct SpelledPen {
    __device__ __forceinline__ MEntry m_entry(int streakM, int, int, bool prevMatch, bool match) const {
        int subx = S::SUB3;
        subx = streakM < 5 ? S::SUB2 : subx; subx = streakM == 0 ? S::SUB : subx;
        MEntry e;
        e.addA = match ? (prevMatch ? S::MATCH2 : S::MATCH) : (prevMatch ? (streakM <= 1 ? S::SUBR : S::SUB) : subx);
        e.bonus = match ? S::MATCH : S::SUB;
        e.t3sub = match ? S::MATCH2 : S::SUB3;
        return e;
    }
    __device__ __forceinline__ NeedPen need(int delNeeded, int insNeeded) const {
        // (a window narrower than the read can need both at once; the planes then take them in the reference's order: :539-541, :601-607, :645-651)
        NeedPen n; n.needDel = delNeeded > 0; n.needIns = insNeeded > 0; n.need = delNeeded + insNeeded; n.X = nullptr;
        n.penDel = S::del_off(delNeeded); n.penIns = S::ins_cum_off(insNeeded);
        n.pen0 = n.needDel ? n.penDel : (n.needIns ? n.penIns : 0);
        return n;
    }
    __device__ __forceinline__ int del_ext(int streak) const {                 // (select chains: no branches in the cell)
        int c = (streak & 3) == 0 ? S::DEL5 : 0;
        c = streak < 80 ? S::DEL4 : c; c = streak < 20 ? S::DEL3 : c; c = streak < 5 ? S::DEL2 : c; c = streak == 0 ? S::DEL : c;
        return c;
    }
    __device__ __forceinline__ int ins_ext(int streak) const {
        int c = S::INS4;
        c = streak < 20 ? S::INS3 : c; c = streak < 5 ? S::INS2 : c; c = streak == 0 ? S::INS : c;
        return c;
    }
    __device__ __forceinline__ void rest(const NeedPen &n, int delNeeded, int insNeeded, int timeD, int timeI, int &penD, int &penI) const {
        penD = n.needIns ? n.penIns : (n.needDel ? S::del_off(timeD + delNeeded) - S::del_off(timeD) : 0);
        penI = n.needDel ? n.penDel : (n.needIns ? S::ins_cum_off(timeI + insNeeded) - S::ins_cum_off(timeI) : 0);
    }
};

template <class S> __device__ __forceinline__ int clamp_cell_time(int t) { return t > S::MAXT ? S::MAXT - 3 : t; }

// CLAMP_ALL: clamp the match and insertion planes' times too (their streaks are bounded by the rows: only reads longer than MAXT need it)
template <class S, bool CLAMP_ALL, class Pen>
__device__ __forceinline__ CellOut cell_update(const Pen &pen, const CellIn &in) {
    constexpr int ONE = S::TMASK + 1;
    CellOut o;
    const int limit = in.limitP - ONE;
    const int delNeeded = max(0, in.row - in.c - 1);
    const int insNeeded = max(0, (in.rows - in.row) - in.insNeededBase);
    const NeedPen np = pen.need(delNeeded, insNeeded);

    // ---- match / substitution plane (diagonal)
    const int streakM = in.dgM & S::TMASK;
    const int sdm = in.dgM & S::SMASK;
    const int mDI = max(in.dgD, in.dgI) & S::SMASK;
    const bool prevMatch = in.pm8 != 0;
    o.mb8 = in.match ? 8 : 0;
    const MEntry me = pen.m_entry(streakM, o.mb8, in.pm8, prevMatch, in.match);
    const int t3 = max(in.floorP, in.limitP - me.t3sub);
    o.pruneM = !in.act | in.gap | (max(in.dgM, max(in.dgD, in.dgI)) < t3);      // (bitwise: no short-circuit branches)
    const int addA = (in.refN | (in.cl1 == 'N')) ? 0 : me.addA;                  // (a match has neither base N)
    const int sa = sdm + addA;
    const int sbc = mDI + me.bonus;
    const bool aWinsM = sa >= sbc;
    const int scoreM = max(sa, sbc);
    o.timeM = (aWinsM & (in.match == prevMatch)) ? streakM + 1 : 1;
    o.goodM = !o.pruneM & (scoreM + np.pen0 >= limit);                           // the offsets are negative: score >= limit - offset
    o.nM = o.goodM ? (scoreM | (CLAMP_ALL ? clamp_cell_time<S>(o.timeM) : o.timeM)) : in.subfloor;

    // ---- deletion plane (same row, previous column)
    const int streakD = in.lD & S::TMASK;
    const int slm = in.lM & S::SMASK, sld = in.lD & S::SMASK;
    o.pruneD = !in.act | (max(in.lM, in.lD) < max(in.limitP, in.delForce));
    const int dsa = slm + S::DEL;
    const int dsb = sld + pen.del_ext(streakD);
    const bool aWinsD = dsa >= dsb;
    const int scoreD = max(dsa, dsb) + in.refPen;
    o.timeD = aWinsD ? 1 : streakD + 1;

    // ---- insertion plane (row above, same column)
    const int streakI = in.upI & S::TMASK;
    const int sum = in.upM & S::SMASK, sui = in.upI & S::SMASK;
    o.pruneI = !in.act | in.gap | (max(in.upM, in.upI) < max(in.limitP, in.insForce));
    const int isa = sum + S::INS;
    const int isb = sui + pen.ins_ext(streakI);
    const bool aWinsI = isa >= isb;
    const int scoreI = max(isa, isb);
    o.timeI = aWinsI ? 1 : streakI + 1;

    int penD, penI;
    pen.rest(np, delNeeded, insNeeded, o.timeD, o.timeI, penD, penI);
    o.goodD = !o.pruneD & (scoreD + penD >= limit);
    o.nD = o.goodD ? (scoreD | clamp_cell_time<S>(o.timeD)) : in.subfloor;
    o.goodI = !o.pruneI & (scoreI + penI >= limit);
    o.nI = o.goodI ? (scoreI | (CLAMP_ALL ? clamp_cell_time<S>(o.timeI) : o.timeI)) : in.subfloor;

    // ---- traceback record (MultiStateAligner11tsJNI.java:389-443): what traceback2 / score2 would decide at this cell (time > 1:
    // stay in the plane; else the predecessor comparison on scores)
    const bool msStay = (o.timeM > 1) | (sdm >= mDI);
    const unsigned nibM = msStay ? 0u : (((in.dgD | S::TMASK) >= in.dgI) ? 1u : 2u);
    o.nib = nibM | (aWinsD ? 0u : 4u) | (aWinsI ? 0u : 8u);
    return o;
}

}  // namespace bbmsa
