// The FINAL ALIGNMENT STAGE of the device mapper: everything BBMapThread.processRead / processReadPair do after the rescue stage,
// i.e. the calls that produce the coordinates, the score and the match string (CIGAR) BBMap prints.  Included by mapper.hip
// inside namespace bbmapper (it uses that file's SiteScore / GapTools / list helpers, Dev and emit machinery).
//
//   processRead tail        current/align2/BBMapThread.java:492-732
//   processReadPair tail    current/align2/BBMapThread.java:1116-1356
//   genMatchString          current/align2/AbstractMapThread.java:860-965
//   genMatchStringForSite   current/align2/AbstractMapThread.java:968-1068
//   realign_new             current/align2/TranslateColorspaceRead.java:229-653  (<= 3 fillLimited + 1 fillUnlimited, score, traceback)
//   SiteScore               current/stream/SiteScore.java:430-491, :493-681 (clipTipIndels ...), :686-838 fixXY, :916-931 fixLimitsXY
//   MSA                     current/align2/MSA.java:216-484 toLocalAlignment, :488-558 score(match)
//   Read                    current/stream/Read.java:1172-1286, :1419-1469, :2051-2056, :2494-2512
//   AbstractMapThread       :1328-1349 removeDuplicateBestSites, :1820-1910 applyClearzone3, :1919-2095 pairSiteScoresFinal,
//                           :2097-2164 canPair, :2499-2609 calcTipScorePenalty / applyScorePenalty
//   Tools                   current/align2/Tools.java:913-930 countTopScores, :986-1003 removeLowQualitySitesUnpaired
// Configuration: bbmap.sh defaults (ambiguous=best, KILL_BAD_PAIRS / LOCAL_ALIGN / PRINT_SECONDARY_ALIGNMENTS / STRICT_MAX_INDEL off,
// no identity / edit filters).
//
// Shape on the device.  genMatchString is a per-read SEQUENCE of DP calls whose windows and minimum scores follow from the results
// before them (realign_new: up to four fills, called up to twice per site with a tail recursion of its own), so -- as in scoreSlow --
// every read carries a small resumable state machine (FinalRead.pc) and the DP kernels run in ROUNDS: a round advances every active
// read until it needs a fill, the fills of all reads run through the two DP contexts, the next round consumes them.  Nearly every
// imperfect read needs exactly one fill (one round); the later rounds are a handful of reads.  Match strings live in a bump-allocated
// byte pool in HBM (a site refers to its string by pool offset); a step allocates BEFORE it changes anything, so that a full pool
// (or a full fill log) leaves the read where it was and the round is simply repeated after the host has grown the buffer.
// One thread per read (per pair in the two policy kernels): the work per read is a few hundred bytes of string walking.

// ---------------------------------------------------------------------------------------------- per-read state
struct FinalRead {
    // stream.Read's mapping fields
    int mapped, paired, ambiguous, perfect, rescued;
    int chrom, strand, start, stop, mapScore;
    int match;                  // pool reference of Read.match (0 = null), length in matchLen
    int matchLen;
    // genMatchString's state
    int pc;                     // where to resume (PC_*), PC_DONE when the read has finished
    int i;                      // loop index over the sites
    int best, scoreChanged, sorting, topObj_, pairedLost;
    int oldSlow, oldScoreS;     // the site's scores before its match string was made
    // genMatchStringForSite
    int oldScoreG, gstep;
    // realign_new
    int recur, padding, forbidIndels, fixXY, minValid;
    int scoreNoIndel, minLoc, maxLoc, old0, epl, epr, fillKind, minscore, pending, haveMax, cols3;
    int seq;                    // fills issued for this read so far (continues scoreSlow's / rescue's numbering)
    int needLocal;              // the end kernel: toLocalAlignment is due (second pass, with pool space reserved)
    int reservedI;
};

enum { PC_DONE = 0, PC_SITE_LOOP, PC_GEN_START, PC_REALIGN_START, PC_EMIT_FILL, PC_FILL_BACK, PC_REALIGN_POST, PC_GEN_AFTER_REALIGN,
       PC_GEN_CLIP, PC_SITE_DONE, PC_AFTER_LOOP, PC_SORT_LOOP, PC_FINISH };

#define SITE_MATCH(ss) ((ss).reserved[0])                  // pool reference (offset / 4 + 1), 0 = match == null
#define SITE_MLEN(ss) ((ss).reserved[1] & 0x7fffffff)      // its length
// genMatchString compares SiteScore REFERENCES (`r.topSite()!=top`, :939): the top site is marked before the list is merged and sorted
// (a merge keeps the first of two equal sites and drops the second, marks included), and looked for afterwards
#define SITE_TOPMARK 0x80000000u
__device__ inline void site_set_match(Site &ss, int ref, int len) { ss.reserved[0] = ref; ss.reserved[1] = (int)(((unsigned)ss.reserved[1] & SITE_TOPMARK) | ((unsigned)len & 0x7fffffffu)); }

__device__ inline uint8_t *pool_ptr(const Dev &D, int ref) { return D.pool + 4ll * (ref - 1); }
// Bump allocation in 4-byte units; 0 = the pool is full (nothing changed; the host grows it before the round is repeated).  One
// atomicAdd, no compare-and-swap loop (a million threads retrying on one word took seconds): a request that does not fit leaves the
// counter beyond the capacity, so every later request of the round fails too, the strings handed out so far are exactly
// [0, smallest `old` of a failed request), and the host sets the counter back to that before the next round (counters[21]).
__device__ int pool_alloc_units(const Dev &D, unsigned units) {
    const unsigned old = atomicAdd(&D.counters[20], units);
    if ((long long)old + units > D.poolUnits) { atomicMin(&D.counters[21], old); atomicAdd(&D.counters[22], 1u); return 0; }
    return (int)old + 1;
}
__device__ inline unsigned pool_units(int bytes) { return (unsigned)((bytes + 3) >> 2) + 1u; }
__device__ inline int pool_alloc(const Dev &D, int bytes) { return pool_alloc_units(D, pool_units(bytes)); }
// a round's first request of every read is made for the whole wavefront at once (final_round_kernel); the read takes it here
struct PreAlloc { int ref; unsigned units; };
__device__ inline int pool_take(const Dev &D, PreAlloc &pre, int bytes) {
    if (pre.ref && pre.units >= pool_units(bytes)) { const int ref = pre.ref; pre.ref = 0; return ref; }
    return pool_alloc(D, bytes);
}

__device__ inline uint8_t ca_get(const Dev &D, int chrom, int loc) {           // ChromosomeArray.get (current/dna/ChromosomeArray.java:232-234)
    return (loc < 0 || loc >= D.chromArrLen[chrom] - 1) ? (uint8_t)'N' : D.chromArr[chrom][loc];
}
__device__ inline bool fully_defined_b(uint8_t b) {                             // AminoAcid.isFullyDefined(byte): A C G T U, either case
    const uint8_t u = b & 0xDF;
    return b < 128 && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U');
}

// Read.calcMatchLength for long-format strings (Read.java:1419-1469): every symbol but 'I' (and unknown ones) spans a reference base
__device__ int calc_match_length(const uint8_t *m, int n) {
    int len = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t c = m[i];
        len += (c == 'm' || c == 'S' || c == 'D' || c == 'C' || c == 'X' || c == 'Y' || c == 'N' || c == 'R') ? 1 : 0;
    }
    return len;
}
__device__ inline bool match_contains_xy(const uint8_t *m, int n) {            // SiteScore.java:430-434
    if (!m || n < 1) return false;
    const uint8_t a = m[0], b = m[n - 1];
    return a == 'X' || a == 'Y' || b == 'X' || b == 'Y';
}

// MSA.calcSubScore / calcInsScore / calcDelScore(len, approximateGaps = true) (MSA.java:726-747; MultiStateAligner11tsJNI.java:1347-1405)
__device__ int f_sub_score(const Settings &S, int len) {
    int score = S.ptsSub;
    if (len > 5) { score += (len - 5) * S.ptsSub3; len = 5; }
    if (len > 1) score += (len - 1) * S.ptsSub2;
    return score;
}
__device__ int f_ins_score(int len) {                       // POINTS_INS_ARRAY_C[len]: -395, then -39 (x4), -23 (x15), -8
    if (len <= 0) return 0;
    return -395 + imin(len - 1, 4) * -39 + imin(imax(len - 5, 0), 15) * -23 + imax(len - 20, 0) * -8;
}
__device__ int f_del_score(int len) {
    if (len <= 0) return 0;
    int score = -472;
    if (len > MINGAP) { const int rem = len % GAPLEN, div = (len - GAPBUFFER2) / GAPLEN; score += div * -2; len = rem + GAPBUFFER2; }
    if (len > 80) { score += ((len - 80 + 3) / 4) * -1; len = 80; }
    if (len > 20) { score += (len - 20) * -1; len = 20; }
    if (len > 5) { score += (len - 5) * -9; len = 5; }
    if (len > 1) score += (len - 1) * -33;
    return score;
}
__device__ int msa_score_mode(const Settings &S, uint8_t mode, int current, uint8_t prevMode, int prevStreak) {
    if (mode == 'm') return S.ptsMatch + (current - 1) * S.ptsMatch2;
    if (mode == 'S') {
        int s = f_sub_score(S, current);
        if (prevMode == 'N' || prevMode == 'R') s += S.ptsSub2 - S.ptsSub;
        else if (prevMode == 'm' && prevStreak < 2) s += -20;                   // POINTS_SUBR - POINTS_SUB
        return s;
    }
    if (mode == 'D') return f_del_score(current);
    if (mode == 'I' || mode == 'X' || mode == 'Y') return f_ins_score(current);
    return 0;
}
__device__ int msa_score_match(const Settings &S, const uint8_t *match, int n) {          // MSA.score(match) :488-558
    if (!match || n < 1) return 0;
    uint8_t mode = match[0], prevMode = '0'; int current = 0, prevStreak = 0, score = 0;
    for (int mpos = 0; mpos < n; mpos++) {
        const uint8_t c = match[mpos];
        if (mode == c) current++;
        else { score += msa_score_mode(S, mode, current, prevMode, prevStreak); prevMode = mode; prevStreak = current; mode = c; current = 1; }
    }
    if (current > 0) score += msa_score_mode(S, mode, current, prevMode, prevStreak);
    return score;
}

// SiteScore.leftPaddingNeeded / rightPaddingNeeded (:448-491), literally
__device__ int left_padding_needed(const uint8_t *m, int n, int tiplen, int maxIndel) {
    if (!m || n < 1) return 0;
    int insertion = 0, xy = 0;
    for (int mloc = 0; mloc < n; mloc++) {
        const uint8_t c = m[mloc];
        if (c == 'I') insertion++;
        else if (c == 'X' || c == 'Y') xy++;
        else if (c == 'D') return insertion + xy;
        else if (mloc >= tiplen) break;
    }
    if (insertion > maxIndel || xy > 0 || m[0] == 'I') return insertion + xy;
    return 0;
}
__device__ int right_padding_needed(const uint8_t *m, int n, int tiplen, int maxIndel) {
    if (!m || n < 1) return 0;
    int insertion = 0, xy = 0;
    for (int mloc = n - 1; mloc >= 0; mloc--) {
        const uint8_t c = m[mloc];
        if (c == 'I') insertion++;
        else if (c == 'X' || c == 'Y') xy++;
        else if (c == 'D') return insertion + xy;
        else if (mloc >= tiplen) break;
    }
    if (insertion > maxIndel || xy > 0 || m[n - 1] == 'I') return insertion + xy;
    return 0;
}

// SiteScore.clipLeftTipIndel (:523-575); the shortened string is compacted in place (the reference makes a new array)
__device__ bool clip_left_tip_indel(Site &ss, uint8_t *match, int &n, int tiplen, int maxIndel) {
    if (!match || n < maxIndel) return false;
    if (match[0] == 'C' || match[0] == 'Y' || match[0] == 'X') return false;
    int neutral = 0, insertion = 0, deletion = 0;
    {
        int mloc = 0;
        for (; mloc < n; mloc++) {
            const uint8_t c = match[mloc];
            if (c == 'I') insertion++;
            else if (c == 'D') deletion++;
            else { neutral++; if (mloc >= tiplen) break; }
        }
        while (mloc >= 0 && mloc < n && match[mloc] == 'm') { mloc--; neutral--; }
    }
    if (insertion <= maxIndel && deletion <= 4 * maxIndel) return false;
    int sum = neutral + insertion + deletion;
    if (deletion > 0) {
        int i = 0, j = 0;
        for (; i < sum; i++) if (match[i] != 'D') match[j++] = match[i];
        for (; i < n; i++, j++) match[j] = match[i];
        n -= deletion;
    }
    sum = neutral + insertion;
    for (int i = 0; i < sum; i++) match[i] = 'C';
    set_start(ss, ss.start - (insertion - deletion));
    return true;
}
// SiteScore.clipRightTipIndel (:577-636)
__device__ bool clip_right_tip_indel(Site &ss, uint8_t *match, int &n, int tiplen, int maxIndel) {
    if (!match || n < maxIndel) return false;
    const int lastIndex = n - 1;
    if (match[lastIndex] == 'C' || match[lastIndex] == 'Y' || match[lastIndex] == 'X') return false;
    int neutral = 0, insertion = 0, deletion = 0;
    {
        int mloc = lastIndex;
        for (const int mn = lastIndex - tiplen; mloc >= 0; mloc--) {
            const uint8_t c = match[mloc];
            if (c == 'I') insertion++;
            else if (c == 'D') deletion++;
            else { neutral++; if (mloc <= mn) break; }
        }
        while (mloc >= 0 && mloc < n && match[mloc] == 'm') { mloc++; neutral--; }
    }
    if (insertion <= maxIndel && deletion <= 4 * maxIndel) return false;
    const int sum = neutral + insertion + deletion;
    const int limit = n - sum;
    if (deletion > 0) {
        int j = limit;
        for (int i = limit; i < n; i++) if (match[i] != 'D') match[j++] = match[i];
        n -= deletion;
    }
    for (int i = limit; i < n; i++) match[i] = 'C';
    set_stop(ss, ss.stop + (insertion - deletion));
    return true;
}
// SiteScore.unclip (:638-681)
__device__ void unclip(const Dev &D, const Site &ss, uint8_t *match, int n, const uint8_t *bases) {
    if (!match || n < 1) return;
    if (match[0] != 'C' && match[n - 1] != 'C') return;
    for (int rloc = ss.start, cloc = 0, mloc = 0; mloc < n; mloc++) {
        const uint8_t mm = match[mloc];
        if (mm == 'C') {
            const uint8_t c = bases[cloc], r = ca_get(D, ss.chrom, rloc);
            if (!fully_defined_b(c) || !fully_defined_b(r)) match[mloc] = 'N';
            else match[mloc] = (c == r ? 'm' : 'S');
            rloc++; cloc++;
        } else if (mm == 'm' || mm == 'N' || mm == 'S' || mm == 'X' || mm == 'Y') { rloc++; cloc++; }
        else if (mm == 'I') cloc++;
        else if (mm == 'D') rloc++;
    }
}
// SiteScore.clipTipIndels(bases, tiplen, maxIndel, msa) (:497-521) on the site's pool string
__device__ bool clip_tip_indels(const Dev &D, Site &ss, const uint8_t *bases, int L, int tiplen, int maxIndel) {
    if (SITE_MATCH(ss) == 0 || SITE_MLEN(ss) < maxIndel) return false;
    uint8_t *match = pool_ptr(D, SITE_MATCH(ss)); int n = SITE_MLEN(ss);
    const bool left = clip_left_tip_indel(ss, match, n, tiplen, maxIndel);
    const bool right = clip_right_tip_indel(ss, match, n, tiplen, maxIndel);
    site_set_match(ss, SITE_MATCH(ss), n);
    if (left || right) {
        unclip(D, ss, match, n, bases);
        const int oldScore = ss.slowScore;
        set_slow_score(ss, msa_score_match(D.S, match, n));
        ss.score = ss.score + (ss.slowScore - oldScore);
        set_perfect(ss, bases, L, D.chromArr[ss.chrom], D.chromArrLen[ss.chrom]);
    }
    return left | right;
}

// SiteScore.fixXY(bases, nullifyOnFailure = false, msa) (:686-838)
__device__ bool fix_xy(const Dev &D, Site &ss, const uint8_t *bases, int L) {
    if (SITE_MATCH(ss) == 0) return true;
    uint8_t *match = pool_ptr(D, SITE_MATCH(ss)); const int n = SITE_MLEN(ss);
    if (!match_contains_xy(match, n)) return true;
    bool success = true;
    {
        int mloc = 0;
        while (mloc < n && (match[mloc] == 'X' || match[mloc] == 'Y')) mloc++;
        if (mloc >= n || mloc >= L) success = false;
        else if (mloc > 0) {
            mloc--;
            const int numX = mloc + 1;
            int rloc = ss.start + mloc, cloc = mloc, subs = 0, firstSub = -1;
            while (mloc >= 0) {
                const uint8_t c = bases[cloc], r = ca_get(D, ss.chrom, rloc);
                if (r == 'N' || c == 'N') match[mloc] = 'N';
                else if (c == r) match[mloc] = 'm';
                else { match[mloc] = 'S'; subs++; if (subs == 1) firstSub = mloc; }
                mloc--; rloc--; cloc--;
            }
            if ((ss.stop - ss.start + 1) != calc_match_length(match, n)) set_start(ss, ss.start - numX);
            if (subs > 5 && (float)subs > __fmul_rn((float)numX, 0.4f)) for (int i = 0; i <= firstSub; i++) match[i] = 'C';
        }
    }
    if (success) {
        int mloc = n - 1;
        while (mloc >= 0 && (match[mloc] == 'X' || match[mloc] == 'Y')) mloc--;
        const int dif = n - 1 - mloc;
        if (mloc < 0) success = false;
        else if (dif > 0) {
            mloc++;
            const int numX = n - mloc;
            int rloc = ss.stop - dif + 1, cloc = L - dif, subs = 0, firstSub = -1;
            if (cloc < 0) success = false;
            else {
                while (mloc < n) {
                    const uint8_t c = bases[cloc], r = ca_get(D, ss.chrom, rloc);
                    if (r == 'N' || c == 'N') match[mloc] = 'N';
                    else if (c == r) match[mloc] = 'm';
                    else { match[mloc] = 'S'; subs++; if (subs == 1) firstSub = mloc; }
                    mloc++; rloc++; cloc++;
                }
            }
            if (success) {
                if ((ss.stop - ss.start + 1) != calc_match_length(match, n)) set_stop(ss, ss.stop + numX);
                if (subs > 5 && (float)subs > __fmul_rn((float)numX, 0.4f)) for (int i = firstSub; i < n; i++) match[i] = 'C';
            }
        }
    }
    success = success && !match_contains_xy(match, n);
    const int oldScore = ss.slowScore;
    set_slow_score(ss, msa_score_match(D.S, match, n));
    ss.score = ss.score + (ss.slowScore - oldScore);
    set_perfect(ss, bases, L, D.chromArr[ss.chrom], D.chromArrLen[ss.chrom]);
    return success;
}
__device__ void fix_limits_xy(const Dev &D, Site &ss) {                         // SiteScore.fixLimitsXY (:916-931)
    if (SITE_MATCH(ss) == 0 || SITE_MLEN(ss) < 1) return;
    const uint8_t *m = pool_ptr(D, SITE_MATCH(ss)); const int n = SITE_MLEN(ss);
    int y = 0;
    for (int i = n - 1; i >= 0; i--) { if (m[i] == 'Y') y++; else break; }
    if (y != 0) set_limits(ss, ss.start, ss.stop + y);
}

// MSA.scoreNoIndelsAndMakeMatchString (MultiStateAligner11tsJNI.java:1245-1318): score and the m / S / N string
__device__ int score_no_indels_match(const Settings &S, const uint8_t *read, int len, const uint8_t *ref, int reflen, int refStart, uint8_t *match) {
    if (refStart < 0 || refStart + len > reflen) return -99999;
    int score = 0, mode = -1, t = 0;
    unsigned *mw = reinterpret_cast<unsigned *>(match);              // (pool strings start on a 4-byte boundary: four symbols per store)
    unsigned acc = 0;
    Words A, B;                                                      // (aligned word loads of the read and the reference, as scoreNoIndels)
    A.init(read, len); B.init(ref + refStart, len);
    for (int i0 = 0; i0 < len; i0 += 4) {
        const unsigned c4 = A.next(), r4 = B.next();
        acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + q;
            if (i < len) {
                const int c = (int)((c4 >> (8 * q)) & 255u), r = (int)((r4 >> (8 * q)) & 255u);
                unsigned sym;
                if (c == r && c != 'N') { if (mode == 0) { t++; score += S.ptsMatch2; } else { t = 0; score += S.ptsMatch; } sym = 'm'; mode = 0; }
                else if (c >= 128 || c == 'N') sym = 'N';
                else if (r >= 128 || r == 'N') sym = 'N';
                else { sym = 'S'; if (mode == 1) t++; else t = 0; score += (t + 1 > 5 ? S.ptsSub3 : (t + 1 > 1 ? S.ptsSub2 : S.ptsSub)); mode = 1; }
                acc |= sym << (8 * q);
            }
        }
        if (i0 + 4 <= len) mw[i0 >> 2] = acc;
    }
    for (int i = len & ~3; i < len; i++) match[i] = (uint8_t)(acc >> (8 * (i & 3)));
    return score;
}

// the padding adjustment that appears five times in realign_new (:380-405 and its copies); noFloorWhenUngapped = the minus strand's
// first copy (:534-555), which lacks the else-branch when there is no gap array
__device__ void adjust_extra_pad(const Settings &S, int L, const Site &ss, int minLoc, int maxLoc, int &epl, int &epr, bool noFloorWhenUngapped) {
    const int maxColumns = S.msaMaxColumns;
    int newlen;
    if (ss.ngaps == 0) newlen = (maxLoc - minLoc + 1 + epl + epr);
    else { Site t = ss; t.start = minLoc; t.stop = maxLoc; newlen = (imax(L, calc_gref_len(t)) + 1 + epl + epr); }
    if (newlen >= maxColumns - 80) {
        while (newlen >= maxColumns - 80 && epl > epr) { newlen--; epl--; }
        while (newlen >= maxColumns - 80 && epl < epr) { newlen--; epr--; }
        while (newlen >= maxColumns - 80) { newlen -= 2; epl--; epr--; }
    } else if (!(noFloorWhenUngapped && ss.ngaps == 0)) {
        const int x = imax(0, imin(20, ((maxColumns - newlen) / 2) - 40));
        epl = imax(x, epl); epr = imax(x, epr);
    }
}
// greflimit of makeGref(ref, gaps, a, b) (MultiStateAligner11tsJNI.java:668-757): length of the gapped reference of window [a, b]
__device__ int gref_limit(const Site &ss, int a, int b) {
    int g0 = imin(ss.gaps[0], a), gN = imax(ss.gaps[ss.ngaps - 1], b), gpos = 0;
    for (int i = 0; i < ss.ngaps; i += 2) {
        const int x = i == 0 ? g0 : ss.gaps[i], y = (i + 1 == ss.ngaps - 1) ? gN : ss.gaps[i + 1];
        gpos += y - x + 1;
        if (i + 2 < ss.ngaps) { const int gap = ss.gaps[i + 2] - y - 1; gpos += 64 + gap % GAPLEN + (gap - GAPBUFFER2) / GAPLEN + 64; }
    }
    return gpos;
}

// one fill of realign_new for the plain or the wide log (kind 3 first fill, 4 padded refill, 5 third fill, 6 fillUnlimited).  The
// slot comes from the caller: final_round_kernel takes a wavefront's slots with ONE atomicAdd per log (570 k same-address atomics,
// one per fill, made the first round's kernel 8.5 ms).
__device__ inline bool final_fill_is_wide(const Dev &D, const Site &ss, int minLoc, int maxLoc) {
    return ss.ngaps || (maxLoc - minLoc + 1) > D.plainColumns;
}
__device__ void write_final_fill(const Dev &D, long long r, const bbidx_read &rr, const Site &ss, int site, int minLoc, int maxLoc, int minscore, int kind, int seq,
                                 bool wide, unsigned k) {
    bbmap_jobinfo info; info.read = (int)r; info.seq = seq; info.kind = kind; info.site = site;
    bbmsa_job j;
    j.read_off = rr.bases_off + (ss.strand ? D.minusDelta : 0);
    j.ref_off = (long long)(D.chromArr[ss.chrom] - D.refsBase);
    j.read_len = rr.len; j.ref_len = D.chromArrLen[ss.chrom];
    j.refStartLoc = minLoc; j.refEndLoc = maxLoc; j.minScore = minscore;
    // (gap symbols stay compact in the log's string: a long deletion's 'D's would not fit its slot; the pool copy expands them)
    j.flags = (kind == 6 ? BBMSA_FILL_UNLIMITED_RAW : BBMSA_FILL_LIMITED) | BBMSA_DO_SCORE | BBMSA_DO_TRACEBACK | BBMSA_TRACE_KEEP_GAPS;
    if (wide) {
        D.gjobs[k] = j; D.ginfo[k] = info;
        bbmsa_gaps g; g.ngaps = ss.ngaps;
        for (int q = 0; q < BBMSA_MAX_GAPS; q++) g.gaps[q] = q < ss.ngaps ? ss.gaps[q] : 0;
        D.ggaps[k] = g;
    } else { D.jobs[k] = j; D.jinfo[k] = info; }
}
__device__ inline const uint8_t *fill_match(const Dev &D, int job) {
    return (job & GAPPED_BIT) ? D.gmatch + (long long)(job & ~GAPPED_BIT) * D.gmatchStride : D.match + (long long)job * D.matchStride;
}

// ---------------------------------------------------------------------------------------------- Read setters
__device__ inline void r_clear_site(FinalRead &r) { r.chrom = -1; r.strand = 0; r.start = -1; r.stop = -1; r.mapScore = 0; }
__device__ inline void r_set_from_site(FinalRead &r, Site &ss) {                // Read.setFromSite (:1172-1190)
    r.chrom = ss.chrom; r.strand = ss.strand; r.start = ss.start; r.stop = ss.stop; r.mapScore = ss.slowScore;
    r.rescued = ss.rescued; r.perfect = ss.perfect; r.match = SITE_MATCH(ss); r.matchLen = SITE_MLEN(ss);
    if (ss.ngaps) fix_gaps(ss);
}
__device__ inline void r_set_from_top_site(FinalRead &r, Site *s, int n) {      // Read.setFromTopSite(false, true, .) (:1213-1225)
    if (n == 0) { r_clear_site(r); r.mapped = 0; return; }
    r.mapped = 1;
    r_set_from_site(r, s[0]);
}
__device__ inline void r_clear_mapping(FinalRead &r, int &n) {                  // Read.clearMapping (:1269-1276); the mate's flag: by the caller
    r_clear_site(r); r.match = 0; r.matchLen = 0; n = 0; r.mapped = 0; r.paired = 0;
}

__device__ int count_top_scores(const Site *s, int n, int thresh) {             // Tools.countTopScores (Tools.java:913-930)
    if (n == 0) return 0;
    int count = 1; const int limit = s[0].score - thresh;
    for (int i = 1; i < n; i++) {
        if (s[i].score < limit) break;
        if (s[0].start != s[i].start && s[0].stop != s[i].stop) count++;
    }
    return count;
}
__device__ inline void remove_at(Site *s, int &n, int i) { for (int j = i + 1; j < n; j++) s[j - 1] = s[j]; n--; }
__device__ void remove_duplicate_best_sites(Site *s, int &n) {                  // AbstractMapThread.java:1328-1349
    if (n < 2) return;
    for (int i = n - 1; i > 0; i--) {
        if (s[0].chrom == s[i].chrom && s[0].strand == s[i].strand && s[0].start == s[i].start && s[0].stop == s[i].stop) remove_at(s, n, i);
        else break;
    }
}
// Tools.mergeDuplicateSites(list, false, false) (Tools.java:697-759): exact positional matches only
__device__ int merge_duplicate_sites_exact(Site *s, int n) {
    if (n < 2) return n;
    sort_sites<true>(s, n);
    DeadSet dead;
    int ai = 0;
    for (int i = 1; i < n; i++) {
        Site &a = s[ai];
        const Site &b = s[i];
        if (positional_match(a, b, true)) {
            set_slow_score(a, imax(a.slowScore, b.slowScore));
            a.pairedScore = (a.pairedScore <= a.slowScore && b.pairedScore <= a.slowScore) ? 0 : imax(0, imax(a.pairedScore, b.pairedScore));
            a.score = imax(a.score, b.score);
            a.perfect = (a.perfect || b.perfect);
            a.semiperfect = (a.semiperfect || b.semiperfect);
            dead.mark(s, i);
        } else ai = i;
    }
    // (DeadSet's marks beyond position 63 live in reserved[1], which holds the match length here: lists that long only exist in the
    // overflow tier; their marks are written over the length of a record that is being removed)
    return condense(s, n, dead);
}

__device__ int clearzone_single(const Settings &S, bool perfect, int score, int maxSw) {       // BBMapThread.java:508-525
    const int M2 = S.ptsMatch2;
    const int CZ1 = (int)__fmul_rn(2.0f, (float)M2), CZ1b = (int)__fmul_rn(2.6f, (float)M2), CZ1c = (int)__fmul_rn(4.6f, (float)M2), CZP = (int)__fmul_rn(1.6f, (float)M2);
    if (perfect) return CZP;
    const float cz1blimit = __fsub_rn(__fmul_rn((float)maxSw, 0.97f), __fmul_rn(12.f, (float)M2));
    const float cz1climit = __fsub_rn(__fmul_rn((float)maxSw, 0.92f), __fmul_rn(26.f, (float)M2));
    if ((float)score > cz1blimit)
        return (int)__fdiv_rn(__fadd_rn((float)((maxSw - score) * CZ1b), __fmul_rn(__fsub_rn((float)score, cz1blimit), (float)CZ1)), __fsub_rn((float)maxSw, cz1blimit));
    if ((float)score > cz1climit)
        return (int)__fdiv_rn(__fadd_rn(__fmul_rn(__fsub_rn(cz1blimit, (float)score), (float)CZ1c), __fmul_rn(__fsub_rn((float)score, cz1climit), (float)CZ1b)), __fsub_rn(cz1blimit, cz1climit));
    return CZ1c;
}
__device__ int clearzone_paired(const Settings &S, bool perfect, int score, int maxSw) {        // :1158-1160
    const int M2 = S.ptsMatch2;
    const int CZ1 = (int)__fmul_rn(2.0f, (float)M2), CZ1b = (int)__fmul_rn(2.6f, (float)M2), CZ1c = (int)__fmul_rn(4.6f, (float)M2), CZP = (int)__fmul_rn(1.6f, (float)M2);
    if (perfect) return CZP;
    if (score >= (int)__fsub_rn(__fmul_rn((float)maxSw, 0.97f), __fmul_rn(12.f, (float)M2))) return CZ1;
    if (score >= (int)__fsub_rn(__fmul_rn((float)maxSw, 0.92f), __fmul_rn(26.f, (float)M2))) return CZ1b;
    return CZ1c;
}

// pairSiteScoresFinal(r, r2, trim = true, setScore = true, ...) (AbstractMapThread.java:1919-2095)
__device__ void pair_final(const Settings &S, Site *s1, int &n1, Site *s2, int &n2, int len1, int len2) {
    for (int i = 0; i < n1; i++) s1[i].pairedScore = 0;
    for (int i = 0; i < n2; i++) s2[i].pairedScore = 0;
    if (n1 < 1 || n2 < 1) return;
    sort_sites<true>(s1, n1); sort_sites<true>(s2, n2);
    int maxPaired1 = -1, maxPaired2 = -1;
    const float q1 = __fdiv_rn((float)len1, __fmul_rn(4.f, (float)len2)), q2 = __fdiv_rn((float)len2, __fmul_rn(4.f, (float)len1));
    const float h1 = 0.25f > q1 ? 0.25f : q1, h2 = 0.25f > q2 ? 0.25f : q2;
    const float mult1 = 0.5f < h1 ? 0.5f : h1, mult2 = 0.5f < h2 ? 0.5f : h2;
    const int ilimit = n1 - 1, jlimit = n2 - 1;
    const int outerDistLimit = (imax(len1, len2) * OUTER_DIST_MULT) / OUTER_DIST_DIV;
    const int expectedFragLength = S.averagePairDist + len1 + len2, MPD = S.maxPairDist;
    const int den = imax(100, (10 * expectedFragLength + 100));
    for (int i = 0, j = 0; i <= ilimit && j <= jlimit; i++) {
        Site &a = s1[i];
        while (j < jlimit && (s2[j].chrom < a.chrom || (s2[j].chrom == a.chrom && a.start - s2[j].stop > MPD))) j++;
        for (int k = j; k <= jlimit; k++) {
            Site &b = s2[k];
            if (b.chrom > a.chrom) break;
            if (b.start - a.stop > MPD) break;
            int innerdist, outerdist;
            if (a.strand != b.strand) {
                if (a.strand == 0) { innerdist = b.start - a.stop; outerdist = b.stop - a.start; }
                else { innerdist = a.start - b.stop; outerdist = a.stop - b.start; }
            } else if (a.start <= b.start) { innerdist = b.start - a.stop; outerdist = b.stop - a.start; }
            else { innerdist = a.start - b.stop; outerdist = a.stop - b.start; }
            if (outerdist >= outerDistLimit && innerdist <= MPD && a.strand != b.strand) {
                const int deviation = iabsdif(S.averagePairDist, innerdist);
                const int ps1 = a.score + 1 + imax(1, (int)__fmul_rn((float)b.score, mult1) - ((deviation * b.score) / den));
                const int ps2 = b.score + 1 + imax(1, (int)__fmul_rn((float)a.score, mult2) - ((deviation * a.score) / den));
                a.pairedScore = imax(a.pairedScore, ps1);
                b.pairedScore = imax(b.pairedScore, ps2);
                maxPaired1 = imax(a.score, maxPaired1);
                maxPaired2 = imax(b.score, maxPaired2);
            }
        }
    }
    for (int i = 0; i < n1; i++) if (s1[i].pairedScore > s1[i].score) s1[i].score = s1[i].pairedScore;
    for (int i = 0; i < n2; i++) if (s2[i].pairedScore > s2[i].score) s2[i].score = s2[i].pairedScore;
    n1 = trim_below_cutoff(s1, n1, (int)__fmul_rn((float)maxPaired1, 0.95f), false, 1, S.maxTrimSitesToRetain);
    n2 = trim_below_cutoff(s2, n2, (int)__fmul_rn((float)maxPaired2, 0.95f), false, 1, S.maxTrimSitesToRetain);
}
__device__ bool can_pair(const Site &a, const Site &b, int len1, int len2, int MPD) {            // :2097-2164
    if (a.chrom != b.chrom || a.strand == b.strand) return false;
    const int outerDistLimit = (imax(len1, len2) * OUTER_DIST_MULT) / OUTER_DIST_DIV;
    int innerdist, outerdist;
    if (a.strand == 0) { innerdist = b.start - a.stop; outerdist = b.stop - a.start; }
    else { innerdist = a.start - b.stop; outerdist = a.stop - b.start; }
    return outerdist >= outerDistLimit && innerdist <= MPD;
}

__device__ inline void fin_init(FinalRead &f, int seq) {
    f.mapped = 0; f.paired = 0; f.ambiguous = 0; f.perfect = 0; f.rescued = 0; f.chrom = -1; f.strand = 0; f.start = -1; f.stop = -1; f.mapScore = 0;
    f.match = 0; f.matchLen = 0; f.pc = PC_DONE; f.i = 0; f.best = INT_MIN; f.scoreChanged = 0; f.sorting = 0; f.topObj_ = 0; f.pairedLost = 0;
    f.oldSlow = 0; f.oldScoreS = 0; f.oldScoreG = 0; f.gstep = 0; f.recur = 0; f.padding = 0; f.forbidIndels = 0; f.fixXY = 0; f.minValid = 0;
    f.scoreNoIndel = 0; f.minLoc = 0; f.maxLoc = 0; f.old0 = 0; f.epl = 0; f.epr = 0; f.fillKind = 0; f.minscore = 0; f.pending = -1; f.haveMax = 0; f.cols3 = 0;
    f.seq = seq; f.needLocal = 0; f.reservedI = 0;
}
__device__ inline void tag_objects(Site *s, int n) { for (int i = 0; i < n; i++) { s[i].reserved[0] = 0; s[i].reserved[1] = 0; } }
__device__ inline void gen_begin(FinalRead &f, int n) {      // genMatchString's entry: `if(USE_SS_MATCH_FOR_PRIMARY && topSite().match!=null)` never holds here
    if (n > 0) { f.pc = PC_SITE_LOOP; f.i = 0; f.best = INT_MIN; f.scoreChanged = 0; f.sorting = 0; }
}

// ---------------------------------------------------------------------------------------------- kernel 1: policy before genMatchString
// single-ended: BBMapThread.java:504-557; paired: :1116-1206.  One thread per read / pair.
__global__ __launch_bounds__(128) void final_begin_kernel(const Dev D) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const Settings &S = D.S;
    if (S.paired) {
        if (2 * u + 1 >= D.nreads) return;
        const long long r1 = 2 * u, r2 = r1 + 1;
        FinalRead f1, f2; fin_init(f1, D.slow[r1].seq); fin_init(f2, D.slow[r2].seq);
        int n1 = D.mcount[r1], n2 = D.mcount[r2];
        if (n1 < 0 || n2 < 0) { D.fin[r1] = f1; D.fin[r2] = f2; return; }                       // flagged pair (overflow): nothing to finish here
        Site *s1 = D.ms + r1 * D.cap, *s2 = D.ms + r2 * D.cap;
        const int len1 = D.reads[r1].len, len2 = D.reads[r2].len, maxSw1 = max_quality(S, len1), maxSw2 = max_quality(S, len2);
        if (n1 > 1) sort_sites<false>(s1, n1);
        if (n2 > 1) sort_sites<false>(s2, n2);
        n1 = remove_low_quality_paired(s1, n1, maxSw1, S.minRatio, S.ratioPaired);
        n2 = remove_low_quality_paired(s2, n2, maxSw2, S.minRatio, S.ratioPaired);
        pair_final(S, s1, n1, s2, n2, len1, len2);
        if (n1 > 0) sort_sites<false>(s1, n1);
        if (n2 > 0) sort_sites<false>(s2, n2);
        tag_objects(s1, n1); tag_objects(s2, n2);
        f1.perfect = n1 > 0 && (s1[0].slowScore == maxSw1 || s1[0].perfect);                    // Read.setPerfectFlag (Read.match is null here)
        f2.perfect = n2 > 0 && (s2[0].slowScore == maxSw2 || s2[0].perfect);
        if (n1 > 1 && count_top_scores(s1, n1, clearzone_paired(S, f1.perfect, s1[0].score, maxSw1)) > 1) f1.ambiguous = 1;
        if (n2 > 1 && count_top_scores(s2, n2, clearzone_paired(S, f2.perfect, s2[0].score, maxSw2)) > 1) f2.ambiguous = 1;
        if (n1 > 0 && n2 > 0 && can_pair(s1[0], s2[0], len1, len2, S.maxPairDist)) { f1.paired = 1; f2.paired = 1; }
        r_set_from_top_site(f1, s1, n1); r_set_from_top_site(f2, s2, n2);
        gen_begin(f1, n1); gen_begin(f2, n2);
        D.mcount[r1] = n1; D.mcount[r2] = n2;
        D.fin[r1] = f1; D.fin[r2] = f2;
    } else {
        if (u >= D.nreads) return;
        FinalRead f; fin_init(f, D.slow[u].seq);
        int n = D.mcount[u];
        if (n < 0) { D.fin[u] = f; return; }
        Site *s = D.ms + u * D.cap;
        const int L = D.reads[u].len, maxSw = max_quality(S, L);
        tag_objects(s, n);
        f.perfect = n > 0 && (s[0].slowScore == maxSw || s[0].perfect);
        if (n > 1) {
            const int score = s[0].score;
            const int clearzone = clearzone_single(S, f.perfect, score, maxSw);
            int numBest = count_top_scores(s, n, clearzone);
            if (numBest > 1) f.ambiguous = 1;
            else {
                const int lim = (f.perfect ? 160 : (score + S.clearzone1e >= maxSw ? 80 : 40)) + 1;       // CLEARZONE_LIMIT1e = 40
                if (n > lim && clearzone < S.clearzone1e) { numBest = count_top_scores(s, n, S.clearzone1e); if (numBest > lim) f.ambiguous = 1; }
            }
        }
        if (n > 0) {
            const int lim = (int)__fmul_rn((float)maxSw, S.minRatio);
            if (s[0].score < lim) n = 0;
            else {                                         // Tools.removeLowQualitySitesUnpaired: never positions 0 and 1
                const int thresh = imin(lim, imax(1, lim - S.clearzone3));
                for (int i = n - 1; i > 1; i--) if (s[i].slowScore < thresh) remove_at(s, n, i);
            }
        }
        r_set_from_top_site(f, s, n);
        gen_begin(f, n);
        D.mcount[u] = n;
        D.fin[u] = f;
    }
}

// ---------------------------------------------------------------------------------------------- kernel 2: one round of genMatchString
// Advances a read until it needs a fill (returns true: still active) or has finished genMatchString (PC_DONE).
__device__ bool final_advance(const Dev &D, long long r, FinalRead &f, PreAlloc &pre, bool &wantEmit) {
    const Settings &S = D.S;
    const bbidx_read rr = D.reads[r];
    const int L = rr.len, maxSw = max_quality(S, L), maxI = max_imperfect(S, L);
    Site *s = D.ms + r * D.cap;
    int n = D.mcount[r];
    const bool single = !S.paired;
    for (;;) {
        const int cur = f.sorting ? 0 : f.i;                                   // the site being worked on
        switch (f.pc) {
        case PC_DONE: D.mcount[r] = n; return false;
        case PC_SITE_LOOP: {                                                    // `for(int i=0; i<r.sites.size(); i++)` (:882-907)
            if (f.i >= n) { f.pc = PC_AFTER_LOOP; break; }
            const Site &ss = s[f.i];
            if (f.i > 0 && f.best >= ss.slowScore) { f.pc = PC_AFTER_LOOP; break; }
            f.oldSlow = ss.slowScore; f.oldScoreS = ss.score;
            f.pc = SITE_MATCH(ss) == 0 ? PC_GEN_START : PC_SITE_DONE;
            break;
        }
        case PC_GEN_START: {                                                    // genMatchStringForSite (:968-1001)
            Site ss = s[cur];
            if (ss.perfect) {
                const int ref = pool_take(D, pre, L);
                if (!ref) { D.mcount[r] = n; return true; }
                uint8_t *m = pool_ptr(D, ref);
                unsigned *mw = reinterpret_cast<unsigned *>(m);                 // (4-byte aligned, see pool_units)
                for (int q = 0; q < (L >> 2); q++) mw[q] = 0x6d6d6d6du;
                for (int q = L & ~3; q < L; q++) m[q] = 'm';
                site_set_match(ss, ref, L); s[cur] = ss;
                f.pc = PC_GEN_CLIP;
            } else {
                f.oldScoreG = ss.slowScore;
                f.padding = (ss.perfect || ss.semiperfect) ? 0 : imax(S.slowAlignPadding, 6);
                f.recur = 1; f.forbidIndels = S.maxIndel < 1; f.fixXY = 0; f.gstep = 0;
                f.minValid = -1 + (int)__fmul_rn(S.paired ? S.ratioPaired : S.minRatio, (float)maxSw);
                f.pc = PC_REALIGN_START;
            }
            break;
        }
        case PC_REALIGN_START: {                                                // realign_new up to its first fill (:229-370, :487-522)
            const int ref = pool_take(D, pre, L);                               // the string scoreNoIndelsAndMakeMatchString writes
            if (!ref) { D.mcount[r] = n; return true; }
            Site ss = s[cur];
            const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
            const uint8_t *chr = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom], maxIndex = reflen - 1;
            if (SITE_MATCH(ss) && match_contains_xy(pool_ptr(D, SITE_MATCH(ss)), SITE_MLEN(ss))) fix_xy(D, ss, bases, L);
            clip_tip_indels(D, ss, bases, L, 4, 10);
            int padding = imax(imin(f.padding, (S.msaMaxColumns - L) / 2 - 20), 0);
            if (calc_gref_len(ss) > S.msaMaxColumns - 20) { set_stop(ss, ss.start + imin(L + 40, S.msaMaxColumns - 20)); if (ss.ngaps) fix_gaps(ss); }
            if (ss.start < 0) set_start(ss, 0);
            if (ss.stop > maxIndex) set_stop(ss, maxIndex);
            { const int b = ss.stop - ss.start + 1; if (b < L) padding = imax(padding, imin(L, L - b + 10) / 2 + 1); }
            padding = imax(0, imin(padding, (S.msaMaxColumns - imax(L, calc_gref_len(ss))) / 2 - 100));
            if (f.forbidIndels) padding = 0;
            uint8_t *m = pool_ptr(D, ref);
            f.scoreNoIndel = score_no_indels_match(S, bases, L, chr, reflen, ss.start, m);
            site_set_match(ss, ref, L);
            if (f.scoreNoIndel >= maxI || f.forbidIndels) {
                set_stop(ss, ss.start + L - 1); set_slow_score(ss, f.scoreNoIndel);
                s[cur] = ss; f.pc = PC_REALIGN_POST;
            } else {
                f.minLoc = imax(ss.start - padding, 0); f.maxLoc = imin(ss.stop + padding, maxIndex);
                f.minscore = imax(f.scoreNoIndel, f.minValid); f.fillKind = 3; f.haveMax = 0;
                s[cur] = ss; f.pc = PC_EMIT_FILL;
            }
            break;
        }
        case PC_EMIT_FILL:                                                      // the kernel emits (a wavefront's fills together) and parks the read
            D.mcount[r] = n;
            wantEmit = true;
            return true;
        case PC_FILL_BACK: {                                                    // a fill came back (:371-483, :523-622)
            const bbmsa_result &res = fill_result(D, f.pending);
            const int nsc = res.status == BBMSA_ST_OK ? res.score_len : 0;
            Site ss = s[cur];
            const int maxIndex = D.chromArrLen[ss.chrom] - 1;
            bool again = false;
            if (f.fillKind == 3 && nsc > 6) {
                f.old0 = res.score[0]; f.epl = res.score[6]; f.epr = res.score[7];
                adjust_extra_pad(S, L, ss, f.minLoc, f.maxLoc, f.epl, f.epr, ss.strand != 0);
                f.minLoc = imax(0, f.minLoc - f.epl); f.maxLoc = imin(maxIndex, f.maxLoc + f.epr);
                f.fillKind = 4; again = true;
            } else if (f.fillKind == 4 && (nsc == 0 || res.score[0] < f.old0)) {
                adjust_extra_pad(S, L, ss, f.minLoc, f.maxLoc, f.epl, f.epr, false);
                f.minLoc = imax(0, f.minLoc - f.epl); f.maxLoc = imin(maxIndex, f.maxLoc + f.epr);
                f.fillKind = 5; again = true;
            } else if (f.fillKind == 5 && ss.strand == 0 && f.minLoc > 0 && f.maxLoc < maxIndex && (nsc == 0 || res.score[0] < f.old0)) {
                // fillUnlimited (:453-459, plus strand only).  The JNI class's direct fillUnlimited does not set its rows / columns fields
                // (MultiStateAligner11tsJNI.java:178-192): traceback2's 'Y' test (:448) then sees the columns of the third fill
                f.cols3 = ss.ngaps ? gref_limit(ss, f.minLoc, f.maxLoc) + 1 : f.maxLoc - f.minLoc + 1;
                f.minLoc = imax(ss.start - 8, 0); f.maxLoc = imin(ss.stop + 8, maxIndex);
                f.fillKind = 6; again = true;
            }
            if (again) { f.pc = PC_EMIT_FILL; break; }
            if (nsc > 0) {                                                      // max != null: traceback, limits, score (:469-476)
                const int clen = res.match_len > 0 ? res.match_len : 0;          // the log's string, gap symbols compact
                uint8_t *src = const_cast<uint8_t *>(fill_match(D, f.pending));
                int gsyms = 0;
                if (ss.ngaps) for (int q = 0; q < clen; q++) gsyms += src[q] == '-';
                const int mlen = clen + gsyms * (GAPLEN - 1);                    // traceback2 :481-493: each '-' stands for 128 'D'
                const int ref = pool_take(D, pre, mlen);
                if (!ref) { D.mcount[r] = n; return true; }
                if (f.fillKind == 6 && ss.ngaps == 0) {
                    // the stale `columns` of the JNI class (see above): an insertion is 'Y' at or beyond the THIRD fill's column count.  Redone
                    // on the log's string (idempotent); not emulated for gapped references, whose third fill counts gapped columns
                    int col = res.score[1] - f.minLoc;                          // column of the cell before the path's first symbol
                    for (int q = 0; q < clen; q++) {
                        const uint8_t c = src[q];
                        if (c == 'I' || c == 'Y') src[q] = (col >= f.cols3) ? 'Y' : 'I';
                        else col++;                                              // (leading 'X's: score[1] lies that many columns left of the window)
                    }
                }
                uint8_t *m = pool_ptr(D, ref);
                if (gsyms == 0) {                                               // (log slots and pool strings both start on 4-byte boundaries)
                    const unsigned *sw = reinterpret_cast<const unsigned *>(src);
                    unsigned *dw = reinterpret_cast<unsigned *>(m);
                    for (int q = 0; q < (clen >> 2); q++) dw[q] = sw[q];
                    for (int q = clen & ~3; q < clen; q++) m[q] = src[q];
                } else {
                    int o = 0;
                    for (int q = 0; q < clen; q++) { const uint8_t ch = src[q]; if (ch != '-') m[o++] = ch; else for (int g = 0; g < GAPLEN; g++) m[o++] = 'D'; }
                }
                site_set_match(ss, ref, mlen);
                set_limits(ss, res.score[1], res.score[2]);
                fix_limits_xy(D, ss);
                set_slow_score(ss, res.score[0]);
                ss.match_job = f.pending;
            } else { set_stop(ss, ss.start + L - 1); set_slow_score(ss, f.scoreNoIndel); }
            s[cur] = ss; f.pc = PC_REALIGN_POST;
            break;
        }
        case PC_REALIGN_POST: {                                                 // :629-651
            Site ss = s[cur];
            const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
            const int maxIndex = D.chromArrLen[ss.chrom] - 1;
            const uint8_t *m = SITE_MATCH(ss) ? pool_ptr(D, SITE_MATCH(ss)) : nullptr; const int mlen = SITE_MLEN(ss);
            const int lp = left_padding_needed(m, mlen, 4, 5), rp = right_padding_needed(m, mlen, 4, 5);
            if (ss.stop < maxIndex && ss.start > 0 && (lp > 0 || rp > 0)) {
                if (f.recur > 0) {                                              // the tail recursion: realign_new again, recur - 1
                    fix_gaps(ss);
                    f.padding = imin(10 + imax(lp, rp), (S.msaMaxColumns - L) / 2 - 20);
                    f.recur--; s[cur] = ss; f.pc = PC_REALIGN_START;
                    break;
                }
                if (f.fixXY && match_contains_xy(m, mlen)) fix_xy(D, ss, bases, L);
            }
            set_perfect(ss, bases, L, D.chromArr[ss.chrom], D.chromArrLen[ss.chrom]);
            s[cur] = ss; f.pc = PC_GEN_AFTER_REALIGN;
            break;
        }
        case PC_GEN_AFTER_REALIGN: {                                            // genMatchStringForSite :1000-1034
            Site ss = s[cur];
            const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
            fix_gaps(ss);
            if (f.gstep == 0) {
                const uint8_t *m = SITE_MATCH(ss) ? pool_ptr(D, SITE_MATCH(ss)) : nullptr; const int mlen = SITE_MLEN(ss);
                const int lp = left_padding_needed(m, mlen, 4, 5), rp = right_padding_needed(m, mlen, 4, 5);
                if (ss.slowScore < f.oldScoreG || lp > 0 || rp > 0) {
                    int extra = (S.maxIndel > 0 ? 80 : 20) + S.slowAlignPadding;
                    const int remaining = (S.msaMaxColumns - calc_gref_len(ss) - 2);
                    extra = imax(0, imin(remaining / 2, extra));
                    f.padding = extra; f.recur = 2; f.forbidIndels = 0; f.fixXY = 1; f.gstep = 1;
                    s[cur] = ss; f.pc = PC_REALIGN_START;
                    break;
                }
            }
            if (maxSw == ss.slowScore) ss.perfect = ss.semiperfect = 1;         // SiteScore.setPerfectFlag
            else set_perfect(ss, bases, L, D.chromArr[ss.chrom], D.chromArrLen[ss.chrom]);
            s[cur] = ss; f.pc = PC_GEN_CLIP;
            break;
        }
        case PC_GEN_CLIP: {                                                     // :1064
            Site ss = s[cur];
            const uint8_t *bases = D.bases + rr.bases_off + (ss.strand ? D.minusDelta : 0);
            clip_tip_indels(D, ss, bases, L, 4, 10);
            if (single) ss.score = ss.slowScore;                                // `if(setSSScore){ss.setScore(ss.slowScore);}`
            s[cur] = ss;
            f.pc = f.sorting ? PC_SORT_LOOP : PC_SITE_DONE;
            break;
        }
        case PC_SITE_DONE: {                                                    // :898-904 (a generated match string is never null)
            const Site &ss = s[f.i];
            if (f.oldScoreS != ss.score || f.oldSlow != ss.slowScore) f.scoreChanged++;
            f.best = imax(ss.slowScore, f.best);
            f.i++; f.pc = PC_SITE_LOOP;
            break;
        }
        case PC_AFTER_LOOP: {                                                   // :911-916
            bool ordered = true;
            for (int q = 1; q < n; q++) if (s[q].score > s[q - 1].score) { ordered = false; break; }
            if (f.scoreChanged > 0 && !ordered) { f.sorting = 2; f.pc = PC_SORT_LOOP; }
            else f.pc = PC_FINISH;
            break;
        }
        case PC_SORT_LOOP: {                                                    // `while(needsSorting)` (:916-943); sorting: 1 = inside, 2 = (re-)enter
            if (f.sorting == 2) {
                f.sorting = 1;
                s[0].reserved[1] = (int)((unsigned)s[0].reserved[1] | SITE_TOPMARK);             // `final SiteScore top=r.topSite();`
                n = merge_duplicate_sites_exact(s, n);
                sort_sites<false>(s, n);
                if (n > 0 && SITE_MATCH(s[0]) == 0) { f.pc = PC_GEN_START; break; }            // comes back here with sorting = 1
                f.sorting = 0;
            }
            // `if(r.paired() && r.topSite()!=top)`, after the top site's genMatchStringForSite when it had none
            const bool sameTop = ((unsigned)s[0].reserved[1] & SITE_TOPMARK) != 0;
            for (int q = 0; q < n; q++) s[q].reserved[1] = (int)((unsigned)s[q].reserved[1] & ~SITE_TOPMARK);
            if (f.paired && !sameTop) { f.paired = 0; f.pairedLost = 1; }
            if (f.sorting == 1) f.sorting = 2;                                   // needsSorting = true: the loop body once more
            else f.pc = PC_FINISH;
            break;
        }
        case PC_FINISH: {                                                       // :946-959
            f.sorting = 0;
            const Site &ss = s[0];
            f.start = ss.start; f.stop = ss.stop; f.chrom = ss.chrom; f.strand = ss.strand;
            f.match = SITE_MATCH(ss); f.matchLen = SITE_MLEN(ss);
            f.mapScore = ss.slowScore; f.perfect = ss.perfect; f.rescued = ss.rescued;
            if (single) {                                                       // the do-while of processRead (:592-613)
                s[0].score = s[0].slowScore;
                if (n > 1 && s[0].score < s[1].score) {
                    sort_sites<false>(s, n);
                    r_set_from_top_site(f, s, n);
                    f.pc = PC_SITE_LOOP; f.i = 0; f.best = INT_MIN; f.scoreChanged = 0;
                    break;
                }
            }
            f.pc = PC_DONE;
            break;
        }
        default: f.pc = PC_DONE; break;
        }
    }
}

// bytes of the first string a read will ask for in this round (0 = none that can be told in advance): a site's first string has the
// read's length ('m's of a perfect site, or what scoreNoIndelsAndMakeMatchString writes); a fill that came back brings its traceback
__device__ int first_request(const Dev &D, long long r, const FinalRead &f) {
    if (f.pc == PC_SITE_LOOP || f.pc == PC_GEN_START || f.pc == PC_REALIGN_START) return D.reads[r].len;
    if (f.pc == PC_FILL_BACK) {
        const bbmsa_result &res = fill_result(D, f.pending);
        if (res.status != BBMSA_ST_OK || res.score_len <= 0 || res.match_len <= 0) return D.reads[r].len;       // (the next realign_new's string)
        const Site &ss = D.ms[r * D.cap + (f.sorting ? 0 : f.i)];
        int len = res.match_len;
        if (ss.ngaps) { const uint8_t *src = fill_match(D, f.pending); int g = 0; for (int q = 0; q < res.match_len; q++) g += src[q] == '-'; len += g * (GAPLEN - 1); }
        return len;
    }
    return 0;
}

// (Left to itself the compiler gives this state machine 248 VGPRs -- site records are 32 registers each -- and ONE wavefront per SIMD,
// for a kernel that waits on scattered loads.  Asked for 4 blocks per CU it keeps 128 and spills 112 bytes more per lane: the final
// stage 74.9 -> 72.6 ms; at 6 / 8 blocks (80 / 64 VGPRs) the spills cost more than the waves bring: 77.2 / 77.4.)
#ifndef FINAL_ROUND_MIN_BLOCKS
#define FINAL_ROUND_MIN_BLOCKS 4
#endif
__global__ __launch_bounds__(128, FINAL_ROUND_MIN_BLOCKS) void final_round_kernel(const Dev D) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long count = D.activeIn ? D.nActiveIn : D.nreads;
    const int lane = threadIdx.x & 63;
    bool stillActive = false;
    long long r = -1;
    FinalRead f; f.pc = PC_DONE;
    if (t < count) { r = D.activeIn ? D.activeIn[t] : t; f = D.fin[r]; }
    // one pool request per wavefront for the reads' first strings: inclusive prefix sum of the lanes' units, one atomicAdd
    PreAlloc pre; pre.ref = 0; pre.units = 0;
    {
        const int bytes = f.pc != PC_DONE ? first_request(D, r, f) : 0;
        const unsigned units = bytes > 0 ? pool_units(bytes) : 0u;
        unsigned incl = units;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        const unsigned total = __shfl(incl, 63, 64);
        int base = 0;
        if (total) { if (lane == 0) base = pool_alloc_units(D, total); base = __shfl(base, 0, 64); }
        if (base && units) { pre.ref = base + (int)(incl - units); pre.units = units; }
    }
    bool wantEmit = false;
    if (f.pc != PC_DONE) stillActive = final_advance(D, r, f, pre, wantEmit);
    {   // the fills this wavefront asks for: one atomicAdd per log, slots by rank.  A slot beyond the log's capacity is not written: the
        // read stays in PC_EMIT_FILL and asks again next round, before which the host has grown the log (as NO_ROOM in emit_fill)
        const int cur = f.sorting ? 0 : f.i;
        bool wide = false;
        if (wantEmit) wide = final_fill_is_wide(D, D.ms[r * D.cap + cur], f.minLoc, f.maxLoc);
        const unsigned long long mp = __ballot(wantEmit && !wide), mg = __ballot(wantEmit && wide);
        unsigned bp = 0, bg = 0;
        if (mp) { if (lane == __builtin_ctzll(mp)) bp = atomicAdd(&D.counters[0], (unsigned)__builtin_popcountll(mp)); bp = __shfl(bp, __builtin_ctzll(mp)); }
        if (mg) { if (lane == __builtin_ctzll(mg)) bg = atomicAdd(&D.counters[1], (unsigned)__builtin_popcountll(mg)); bg = __shfl(bg, __builtin_ctzll(mg)); }
        if (wantEmit) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned k = wide ? bg + (unsigned)__builtin_popcountll(mg & below) : bp + (unsigned)__builtin_popcountll(mp & below);
            if ((long long)k < (wide ? D.gjobCap : D.jobCap)) {
                write_final_fill(D, r, D.reads[r], D.ms[r * D.cap + cur], cur, f.minLoc, f.maxLoc, f.fillKind == 6 ? 0 : f.minscore, f.fillKind, f.seq, wide, k);
                f.seq++; f.pending = wide ? ((int)k | GAPPED_BIT) : (int)k; f.pc = PC_FILL_BACK;
            }
        }
    }
    if (r >= 0) D.fin[r] = f;
    const unsigned long long m = __ballot(stillActive);
    if (m) {
        unsigned base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&D.counters[2], (unsigned)__builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (stillActive) D.activeOut[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)r;
    }
}

// ---------------------------------------------------------------------------------------------- kernel 3: policy after genMatchString
// applyClearzone3 (AbstractMapThread.java:1820-1870)
__device__ bool apply_clearzone3(FinalRead &f, Site *s, int n, int L, int CZ3, float INV_CZ3) {
    if (!f.mapped || f.ambiguous || n < 2) return false;
    const float mults[7] = {0.f, 1.f, .75f, 0.5f, 0.25f, 0.125f, 0.0625f};
    const int score1 = s[0].slowScore;
    float sub = 0;
    const int mx = imin(7, n);
    for (int i = 1; i < mx; i++) {
        if (i > 2 && s[i].slowScore < s[i - 1].slowScore) break;
        const int dif = score1 - s[i].slowScore;
        float fr;
        if (dif >= CZ3) fr = 0;
        else { const float g = __fmul_rn((float)(CZ3 - dif), INV_CZ3), g2 = __fmul_rn(g, g); fr = __fadd_rn(__fadd_rn(g, __fmul_rn(2.f, g2)), __fmul_rn(__fmul_rn(2.f, g2), g)); }
        if (fr <= 0) break;
        sub = __fadd_rn(sub, __fmul_rn(fr, mults[i]));
    }
    if (sub <= 0) return false;
    const float asymptote = __fadd_rn(4.f, __fmul_rn(0.03f, (float)L));
    sub = __fmul_rn(sub, 1.8f);
    const float sub2 = __fmul_rn((float)CZ3, __fdiv_rn(__fmul_rn(asymptote, sub), __fadd_rn(sub, asymptote)));
    int subi = (int)__fadd_rn(sub2, 0.5f);
    if (subi >= f.mapScore - 300) subi = f.mapScore - 300;
    if (subi <= 0) return false;
    for (int i = 0; i < n; i++) { set_slow_score(s[i], s[i].slowScore - subi); s[i].score -= subi; }
    f.mapScore -= subi;
    return true;
}
// calcTipScorePenalty(r, maxScore, tiplen) (:2499-2573)
__device__ int calc_tip_score_penalty(const Dev &D, const FinalRead &f, const uint8_t *bases, int L, int maxScore, int tiplen) {
    if (!f.mapped || f.match == 0 || L < 2 * tiplen) return 0;
    const uint8_t *match = pool_ptr(D, f.match); const int n = f.matchLen, last = L - 1;
    int points = 0;
    uint8_t prev = 'm';
    for (int i = 0, cpos = 0; cpos <= tiplen && i < n; i++) {
        const uint8_t b = match[i];
        if (b == 'm') cpos++;
        else if (b == 'D') { if (prev != 'D') points += 2 * (tiplen + 2 - cpos); }
        else if (b == 'N' || b == 'C') { points += (tiplen + 2 - cpos); cpos++; }
        else { points += 2 * (tiplen + 2 - cpos); cpos++; }
        prev = b;
    }
    prev = 'm';
    for (int i = n - 1, cpos = 0; cpos <= tiplen && i >= 0; i--) {
        const uint8_t b = match[i];
        if (b == 'm') cpos++;
        else if (b == 'D') { if (prev != 'D') points += 2 * (tiplen + 2 - cpos); }
        else if (b == 'N' || b == 'C') { points += (tiplen + 2 - cpos); cpos++; }
        else { points += 2 * (tiplen + 2 - cpos); cpos++; }
        prev = b;
    }
    uint8_t b = bases[0];
    if (b != 'N' && b == bases[1]) for (int i = 2; i <= tiplen && bases[i] == b; i++) points++;
    b = bases[last];
    if (b != 'N' && b == bases[last - 1]) for (int i = last - 2; i >= (last - tiplen) && bases[i] == b; i--) points++;
    if (points < 1) return 0;
    const float fr = __fdiv_rn(__fmul_rn(80.f, (float)points), __fadd_rn((float)points, 80.f));
    const int penalty = (int)__fmul_rn(__fmul_rn(fr, .0022f), (float)maxScore);
    const int maxPenalty = f.mapScore - maxScore / 10;
    if (maxPenalty <= 0) return 0;
    return imin(penalty, maxPenalty);
}
__device__ inline bool contains_xyc(const Dev &D, const FinalRead &f) {          // Read.containsXYC (:2051-2056)
    if (f.match == 0 || f.matchLen < 1) return false;
    const uint8_t *m = pool_ptr(D, f.match);
    return m[0] == 'X' || m[f.matchLen - 1] == 'Y' || m[0] == 'C' || m[f.matchLen - 1] == 'C';
}

// MSA.toLocalAlignment(r, ss, basesM, minToClip, 1f) (MSA.java:216-484).  `room`: pool space reserved for this read by the caller
// (its strings only get as long as the old one plus the read).  Returns false when the mapping was cleared.
__device__ void to_local_alignment(const Dev &D, FinalRead &f, Site *s, int &n, const uint8_t *bp, const uint8_t *bm, int L, int minToClip, int room) {
    const Settings &S = D.S;
    for (int depth = 0; depth < 3; depth++) {                                   // the self-call after a semiperfect regeneration (:477-481)
        Site &ss = s[0];
        const uint8_t *bases = f.strand == 0 ? bp : bm;
        if (f.match == 0 || f.matchLen < 1) return;
        uint8_t *match = pool_ptr(D, f.match); const int mn = f.matchLen;
        if (match[0] == 'X' || match[mn - 1] == 'Y') { fix_xy(D, ss, bases, L); f.start = ss.start; f.stop = ss.stop; }
        int maxScore = -1, startLocC = -1, stopLocC = -1, lastZeroC = 0, startLocM = -1, stopLocM = -1, lastZeroM = 0, startLocR = -1, stopLocR = -1, lastZeroR = 0;
        uint8_t mode = match[0], prevMode = '0';
        int current = 0, prevStreak = 0, cpos = 0, rpos = f.start, score = 0;
        for (int mpos = 0; mpos <= mn; mpos++) {
            const bool atEnd = mpos == mn;
            if (!atEnd && mode == match[mpos]) { current++; continue; }
            if (atEnd && current <= 0) break;
            if (mode == 'm') {
                if (score <= 0) { score = 0; lastZeroC = cpos; lastZeroM = mpos - current; lastZeroR = rpos; }
                score += S.ptsMatch + (current - 1) * S.ptsMatch2;
                cpos += current; rpos += current;
                if (score > maxScore) { maxScore = score; startLocC = lastZeroC; startLocM = lastZeroM; startLocR = lastZeroR; stopLocC = cpos - 1; stopLocM = mpos - 1; stopLocR = rpos - 1; }
            } else if (mode == 'S') {
                score += f_sub_score(S, current);
                if (prevMode == 'N' || prevMode == 'R') score += S.ptsSub2 - S.ptsSub;
                else if (prevMode == 'm' && prevStreak < 2) score += -20;
                cpos += current; rpos += current;
            } else if (mode == 'D') { score += f_del_score(current); rpos += current; }
            else if (mode == 'I') { score += f_ins_score(current); cpos += current; }
            else if (mode == 'C') { cpos += current; rpos += current; }
            else if (mode == 'X' || mode == 'Y') { score += f_ins_score(current); cpos += current; rpos += current; }
            else if (mode == 'N' || mode == 'R') { cpos += current; rpos += current; }
            if (atEnd) break;
            prevMode = mode; prevStreak = current; mode = match[mpos]; current = 1;
        }
        if (startLocC < 0 || stopLocC < 0) { r_clear_mapping(f, n); f.pairedLost = 1; return; }
        int headTrimR = startLocC, headTrimM = startLocM, tailTrimR = L - stopLocC - 1, tailTrimM = mn - stopLocM - 1;
        if (headTrimR <= minToClip && headTrimM <= minToClip) headTrimR = headTrimM = 0;
        if (tailTrimR <= minToClip && tailTrimM <= minToClip) tailTrimR = tailTrimM = 0;
        if (headTrimR == 0 && headTrimM == 0 && tailTrimR == 0 && tailTrimM == 0) return;
        if (headTrimR - headTrimM == 0 && tailTrimR - tailTrimM == 0) {
            for (int i = 0; i < headTrimM; i++) match[i] = 'C';
            for (int i = mn - tailTrimM; i < mn; i++) match[i] = 'C';
        } else {
            const int newlen = mn - headTrimM - tailTrimM + headTrimR + tailTrimR;
            const int ref2 = room >= newlen ? pool_alloc(D, newlen) : 0;
            if (!ref2) return;                                                  // cannot happen: the caller reserved mn + 4 L bytes
            room -= newlen;
            uint8_t *match2 = pool_ptr(D, ref2);
            for (int i = 0; i < headTrimR; i++) match2[i] = 'C';
            for (int i = newlen - tailTrimR; i < newlen; i++) match2[i] = 'C';
            for (int i = headTrimM, i2 = headTrimR, lim = newlen - tailTrimR; i2 < lim; i++, i2++) match2[i2] = match[i];
            f.match = ref2; f.matchLen = newlen;
        }
        if (headTrimR != 0) f.start = startLocR - headTrimR;
        if (tailTrimR != 0) f.stop = stopLocR + tailTrimR;
        maxScore = imax(maxScore, ss.slowScore);
        f.mapScore = maxScore;
        site_set_match(ss, f.match, f.matchLen);
        set_limits(ss, f.start, f.stop);
        // SiteScore.isPerfect / isSemiPerfect (SiteScore.java:175-223)
        bool isPerfect = false, isSemi = false;
        if (L == ss.stop - ss.start + 1) {
            const uint8_t *ref = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom];
            if (ss.start >= 0 && ss.stop < reflen) { isPerfect = true; for (int i = 0; i < L; i++) { const uint8_t c = bases[i], rr = ref[ss.start + i]; if (c != rr || c == 'N') { isPerfect = false; break; } } }
            int readStart = 0, readStop = L, maxNoref = L / 2; const int refStop = ss.start + L;
            if (ss.start < 0) readStart = -ss.start;
            if (refStop > reflen) readStop -= (refStop - reflen);
            isSemi = true;
            for (int i = readStart; i < readStop; i++) {
                const uint8_t c = bases[i], rr = ref[ss.start + i];
                if (c == 'N') { isSemi = false; break; }
                if (c != rr) { maxNoref--; if (maxNoref < 0 || rr != 'N') { isSemi = false; break; } }
            }
        }
        if (!ss.perfect && isPerfect) {
            ss.perfect = ss.semiperfect = 1; f.perfect = 1;
            uint8_t *m2 = pool_ptr(D, f.match);
            for (int i = 0; i < f.matchLen; i++) m2[i] = 'm';
            set_slow_score(ss, maxScore);
            return;
        }
        if (!ss.semiperfect && isSemi) {
            ss.semiperfect = 1;
            const int ref3 = room >= L ? pool_alloc(D, L) : 0;
            if (!ref3) return;
            room -= L;
            uint8_t *m3 = pool_ptr(D, ref3);
            const uint8_t *ref = D.chromArr[ss.chrom]; const int reflen = D.chromArrLen[ss.chrom];
            for (int i = 0, j = ss.start; i < L; i++, j++) {                    // MSA.genMatchNoIndels (MultiStateAligner11tsJNI.java:1092-1108)
                const uint8_t c = bases[i], rr = (j < 0 || j >= reflen) ? (uint8_t)'N' : ref[j];
                m3[i] = (c == 'N' || rr == 'N') ? 'N' : (c == rr ? 'm' : 'S');
            }
            f.match = ref3; f.matchLen = L; site_set_match(ss, ref3, L);
            continue;                                                           // `return toLocalAlignment(r, ss, ...)`
        }
        return;
    }
}

// what is left of processRead / processReadPair once genMatchString is through, up to (not including) toLocalAlignment, which needs
// pool space and therefore runs as a pass of its own over the reads flagged here (FinalRead.needLocal)
__global__ __launch_bounds__(128) void final_end_kernel(const Dev D) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const Settings &S = D.S;
    if (S.paired) {
        if (2 * u + 1 >= D.nreads) return;
        const long long r1 = 2 * u, r2 = r1 + 1;
        int n1 = D.mcount[r1], n2 = D.mcount[r2];
        if (n1 < 0 || n2 < 0) return;
        FinalRead f1 = D.fin[r1], f2 = D.fin[r2];
        Site *s1 = D.ms + r1 * D.cap, *s2 = D.ms + r2 * D.cap;
        if (f1.pairedLost || f2.pairedLost) { f1.paired = 0; f2.paired = 0; }                     // genMatchString :939-942 (either mate's)
        // :1262-1285 (`mapScore>0 && sites==null` cannot hold: an empty list has mapScore 0)
        if (f1.mapScore <= 0 && n1 > 0) { r_clear_mapping(f1, n1); f2.paired = 0; }
        if (f2.mapScore <= 0 && n2 > 0) { r_clear_mapping(f2, n2); f1.paired = 0; }
        remove_duplicate_best_sites(s1, n1); remove_duplicate_best_sites(s2, n2);
        f1.needLocal = f1.mapped && contains_xyc(D, f1); f2.needLocal = f2.mapped && contains_xyc(D, f2);
        if (f1.needLocal) { site_set_match(s1[0], f1.match, f1.matchLen); atomicAdd(&D.counters[24], 1u); atomicAdd(&D.counters[25], (unsigned)((f1.matchLen + 4 * D.reads[r1].len + 64) >> 2)); }
        if (f2.needLocal) { site_set_match(s2[0], f2.match, f2.matchLen); atomicAdd(&D.counters[24], 1u); atomicAdd(&D.counters[25], (unsigned)((f2.matchLen + 4 * D.reads[r2].len + 64) >> 2)); }
        D.mcount[r1] = n1; D.mcount[r2] = n2; D.fin[r1] = f1; D.fin[r2] = f2;
    } else {
        if (u >= D.nreads) return;
        int n = D.mcount[u];
        if (n < 0) return;
        FinalRead f = D.fin[u];
        Site *s = D.ms + u * D.cap;
        const int L = D.reads[u].len, maxSw = max_quality(S, L);
        if (n > 1) remove_duplicate_best_sites(s, n);                           // BBMapThread.java:625-630
        if (n > 0) site_set_match(s[0], f.match, f.matchLen);
        if (n > 0 && f.mapScore <= 0) { f.mapScore = 0; f.mapped = 0; n = 0; }                   // :633-641
        if (f.mapScore <= 0 && n > 0) r_clear_mapping(f, n);                                     // :647-657
        if (n > 0 && !f.ambiguous && S.clearzone3 > 0) {                                         // :668-682
            const float q = __fdiv_rn((float)maxSw, (float)f.mapScore);
            const float cz3v2 = __fmul_rn((float)S.clearzone3, 1.25f < q ? 1.25f : q);
            if (apply_clearzone3(f, s, n, L, (int)cz3v2, __fdiv_rn(1.f, cz3v2))) {
                if (f.mapScore < (int)__fmul_rn((float)maxSw, S.minRatio)) f.ambiguous = 1;
            }
        }
        f.needLocal = f.mapped && contains_xyc(D, f);
        if (f.needLocal) { atomicAdd(&D.counters[24], 1u); atomicAdd(&D.counters[25], (unsigned)((f.matchLen + 4 * L + 64) >> 2)); }
        D.mcount[u] = n; D.fin[u] = f;
    }
}

// toLocalAlignment for the flagged reads and the tail behind it (BBMapThread.java:692-709 / :1334-1349), then the output records
__global__ __launch_bounds__(128) void final_local_kernel(const Dev D) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const Settings &S = D.S;
    const int nper = S.paired ? 2 : 1;
    if (nper * u + (nper - 1) >= D.nreads) return;
    if (S.paired && (D.mcount[2 * u] < 0 || D.mcount[2 * u + 1] < 0)) {
        for (int w = 0; w < 2; w++) { bbmap_final o; memset(&o, 0, sizeof o); o.chrom = -1; o.start = o.stop = -1; o.nsites = D.mcount[2 * u + w]; D.finalOut[2 * u + w] = o; }
        return;
    }
    if (!S.paired && D.mcount[u] < 0) { bbmap_final o; memset(&o, 0, sizeof o); o.chrom = -1; o.start = o.stop = -1; o.nsites = D.mcount[u]; D.finalOut[u] = o; return; }
    FinalRead fs[2]; int ns[2];
    for (int w = 0; w < nper; w++) { fs[w] = D.fin[nper * u + w]; ns[w] = D.mcount[nper * u + w]; }
    for (int w = 0; w < nper; w++) {
        const long long r = nper * u + w;
        FinalRead &f = fs[w];
        if (f.needLocal && f.mapped) {
            const bbidx_read rr = D.reads[r];
            const int before = f.mapped;
            to_local_alignment(D, f, D.ms + r * D.cap, ns[w], D.bases + rr.bases_off, D.bases + rr.bases_off + D.minusDelta, rr.len, 1, f.matchLen + 4 * rr.len + 32);
            if (before && !f.mapped && nper == 2) fs[1 - w].paired = 0;         // clearMapping: mate.setPaired(false)
        }
    }
    for (int w = 0; w < nper; w++) {
        const long long r = nper * u + w;
        FinalRead &f = fs[w];
        int n = ns[w];
        Site *s = D.ms + r * D.cap;
        const bbidx_read rr = D.reads[r];
        const int L = rr.len, maxSw = max_quality(S, L);
        if (!S.paired) {
            if (n == 0 || (!f.ambiguous && (float)f.mapScore < __fmul_rn((float)maxSw, S.minRatio))) r_clear_mapping(f, n);      // :697-699
            if (S.clearzone3 > 0) {                                             // PENALIZE_AMBIG (:706-709)
                const int penalty = calc_tip_score_penalty(D, f, D.bases + rr.bases_off, L, maxSw, 7);
                if (penalty > 0) { f.mapScore -= penalty; for (int i = 0; i < n; i++) { set_slow_score(s[i], s[i].slowScore - penalty); s[i].score -= penalty; } }
            }
        }
        D.mcount[r] = n;
        bbmap_final o; memset(&o, 0, sizeof o);
        o.mapped = f.mapped; o.chrom = f.chrom; o.strand = f.strand; o.start = f.start; o.stop = f.stop; o.mapScore = f.mapScore;
        o.paired = f.paired; o.ambiguous = f.ambiguous; o.perfect = f.perfect; o.rescued = f.rescued;
        o.match_len = f.match ? f.matchLen : 0; o.match_off = f.match ? 4ll * (f.match - 1) : 0; o.nsites = n;
        D.finalOut[r] = o;
        D.fin[r] = f;
    }
}
