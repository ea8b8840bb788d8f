// libbbtoolsjni.so: the native library BBTools' unmodified Java classes load with `usejni=t`
// (System.loadLibrary("bbtoolsjni"), current/align2/MultiStateAligner11tsJNI.java:11-38), implemented on libbbmap_amd.so.
// All nine symbols of the reference's library (jni/makefile.linux builds them from three C files):
//   Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI / fillLimitedXJNI
//        replace jni/MultiStateAligner11tsJNI.c:707-812 (header jni/align2_MultiStateAligner11tsJNI.h:164-174)
//   Java_align2_BandedAlignerJNI_align{Forward,ForwardRC,Reverse,ReverseRC}JNI
//        replace jni/BandedAlignerJNI.c:588-757 (header jni/align2_BandedAlignerJNI.h:17-41)
//   Java_jgi_BBMergeOverlapper_mateByOverlapJNI / mateByOverlapRatioJNI / mateByOverlapRatioJNI_1WithQualities
//        replace jni/BBMergeOverlapper.c:389-520 (header jni/jgi_BBMergeOverlapper.h:21-43); host code, see bbmerge_overlap.cpp
//
// Two layers.  The bbjni_* functions (extern "C", plain pointers) hold everything that is not JNI marshalling and are tested
// through ctypes and through the mock JNIEnv of jni/mock_jni_test.cpp.  The Java_* functions only move data across the JNI
// boundary, and they never wait for the GPU while a GetPrimitiveArrayCritical region is open (a critical region blocks the
// garbage collector for every other mapping thread): inputs are copied out with Get<Type>ArrayRegion (of the reference
// only the window), the fill runs into the library's pinned staging area (bbmsa_fill_submit: calls of all mapping threads that
// arrive together share one launch), and one short critical region copies the fill's rectangle into `packed` (bbmsa_fill_collect,
// a memcpy).
//
// Compiled against <jni.h> when a JDK is installed, otherwise against jni/jni_min.h (this image has no JDK).
#if __has_include(<jni.h>)
#include <jni.h>
#else
#include "jni_min.h"
#endif

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "bbmap_amd.h"
#include "bbmerge_overlap.h"

namespace {

struct MsaSlot { int maxRows, maxColumns, bandwidth; float ratio; bbmsa_ctx *ctx; };
struct BandSlot { int width; bbband_ctx *ctx; };
// MSA contexts are process-wide, one per (maxRows, maxColumns, band): every mapping thread's MSA object of that shape shares it, which
// is what lets the library combine their calls.  BandedAligner contexts stay one per thread, like the objects in BBMap.
std::mutex g_msaMu;
std::vector<MsaSlot> g_msa;
thread_local std::vector<BandSlot> t_band;
thread_local char t_err[320] = "";

bbmsa_ctx *msa_ctx(int maxRows, int maxColumns, int bandwidth, float ratio) {
    std::lock_guard<std::mutex> g(g_msaMu);
    for (const MsaSlot &s : g_msa)
        if (s.maxRows == maxRows && s.maxColumns == maxColumns && s.bandwidth == bandwidth && s.ratio == ratio) return s.ctx;
    bbmsa_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.maxRows = maxRows; cfg.maxColumns = maxColumns; cfg.bandwidth = bandwidth; cfg.bandwidthRatio = ratio;
    cfg.reserved[2] = BBMSA_SCHEME_11TS | BBMSA_LEGACY_ONLY;      // per-call fills only: staging for two batches of calls, no batch buffers
    bbmsa_ctx *ctx = nullptr;
    if (bbmsa_create(&cfg, &ctx) != BBMAP_OK) return nullptr;
    g_msa.push_back(MsaSlot{maxRows, maxColumns, bandwidth, ratio, ctx});
    return ctx;
}
bbband_ctx *band_ctx(int maxWidth) {
    for (const BandSlot &s : t_band) if (s.width == maxWidth) return s.ctx;
    bbband_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.width = maxWidth; cfg.semantics = BBBAND_SEMANTICS_JNI_C;      // the C file's semantics, as this symbol had
    bbband_ctx *ctx = nullptr;
    if (bbband_create(&cfg, &ctx) != BBMAP_OK) return nullptr;
    t_band.push_back(BandSlot{maxWidth, ctx});
    return ctx;
}

}  // namespace

extern "C" {

// jni/MultiStateAligner11tsJNI.c fillUnlimited (:100-117) / fillLimitedX (:361-382) with their own plain signature, minus the
// score tables (the kernels carry them), in two halves.  bbjni_fill_submit runs the fill (blocking; calls from other threads that
// arrive meanwhile share the launch) and returns result (4 ints unlimited, 5 limited), increments *iterations and leaves a ticket;
// bbjni_fill_collect copies the fill's rectangle into `packed` (the Java class's matrix, 3 x (maxRows+1) x (maxColumns+1) ints; rows
// 1..readLen, columns 1..columns only) and, limited fill, vertLimit[0..readLen] / horizLimit[0..columns] (:413-438).  0 = ok.
int bbjni_fill_submit(int limited, const uint8_t *read, int readLen, const uint8_t *ref, int refLen, int refStartLoc, int refEndLoc,
                      int minScore, int32_t *result, int64_t *iterations, int maxRows, int maxColumns, int bandwidth,
                      float bandwidthRatio, bbmsa_ctx **ctxOut, bbmsa_ticket *ticket) {
    bbmsa_ctx *ctx = msa_ctx(maxRows, maxColumns, limited ? bandwidth : 0, limited ? bandwidthRatio : 0.0f);
    if (!ctx) return BBMAP_E_HIP;
    int32_t r5[5] = {0, 0, 0, 0, 0};
    const int rc = bbmsa_fill_submit(ctx, read, readLen, ref, refLen, refStartLoc, refEndLoc, minScore,
                                     limited ? BBMSA_FILL_LIMITED_RAW : BBMSA_FILL_UNLIMITED_RAW, r5, iterations, ticket);
    if (rc != BBMAP_OK) return rc;
    for (int i = 0; i < (limited ? 5 : 4); i++) result[i] = r5[i];
    *ctxOut = ctx;
    return BBMAP_OK;
}
int bbjni_fill_collect(bbmsa_ctx *ctx, bbmsa_ticket *ticket, int32_t *packed, int32_t *vertLimit, int32_t *horizLimit) {
    return bbmsa_fill_collect(ctx, ticket, packed, vertLimit, horizLimit);
}
// both halves, for callers that own `packed` (ctypes, tests); vertLimit / horizLimit may be NULL
int bbjni_fill(int limited, const uint8_t *read, int readLen, const uint8_t *ref, int refLen, int refStartLoc, int refEndLoc,
               int minScore, int32_t *result, int64_t *iterations, int32_t *packed, int maxRows, int maxColumns,
               int bandwidth, float bandwidthRatio) {
    bbmsa_ctx *ctx = nullptr;
    bbmsa_ticket t;
    const int rc = bbjni_fill_submit(limited, read, readLen, ref, refLen, refStartLoc, refEndLoc, minScore, result, iterations,
                                     maxRows, maxColumns, bandwidth, bandwidthRatio, &ctx, &t);
    if (rc != BBMAP_OK) return rc;
    return bbmsa_fill_collect(ctx, &t, packed, nullptr, nullptr);
}

// jni/BandedAlignerJNI.c alignForward / alignForwardRC / alignReverse / alignReverseRC (:123-585): direction = BBBAND_*.
int bbjni_banded(int direction, const uint8_t *query, int qLen, const uint8_t *ref, int rLen, int qstart, int rstart, int maxEdits,
                 int exact, int maxWidth, int32_t *returnVals5, int32_t *edits) {
    bbband_ctx *ctx = band_ctx(maxWidth);
    if (!ctx) return BBMAP_E_HIP;
    std::vector<uint8_t> seqs((size_t)qLen + (size_t)rLen + 1);
    memcpy(seqs.data(), query, (size_t)qLen);
    memcpy(seqs.data() + qLen, ref, (size_t)rLen);
    bbband_job job;
    job.query_off = 0; job.ref_off = qLen; job.query_len = qLen; job.ref_len = rLen;
    job.qstart = qstart; job.rstart = rstart; job.maxEdits = maxEdits;
    job.flags = direction | (exact ? BBBAND_EXACT : 0);
    bbband_result res;
    const int rc = bbband_align_batch(ctx, 1, &job, seqs.data(), (int64_t)qLen + rLen, &res);
    if (rc != BBMAP_OK) return rc;
    returnVals5[0] = res.lastQueryLoc; returnVals5[1] = res.lastRefLoc; returnVals5[2] = res.lastRow;
    returnVals5[3] = res.lastEdits; returnVals5[4] = res.lastOffset;
    *edits = res.edits;
    return BBMAP_OK;
}

// frees the calling thread's contexts (a mapping thread that ends; tests)
void bbjni_release_thread(void) {
    for (BandSlot &s : t_band) bbband_destroy(s.ctx);
    t_band.clear();
}
// bbmsa_legacy_stats (calls, launches, hand-overs, three leader times in ns), summed over the process-wide MSA contexts
void bbjni_legacy_stats(int64_t *stats6) {
    std::lock_guard<std::mutex> g(g_msaMu);
    for (int i = 0; i < 6; i++) stats6[i] = 0;
    for (MsaSlot &s : g_msa) {
        int64_t st[6] = {0, 0, 0, 0, 0, 0};
        if (bbmsa_legacy_stats(s.ctx, st) == BBMAP_OK) for (int i = 0; i < 6; i++) stats6[i] += st[i];
    }
}
// frees the process-wide MSA contexts; no fill may be in flight (tests, library unload)
void bbjni_release_all(void) {
    std::lock_guard<std::mutex> g(g_msaMu);
    for (MsaSlot &s : g_msa) bbmsa_destroy(s.ctx);
    g_msa.clear();
}

}  // extern "C"

namespace {

void throw_runtime(JNIEnv *env, const char *what) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, what);
}

// shared by the two fills; `limited` selects fillLimitedX
void fill_common(JNIEnv *env, bool limited, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jint minScore,
                 jintArray result, jlongArray iterations, jintArray packed, jint maxRows, jint maxColumns, jint bandwidth, jfloat ratio,
                 jintArray vertLimit, jintArray horizLimit) {
    const jsize readLen = env->GetArrayLength(read), refLen = env->GetArrayLength(ref);
    // `ref` is the whole chromosome array; the fill reads ref[refStartLoc + col - 1] for col 1..columns only
    // (jni/MultiStateAligner11tsJNI.c:137-139, :427-438), so only that window leaves the JVM: the job is rebased to offset 0.
    // result[] holds row / column / state indices, never absolute coordinates, so nothing else changes.
    if (refStartLoc < 0 || refEndLoc >= refLen || refEndLoc < refStartLoc) {
        snprintf(t_err, sizeof t_err, "bbtoolsjni: window [%d, %d] outside ref[%d]", (int)refStartLoc, (int)refEndLoc, (int)refLen);
        throw_runtime(env, t_err);
        return;
    }
    const jsize cols = refEndLoc - refStartLoc + 1;
    std::vector<jbyte> rd((size_t)readLen + 1), rf((size_t)cols + 1);
    env->GetByteArrayRegion(read, 0, readLen, rd.data());
    env->GetByteArrayRegion(ref, refStartLoc, cols, rf.data());
    jlong it = 0;
    env->GetLongArrayRegion(iterations, 0, 1, &it);
    int32_t r5[5] = {0, 0, 0, 0, 0};
    int64_t it64 = it;
    bbmsa_ctx *ctx = nullptr;
    bbmsa_ticket ticket;
    const int rc = bbjni_fill_submit(limited ? 1 : 0, (const uint8_t *)rd.data(), readLen, (const uint8_t *)rf.data(), cols, 0, cols - 1,
                                     minScore, r5, &it64, maxRows, maxColumns, bandwidth, ratio, &ctx, &ticket);
    if (rc != BBMAP_OK) {                                          // the reference calls exit(0) here (jni/...c:130-132)
        snprintf(t_err, sizeof t_err, "bbtoolsjni: fill failed (%d): %s", rc, bbmap_last_error());
        throw_runtime(env, t_err);
        return;
    }
    // one short critical region: the fill's rectangle (rows 1..rows, columns 1..columns of the three planes -- what score2 /
    // traceback2 read besides the constructor's row 0 and column 0, current/align2/MultiStateAligner11tsJNI.java:376-658) goes from the
    // pinned staging area into the Java matrix; nothing in here blocks or touches the GPU
    std::vector<int32_t> vl, hl;
    const bool wantLimits = limited && vertLimit && horizLimit && env->GetArrayLength(vertLimit) > readLen &&
                            env->GetArrayLength(horizLimit) > cols;
    if (wantLimits) { vl.resize((size_t)readLen + 1); hl.resize((size_t)cols + 1); }
    jint *jp = (jint *)env->GetPrimitiveArrayCritical(packed, nullptr);
    bbjni_fill_collect(ctx, &ticket, (int32_t *)jp, wantLimits ? vl.data() : nullptr, wantLimits ? hl.data() : nullptr);
    if (jp) env->ReleasePrimitiveArrayCritical(packed, jp, 0);
    if (wantLimits) {      // the native code fills these as a side effect (:413-438); Java only prints them, but they are its arrays
        env->SetIntArrayRegion(vertLimit, 0, readLen + 1, (const jint *)vl.data());
        env->SetIntArrayRegion(horizLimit, 0, cols + 1, (const jint *)hl.data());
    }
    env->SetIntArrayRegion(result, 0, limited ? 5 : 4, (const jint *)r5);
    it = it64;
    env->SetLongArrayRegion(iterations, 0, 1, &it);                // incremented, not set (jni/...c:471)
}

jint band_common(JNIEnv *env, int direction, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits,
                 jboolean exact, jint maxWidth, jintArray returnVals) {
    const jsize qLen = env->GetArrayLength(query), rLen = env->GetArrayLength(ref);
    std::vector<jbyte> q((size_t)qLen + 1), r((size_t)rLen + 1);
    env->GetByteArrayRegion(query, 0, qLen, q.data());
    env->GetByteArrayRegion(ref, 0, rLen, r.data());
    int32_t vals[5] = {0, 0, 0, 0, 0}, edits = 0;
    const int rc = bbjni_banded(direction, (const uint8_t *)q.data(), qLen, (const uint8_t *)r.data(), rLen, qstart, rstart, maxEdits,
                                exact ? 1 : 0, maxWidth, vals, &edits);
    if (rc != BBMAP_OK) {
        snprintf(t_err, sizeof t_err, "bbtoolsjni: banded alignment failed (%d): %s", rc, bbmap_last_error());
        throw_runtime(env, t_err);
        return 0;
    }
    env->SetIntArrayRegion(returnVals, 0, 5, (const jint *)vals);   // fully rewritten, jni/BandedAlignerJNI.c:604-630
    return edits;
}

// BBMerge natives: arrays are small (two reads); copied in, probabilities and rvector copied back
struct MergeArrays {
    std::vector<jbyte> a, b, aq, bq; std::vector<jfloat> ap, bp; jint rv[5];
    jsize alen, blen;
};
void merge_in(JNIEnv *env, MergeArrays &M, jbyteArray a, jbyteArray b, jbyteArray aq, jbyteArray bq, jfloatArray ap, jfloatArray bp, jintArray rv) {
    M.alen = env->GetArrayLength(a); M.blen = env->GetArrayLength(b);
    M.a.resize((size_t)M.alen + 1); M.b.resize((size_t)M.blen + 1);
    env->GetByteArrayRegion(a, 0, M.alen, M.a.data()); env->GetByteArrayRegion(b, 0, M.blen, M.b.data());
    if (aq) { M.aq.resize((size_t)M.alen + 1); env->GetByteArrayRegion(aq, 0, M.alen, M.aq.data()); }
    if (bq) { M.bq.resize((size_t)M.blen + 1); env->GetByteArrayRegion(bq, 0, M.blen, M.bq.data()); }
    // the probability arrays are the caller's scratch: mateByOverlap reads bprob[j] with j running over read a (jni/BBMergeOverlapper.c:70),
    // i.e. up to max(alen, blen) entries of whatever the Java array holds; copy the whole array (zero-padded to that length)
    const jsize need = (M.alen > M.blen ? M.alen : M.blen) + 1;
    if (ap) { const jsize n = env->GetArrayLength(ap); M.ap.assign((size_t)(n > need ? n : need), 0.0f); env->GetFloatArrayRegion(ap, 0, n, M.ap.data()); }
    if (bp) { const jsize n = env->GetArrayLength(bp); M.bp.assign((size_t)(n > need ? n : need), 0.0f); env->GetFloatArrayRegion(bp, 0, n, M.bp.data()); }
    env->GetIntArrayRegion(rv, 0, 5, M.rv);
}

}  // namespace

extern "C" {

// ([B[BII[I[J[I[I[III)V
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI(
    JNIEnv *env, jobject, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jintArray result,
    jlongArray iterationsUnlimited, jintArray packed, jintArray /*POINTSoff_SUB_ARRAY*/, jintArray /*POINTSoff_INS_ARRAY*/,
    jint maxRows, jint maxColumns) {
    fill_common(env, false, read, ref, refStartLoc, refEndLoc, 0, result, iterationsUnlimited, packed, maxRows, maxColumns, 0, 0.0f,
                nullptr, nullptr);
}

// ([B[BIII[I[J[I[I[IIIIF[I[I[B[I)V -- MSA.bandwidth / bandwidthRatio arrive per call and select the context
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(
    JNIEnv *env, jobject, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jint minScore, jintArray result,
    jlongArray iterationsLimited, jintArray packed, jintArray, jintArray, jint maxRows, jint maxColumns, jint bandwidth,
    jfloat bandwidthRatio, jintArray vertLimit, jintArray horizLimit, jbyteArray /*baseToNumber*/, jintArray /*INS_ARRAY_C*/) {
    fill_common(env, true, read, ref, refStartLoc, refEndLoc, minScore, result, iterationsLimited, packed, maxRows, maxColumns,
                bandwidth, bandwidthRatio, vertLimit, horizLimit);
}

JNIEXPORT jint JNICALL Java_align2_BandedAlignerJNI_alignForwardJNI(
    JNIEnv *env, jobject, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits, jboolean exact, jint maxWidth,
    jbyteArray /*baseToNumber*/, jintArray returnVals) {
    return band_common(env, BBBAND_FORWARD, query, ref, qstart, rstart, maxEdits, exact, maxWidth, returnVals);
}
JNIEXPORT jint JNICALL Java_align2_BandedAlignerJNI_alignForwardRCJNI(
    JNIEnv *env, jobject, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits, jboolean exact, jint maxWidth,
    jbyteArray /*baseToNumber*/, jbyteArray /*baseToComplementExtended*/, jintArray returnVals) {
    return band_common(env, BBBAND_FORWARD_RC, query, ref, qstart, rstart, maxEdits, exact, maxWidth, returnVals);
}
JNIEXPORT jint JNICALL Java_align2_BandedAlignerJNI_alignReverseJNI(
    JNIEnv *env, jobject, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits, jboolean exact, jint maxWidth,
    jbyteArray /*baseToNumber*/, jintArray returnVals) {
    return band_common(env, BBBAND_REVERSE, query, ref, qstart, rstart, maxEdits, exact, maxWidth, returnVals);
}
JNIEXPORT jint JNICALL Java_align2_BandedAlignerJNI_alignReverseRCJNI(
    JNIEnv *env, jobject, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits, jboolean exact, jint maxWidth,
    jbyteArray /*baseToNumber*/, jbyteArray /*baseToComplementExtended*/, jintArray returnVals) {
    return band_common(env, BBBAND_REVERSE_RC, query, ref, qstart, rstart, maxEdits, exact, maxWidth, returnVals);
}

// ([B[B[B[B[F[F[IIIIIIII)I
JNIEXPORT jint JNICALL Java_jgi_BBMergeOverlapper_mateByOverlapJNI(
    JNIEnv *env, jclass, jbyteArray a_bases, jbyteArray b_bases, jbyteArray a_quality, jbyteArray b_quality, jfloatArray aprob,
    jfloatArray bprob, jintArray rvector, jint minOverlap0, jint minOverlap, jint minInsert0, jint margin, jint maxMismatches0,
    jint maxMismatches, jint minq) {
    MergeArrays M;
    merge_in(env, M, a_bases, b_bases, a_quality, b_quality, aprob, bprob, rvector);
    const jint r = bbmerge_mate_by_overlap(M.a.data(), M.alen, M.b.data(), M.blen, a_quality ? M.aq.data() : nullptr,
                                           b_quality ? M.bq.data() : nullptr, M.ap.data(), M.bp.data(), M.rv, minOverlap0, minOverlap,
                                           minInsert0, margin, maxMismatches0, maxMismatches, minq);
    env->SetIntArrayRegion(rvector, 0, 5, M.rv);
    // the reference writes the probabilities through its critical pointers (direct pointers in HotSpot, so JNI_ABORT discards
    // nothing): the Java scratch arrays keep them, and a later call can read them past its own read length
    if (aprob) env->SetFloatArrayRegion(aprob, 0, M.alen, M.ap.data());
    if (bprob) env->SetFloatArrayRegion(bprob, 0, M.blen, M.bp.data());
    return r;
}

// ([B[B[B[B[F[F[IIIIIFFF)I -- the Java method is mateByOverlapRatioJNI_WithQualities (current/jgi/BBMergeOverlapper.java:56), whose
// JNI name escapes the underscore as _1 (as jni/jgi_BBMergeOverlapper.h:32 declares it); the reference's C file defines
// ..._mateByOverlapJNI_WithQualities instead (jni/BBMergeOverlapper.c:389), which no Java method resolves to.  Both are exported.
JNIEXPORT jint JNICALL Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI_1WithQualities(
    JNIEnv *env, jclass, jbyteArray a_bases, jbyteArray b_bases, jbyteArray a_quality, jbyteArray b_quality, jfloatArray aprob,
    jfloatArray bprob, jintArray rvector, jint minOverlap0, jint minOverlap, jint minInsert0, jint minInsert, jfloat maxRatio,
    jfloat margin, jfloat offset) {
    MergeArrays M;
    merge_in(env, M, a_bases, b_bases, a_quality, b_quality, aprob, bprob, rvector);
    const jint r = bbmerge_mate_by_overlap_ratio_with_qualities(M.a.data(), M.alen, M.b.data(), M.blen, M.aq.data(), M.bq.data(),
                                                                M.ap.data(), M.bp.data(), M.rv, minOverlap0, minOverlap, minInsert0,
                                                                minInsert, maxRatio, margin, offset);
    env->SetIntArrayRegion(rvector, 0, 5, M.rv);
    if (aprob) env->SetFloatArrayRegion(aprob, 0, M.alen, M.ap.data());
    if (bprob) env->SetFloatArrayRegion(bprob, 0, M.blen, M.bp.data());
    return r;
}
JNIEXPORT jint JNICALL Java_jgi_BBMergeOverlapper_mateByOverlapJNI_WithQualities(
    JNIEnv *env, jclass c, jbyteArray a_bases, jbyteArray b_bases, jbyteArray a_quality, jbyteArray b_quality, jfloatArray aprob,
    jfloatArray bprob, jintArray rvector, jint minOverlap0, jint minOverlap, jint minInsert0, jint minInsert, jfloat maxRatio,
    jfloat margin, jfloat offset) {
    return Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI_1WithQualities(env, c, a_bases, b_bases, a_quality, b_quality, aprob, bprob,
                                                                          rvector, minOverlap0, minOverlap, minInsert0, minInsert, maxRatio,
                                                                          margin, offset);
}

// ([B[B[IIIIIFFFFF)I
JNIEXPORT jint JNICALL Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI(
    JNIEnv *env, jclass, jbyteArray a_bases, jbyteArray b_bases, jintArray rvector, jint minOverlap0, jint minOverlap, jint minInsert0,
    jint minInsert, jfloat maxRatio, jfloat margin, jfloat offset, jfloat gIncr, jfloat bIncr) {
    MergeArrays M;
    merge_in(env, M, a_bases, b_bases, nullptr, nullptr, nullptr, nullptr, rvector);
    const jint r = bbmerge_mate_by_overlap_ratio(M.a.data(), M.alen, M.b.data(), M.blen, M.rv, minOverlap0, minOverlap, minInsert0,
                                                 minInsert, maxRatio, margin, offset, gIncr, bIncr);
    env->SetIntArrayRegion(rvector, 0, 5, M.rv);
    return r;
}

}  // extern "C"
