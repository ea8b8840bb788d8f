// Drop-in JNI shim: the symbols BBMap's unmodified Java classes bind with `usejni=t`, implemented on libbbmap_amd.so.
//
//   Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI / fillLimitedXJNI
//        replace jni/MultiStateAligner11tsJNI.c:707-812 (header jni/align2_MultiStateAligner11tsJNI.h:164-174)
//   Java_align2_BandedAlignerJNI_align{Forward,ForwardRC,Reverse,ReverseRC}JNI
//        replace jni/BandedAlignerJNI.c:588-757 (header jni/align2_BandedAlignerJNI.h:17-41)
//
// NOT BUILT OR TESTED IN THIS REPOSITORY'S IMAGE: it has no JDK (no <jni.h>, no JVM), so this file is outside the default
// build (bbmap_amd/build.py compiles bbmap_amd/csrc only).  On a machine with a JDK:
//     g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/bbtoolsjni_shim.cpp \
//         -Lbbmap_amd -lbbmap_amd -Wl,-rpath,'$ORIGIN' -o bbmap_amd/libbbtoolsjni.so
// and run BBMap with -Djava.library.path=<repo>/bbmap_amd usejni=t.  Everything below is marshalling: the arithmetic is
// behind the C ABI (include/bbmap_amd.h), which IS tested (tests/test_msa_gpu.py::test_legacy_packed_matrix_feeds_the_java_walkers,
// tests/test_banded_gpu.py).  The three Java_jgi_BBMergeOverlapper_* symbols of the reference's library are not provided.
//
// These per-call entry points keep the reference's shape (one alignment per call, the whole `packed` matrix copied back,
// SURVEY.md R7); they exist so that nothing on the Java side has to change.  The fast path is the batched ABI
// (INTEGRATION.md sections 2 and 2b).
#if __has_include(<jni.h>)
#include <jni.h>

#include <mutex>

#include "bbmap_amd.h"

namespace {

struct MsaSlot { int maxRows, maxColumns; bbmsa_ctx *ctx; };
thread_local MsaSlot t_msa = {0, 0, nullptr};        // one context per mapping thread, like one MSA per thread in BBMap

bbmsa_ctx *msa_ctx(int maxRows, int maxColumns) {
    if (t_msa.ctx && t_msa.maxRows == maxRows && t_msa.maxColumns == maxColumns) return t_msa.ctx;
    if (t_msa.ctx) { bbmsa_destroy(t_msa.ctx); t_msa.ctx = nullptr; }
    bbmsa_config cfg = {};
    cfg.device = 0; cfg.maxRows = maxRows; cfg.maxColumns = maxColumns; cfg.bandwidth = 0; cfg.bandwidthRatio = 0.0f;
    if (bbmsa_create(&cfg, &t_msa.ctx) != BBMAP_OK) return nullptr;
    t_msa.maxRows = maxRows; t_msa.maxColumns = maxColumns;
    return t_msa.ctx;
}

struct BandSlot { int width; bbband_ctx *ctx; };
thread_local BandSlot t_band = {0, nullptr};

bbband_ctx *band_ctx(int maxWidth) {
    if (t_band.ctx && t_band.width == maxWidth) return t_band.ctx;
    if (t_band.ctx) { bbband_destroy(t_band.ctx); t_band.ctx = nullptr; }
    bbband_config cfg = {};
    cfg.device = 0; cfg.width = maxWidth; cfg.semantics = BBBAND_SEMANTICS_JNI_C;      // the C file's semantics, as this symbol had
    if (bbband_create(&cfg, &t_band.ctx) != BBMAP_OK) return nullptr;
    t_band.width = maxWidth;
    return t_band.ctx;
}

void throw_runtime(JNIEnv *env, const char *what) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, what);
}

// shared by the two fills; `limited` selects fillLimitedX
void fill_common(JNIEnv *env, bool limited, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jint minScore,
                 jintArray result, jlongArray iterations, jintArray packed, jint maxRows, jint maxColumns) {
    bbmsa_ctx *ctx = msa_ctx(maxRows, maxColumns);
    if (!ctx) { throw_runtime(env, bbmap_last_error()); return; }
    const jsize readLen = env->GetArrayLength(read), refLen = env->GetArrayLength(ref);
    // the reference borrows every array with GetPrimitiveArrayCritical (jni/MultiStateAligner11tsJNI.c:723-748); so does this
    jbyte *jread = (jbyte *)env->GetPrimitiveArrayCritical(read, nullptr);
    jbyte *jref = (jbyte *)env->GetPrimitiveArrayCritical(ref, nullptr);
    jint *jresult = (jint *)env->GetPrimitiveArrayCritical(result, nullptr);
    jlong *jiter = (jlong *)env->GetPrimitiveArrayCritical(iterations, nullptr);
    jint *jpacked = (jint *)env->GetPrimitiveArrayCritical(packed, nullptr);
    int rc = BBMAP_E_ARG;
    if (jread && jref && jresult && jiter && jpacked) {
        int32_t r5[5] = {0, 0, 0, 0, 0};
        int64_t it = jiter[0];
        rc = bbmsa_fill_packed(ctx, (const uint8_t *)jread, readLen, (const uint8_t *)jref, refLen, refStartLoc, refEndLoc,
                               minScore, limited ? BBMSA_FILL_LIMITED_RAW : BBMSA_FILL_UNLIMITED_RAW, r5, &it, (int32_t *)jpacked);
        if (rc == BBMAP_OK) {
            const int n = limited ? 5 : 4;
            for (int i = 0; i < n; i++) jresult[i] = r5[i];
            jiter[0] = it;                                   // incremented, not set (jni/...c:471)
        }
    }
    if (jpacked) env->ReleasePrimitiveArrayCritical(packed, jpacked, 0);
    if (jiter) env->ReleasePrimitiveArrayCritical(iterations, jiter, 0);
    if (jresult) env->ReleasePrimitiveArrayCritical(result, jresult, 0);
    if (jref) env->ReleasePrimitiveArrayCritical(ref, jref, JNI_ABORT);
    if (jread) env->ReleasePrimitiveArrayCritical(read, jread, JNI_ABORT);
    if (rc != BBMAP_OK) throw_runtime(env, bbmap_last_error());     // the reference calls exit(0) here (jni/...c:130-132)
}

jint band_common(JNIEnv *env, int direction, jbyteArray query, jbyteArray ref, jint qstart, jint rstart, jint maxEdits,
                 jboolean exact, jint maxWidth, jintArray returnVals) {
    bbband_ctx *ctx = band_ctx(maxWidth);
    if (!ctx) { throw_runtime(env, bbmap_last_error()); return 0; }
    const jsize qLen = env->GetArrayLength(query), rLen = env->GetArrayLength(ref);
    // one job: the two sequences back to back in one buffer
    uint8_t *seqs = new uint8_t[(size_t)qLen + (size_t)rLen + 1];
    env->GetByteArrayRegion(query, 0, qLen, (jbyte *)seqs);
    env->GetByteArrayRegion(ref, 0, rLen, (jbyte *)(seqs + qLen));
    bbband_job job;
    job.query_off = 0; job.ref_off = qLen; job.query_len = qLen; job.ref_len = rLen;
    job.qstart = qstart; job.rstart = rstart; job.maxEdits = maxEdits;
    job.flags = direction | (exact ? BBBAND_EXACT : 0);
    bbband_result res;
    const int rc = bbband_align_batch(ctx, 1, &job, seqs, (int64_t)qLen + rLen, &res);
    delete[] seqs;
    if (rc != BBMAP_OK) { throw_runtime(env, bbmap_last_error()); return 0; }
    const jint vals[5] = {res.lastQueryLoc, res.lastRefLoc, res.lastRow, res.lastEdits, res.lastOffset};
    env->SetIntArrayRegion(returnVals, 0, 5, vals);          // fully rewritten, jni/BandedAlignerJNI.c:604-630
    return res.edits;
}

}  // namespace

extern "C" {

// ([B[BII[I[J[I[I[III)V
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI(
    JNIEnv *env, jobject, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jintArray result,
    jlongArray iterationsUnlimited, jintArray packed, jintArray /*POINTSoff_SUB_ARRAY*/, jintArray /*POINTSoff_INS_ARRAY*/,
    jint maxRows, jint maxColumns) {
    fill_common(env, false, read, ref, refStartLoc, refEndLoc, 0, result, iterationsUnlimited, packed, maxRows, maxColumns);
}

// ([B[BIII[I[J[I[I[IIIIF[I[I[B[I)V -- bandwidth / bandwidthRatio of the Java statics are honoured through the context
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(
    JNIEnv *env, jobject, jbyteArray read, jbyteArray ref, jint refStartLoc, jint refEndLoc, jint minScore, jintArray result,
    jlongArray iterationsLimited, jintArray packed, jintArray, jintArray, jint maxRows, jint maxColumns, jint bandwidth,
    jfloat bandwidthRatio, jintArray /*vertLimit*/, jintArray /*horizLimit*/, jbyteArray /*baseToNumber*/, jintArray /*INS_ARRAY_C*/) {
    if (bandwidth >= 1 || bandwidthRatio > 0.0f) {
        // a band changes the fill window: use a context created with that band (not cached per thread here)
        bbmsa_config cfg = {};
        cfg.device = 0; cfg.maxRows = maxRows; cfg.maxColumns = maxColumns; cfg.bandwidth = bandwidth; cfg.bandwidthRatio = bandwidthRatio;
        if (t_msa.ctx) { bbmsa_destroy(t_msa.ctx); t_msa.ctx = nullptr; }
        if (bbmsa_create(&cfg, &t_msa.ctx) != BBMAP_OK) { throw_runtime(env, bbmap_last_error()); return; }
        t_msa.maxRows = maxRows; t_msa.maxColumns = maxColumns;
    }
    fill_common(env, true, read, ref, refStartLoc, refEndLoc, minScore, result, iterationsLimited, packed, maxRows, maxColumns);
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

}  // extern "C"
#endif  // __has_include(<jni.h>)
