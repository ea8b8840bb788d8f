/*
 * libbbmap_amd_jni.so -- the JNI side of the Java classes under jni/java/align2/ (MultiStateAligner11tsHIP, BBIndexHIP, BBMapHIP):
 * the batched seams a BBMap built with those classes binds, as opposed to libbbtoolsjni.so (bbtoolsjni_shim.cpp), which serves the
 * UNMODIFIED classes one fill per call.  Every bulk argument is a direct java.nio.ByteBuffer (little-endian, laid out as the C
 * structs of include/bbmap_amd.h), so no JNI critical region exists anywhere in this file and nothing is copied on the way in
 * (SURVEY.md H2); the only arrays read are the chromosome byte arrays at index construction.
 *
 *   align2.MultiStateAligner11tsHIP   create / destroy / alignBatch / alignGappedBatch -> bbmsa_*
 *        replaces, batched: MSA.fillAndScoreLimited + traceback per site (current/align2/MSA.java:103-134, BBMapThread.java:309, :345)
 *   align2.BBIndexHIP                 build / destroy / setMaxReadLen / findBatch       -> bbidx_*
 *        replaces, batched: AbstractIndex.findAdvanced per read (current/align2/AbstractIndex.java:83, AbstractMapThread.java:736)
 *   align2.BBMapHIP                   create / destroy / mapBatch                       -> bbmap_*
 *        replaces, batched: BBMapThread.processRead / processReadPair up to the end of rescue (current/align2/BBMapThread.java:389-490,
 *        :943-1098)
 *   all three                         lastError                                         -> bbmap_last_error
 *
 * Compiled as C against <jni.h> when a JDK is installed, otherwise against jni/jni_min.h (this image has no JDK); exercised through
 * the mock JNIEnv of jni/mock_jni_test.cpp (`mock_jni_test glue`).
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#include <jni.h>
#define BBGLUE_HAVE_JNI_H 1
#endif
#endif
#ifndef BBGLUE_HAVE_JNI_H
#include "jni_min.h"
#endif

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bbmap_amd.h"

static void throw_runtime(JNIEnv *env, const char *what, int rc) {
    char msg[640];
    snprintf(msg, sizeof msg, "%s failed (%d): %s", what, rc, bbmap_last_error());
    jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (cls) (*env)->ThrowNew(env, cls, msg);
}

/* address of a direct buffer holding at least `need` bytes, or NULL after throwing */
static void *direct(JNIEnv *env, jobject buf, long long need, const char *name) {
    if (!buf) {
        if (need == 0) return NULL;
    } else {
        void *p = (*env)->GetDirectBufferAddress(env, buf);
        const jlong cap = (*env)->GetDirectBufferCapacity(env, buf);
        if (p && cap >= need) return p;
    }
    char msg[200];
    snprintf(msg, sizeof msg, "%s must be a direct ByteBuffer of at least %lld bytes", name, need);
    jclass cls = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (cls) (*env)->ThrowNew(env, cls, msg);
    return NULL;
}

/* ------------------------------------------------------------------ align2.MultiStateAligner11tsHIP */

/* (IIIIFI)J -- scheme: 0 = MultiStateAligner11ts, 1 = MultiStateAligner9PacBio (BBMSA_SCHEME_*) */
JNIEXPORT jlong JNICALL Java_align2_MultiStateAligner11tsHIP_create(JNIEnv *env, jclass cls, jint device, jint maxRows, jint maxColumns,
                                                                    jint bandwidth, jfloat bandwidthRatio, jint scheme) {
    bbmsa_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = device; cfg.maxRows = maxRows; cfg.maxColumns = maxColumns; cfg.bandwidth = bandwidth; cfg.bandwidthRatio = bandwidthRatio;
    cfg.reserved[2] = scheme;
    bbmsa_ctx *ctx = NULL;
    const int rc = bbmsa_create(&cfg, &ctx);
    (void)cls;
    if (rc != BBMAP_OK) { throw_runtime(env, "bbmsa_create", rc); return 0; }
    return (jlong)(intptr_t)ctx;
}

/* (J)V */
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsHIP_destroy(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    bbmsa_destroy((bbmsa_ctx *)(intptr_t)ctx);
}

/* (JILjava/nio/ByteBuffer;Ljava/nio/ByteBuffer;ILjava/nio/ByteBuffer;ILjava/nio/ByteBuffer;Ljava/nio/ByteBuffer;I)V
 * jobs: nJobs bbmsa_job records (40 bytes); reads / refs: the sequence blobs the jobs address; results: nJobs bbmsa_result records
 * (80 bytes); match: nJobs x matchStride bytes, or null with matchStride 0. */
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsHIP_alignBatch(JNIEnv *env, jclass cls, jlong ctx, jint nJobs, jobject jobs, jobject reads,
                                                                       jint readsBytes, jobject refs, jint refsBytes, jobject results,
                                                                       jobject match, jint matchStride) {
    (void)cls;
    if (nJobs <= 0) return;
    const bbmsa_job *pj = (const bbmsa_job *)direct(env, jobs, (long long)nJobs * (long long)sizeof(bbmsa_job), "jobs");
    if (!pj) return;
    const uint8_t *pr = (const uint8_t *)direct(env, reads, readsBytes, "reads");
    if (!pr) return;
    const uint8_t *pf = (const uint8_t *)direct(env, refs, refsBytes, "refs");
    if (!pf) return;
    bbmsa_result *po = (bbmsa_result *)direct(env, results, (long long)nJobs * (long long)sizeof(bbmsa_result), "results");
    if (!po) return;
    uint8_t *pm = matchStride > 0 ? (uint8_t *)direct(env, match, (long long)nJobs * matchStride, "match") : NULL;
    if (matchStride > 0 && !pm) return;
    const int rc = bbmsa_align_batch((bbmsa_ctx *)(intptr_t)ctx, nJobs, pj, pr, readsBytes, pf, refsBytes, po, pm, matchStride);
    if (rc != BBMAP_OK) throw_runtime(env, "bbmsa_align_batch", rc);
}

/* the same with one bbmsa_gaps record (68 bytes) per job: sites that carry a gap array (MSA.fillAndScoreLimited(..., gaps)) */
JNIEXPORT void JNICALL Java_align2_MultiStateAligner11tsHIP_alignGappedBatch(JNIEnv *env, jclass cls, jlong ctx, jint nJobs, jobject jobs,
                                                                             jobject gaps, jobject reads, jint readsBytes, jobject refs,
                                                                             jint refsBytes, jobject results, jobject match, jint matchStride) {
    (void)cls;
    if (nJobs <= 0) return;
    const bbmsa_job *pj = (const bbmsa_job *)direct(env, jobs, (long long)nJobs * (long long)sizeof(bbmsa_job), "jobs");
    if (!pj) return;
    const bbmsa_gaps *pg = (const bbmsa_gaps *)direct(env, gaps, (long long)nJobs * (long long)sizeof(bbmsa_gaps), "gaps");
    if (!pg) return;
    const uint8_t *pr = (const uint8_t *)direct(env, reads, readsBytes, "reads");
    if (!pr) return;
    const uint8_t *pf = (const uint8_t *)direct(env, refs, refsBytes, "refs");
    if (!pf) return;
    bbmsa_result *po = (bbmsa_result *)direct(env, results, (long long)nJobs * (long long)sizeof(bbmsa_result), "results");
    if (!po) return;
    uint8_t *pm = matchStride > 0 ? (uint8_t *)direct(env, match, (long long)nJobs * matchStride, "match") : NULL;
    if (matchStride > 0 && !pm) return;
    const int rc = bbmsa_align_gapped_batch((bbmsa_ctx *)(intptr_t)ctx, nJobs, pj, pg, pr, readsBytes, pf, refsBytes, po, pm, matchStride);
    if (rc != BBMAP_OK) throw_runtime(env, "bbmsa_align_gapped_batch", rc);
}

/* ------------------------------------------------------------------ align2.BBIndexHIP */

/* (IIII[[B)J -- chromArrays[c] = Data.getChromosome(c).array for c = 1..n (entry 0 null): IndexMaker4 + analyzeIndex on the device */
JNIEXPORT jlong JNICALL Java_align2_BBIndexHIP_build(JNIEnv *env, jclass cls, jint device, jint profile, jint k, jint chromBits,
                                                     jobjectArray chromArrays) {
    (void)cls;
    const jsize n1 = (*env)->GetArrayLength(env, chromArrays);          /* nchroms + 1 */
    if (n1 < 2) { throw_runtime(env, "BBIndexHIP.build (no chromosomes)", BBMAP_E_ARG); return 0; }
    const uint8_t **ptr = (const uint8_t **)calloc((size_t)n1, sizeof *ptr);
    int32_t *len = (int32_t *)calloc((size_t)n1, sizeof *len);
    bbidx_ctx *ctx = NULL;
    int rc = (ptr && len) ? BBMAP_OK : BBMAP_E_NOMEM;
    for (jsize c = 1; c < n1 && rc == BBMAP_OK; c++) {
        jbyteArray a = (jbyteArray)(*env)->GetObjectArrayElement(env, chromArrays, c);
        if (!a) { rc = BBMAP_E_ARG; break; }
        len[c] = (*env)->GetArrayLength(env, a);
        uint8_t *copy = (uint8_t *)malloc((size_t)len[c] + 1);
        if (!copy) rc = BBMAP_E_NOMEM;
        else { (*env)->GetByteArrayRegion(env, a, 0, len[c], (jbyte *)copy); ptr[c] = copy; }
        (*env)->DeleteLocalRef(env, a);
    }
    if (rc == BBMAP_OK) rc = bbidx_build_profile(device, profile, k, chromBits, n1 - 1, ptr, len, &ctx);
    if (ptr) for (jsize c = 1; c < n1; c++) free((void *)ptr[c]);
    free(ptr); free(len);
    if (rc != BBMAP_OK) { throw_runtime(env, "bbidx_build_profile", rc); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_align2_BBIndexHIP_destroy(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    bbidx_destroy((bbidx_ctx *)(intptr_t)ctx);
}

JNIEXPORT void JNICALL Java_align2_BBIndexHIP_setMaxReadLen(JNIEnv *env, jclass cls, jlong ctx, jint maxLen) {
    (void)cls;
    const int rc = bbidx_set_max_read_len((bbidx_ctx *)(intptr_t)ctx, maxLen);
    if (rc != BBMAP_OK) throw_runtime(env, "bbidx_set_max_read_len", rc);
}

/* (JILjava/nio/ByteBuffer;Ljava/nio/ByteBuffer;Ljava/nio/ByteBuffer;ILjava/nio/ByteBuffer;ILjava/nio/ByteBuffer;ILjava/nio/ByteBuffer;)V
 * reads: nReads bbidx_read records (24 bytes); bases / baseScores: basesBytes bytes each; keyinfo: keyinfoInts ints (per read
 * offsets[nkeys] then keyScores[nkeys]); sites: nReads x maxSites bbidx_site records (100 bytes); nsites: nReads ints */
JNIEXPORT void JNICALL Java_align2_BBIndexHIP_findBatch(JNIEnv *env, jclass cls, jlong ctx, jint nReads, jobject reads, jobject bases,
                                                        jobject baseScores, jint basesBytes, jobject keyinfo, jint keyinfoInts,
                                                        jobject sites, jint maxSites, jobject nsites) {
    (void)cls;
    if (nReads <= 0) return;
    const bbidx_read *pr = (const bbidx_read *)direct(env, reads, (long long)nReads * (long long)sizeof(bbidx_read), "reads");
    if (!pr) return;
    const uint8_t *pb = (const uint8_t *)direct(env, bases, basesBytes, "bases");
    if (!pb) return;
    const int8_t *ps = (const int8_t *)direct(env, baseScores, basesBytes, "baseScores");
    if (!ps) return;
    const int32_t *pk = (const int32_t *)direct(env, keyinfo, 4LL * keyinfoInts, "keyinfo");
    if (!pk) return;
    bbidx_site *po = (bbidx_site *)direct(env, sites, (long long)nReads * maxSites * (long long)sizeof(bbidx_site), "sites");
    if (!po) return;
    int32_t *pn = (int32_t *)direct(env, nsites, 4LL * nReads, "nsites");
    if (!pn) return;
    const int rc = bbidx_find_batch((bbidx_ctx *)(intptr_t)ctx, nReads, pr, pb, ps, basesBytes, pk, keyinfoInts, po, maxSites, pn);
    if (rc != BBMAP_OK) throw_runtime(env, "bbidx_find_batch", rc);
}

/* ------------------------------------------------------------------ align2.BBMapHIP */

/* (JIZIII)J -- the profile's defaults (bbmap.sh / mapPacBio.sh) with the batch geometry filled in */
JNIEXPORT jlong JNICALL Java_align2_BBMapHIP_create(JNIEnv *env, jclass cls, jlong index, jint profile, jboolean paired, jint maxReads,
                                                    jint maxReadLen, jint maxSites) {
    (void)cls;
    bbmap_config cfg;
    int rc = bbmap_default_config_profile(profile, &cfg);
    bbmap_ctx *ctx = NULL;
    if (rc == BBMAP_OK) {
        cfg.paired = paired ? 1 : 0; cfg.max_reads = maxReads; cfg.max_read_len = maxReadLen; cfg.max_sites = maxSites;
        rc = bbmap_create((bbidx_ctx *)(intptr_t)index, &cfg, &ctx);
    }
    if (rc != BBMAP_OK) { throw_runtime(env, "bbmap_create", rc); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_align2_BBMapHIP_destroy(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    bbmap_destroy((bbmap_ctx *)(intptr_t)ctx);
}

/* returns the number of site records produced (more than sitesCap: call again with a larger buffer).  nsites: nReads ints;
 * offsets: nReads + 1 longs; sites: sitesCap bbmap_msite records (128 bytes), read r's at offsets[r] .. offsets[r] + nsites[r] */
JNIEXPORT jlong JNICALL Java_align2_BBMapHIP_mapBatch(JNIEnv *env, jclass cls, jlong ctx, jint nReads, jobject reads, jobject bases,
                                                      jobject baseScores, jint basesBytes, jobject keyinfo, jint keyinfoInts,
                                                      jobject nsites, jobject offsets, jobject sites, jint sitesCap) {
    (void)cls;
    if (nReads <= 0) return 0;
    const bbidx_read *pr = (const bbidx_read *)direct(env, reads, (long long)nReads * (long long)sizeof(bbidx_read), "reads");
    if (!pr) return 0;
    const uint8_t *pb = (const uint8_t *)direct(env, bases, basesBytes, "bases");
    if (!pb) return 0;
    const int8_t *ps = (const int8_t *)direct(env, baseScores, basesBytes, "baseScores");
    if (!ps) return 0;
    const int32_t *pk = (const int32_t *)direct(env, keyinfo, 4LL * keyinfoInts, "keyinfo");
    if (!pk) return 0;
    int32_t *pn = (int32_t *)direct(env, nsites, 4LL * nReads, "nsites");
    if (!pn) return 0;
    int64_t *pf = (int64_t *)direct(env, offsets, 8LL * (nReads + 1), "offsets");
    if (!pf) return 0;
    bbmap_msite *po = (bbmap_msite *)direct(env, sites, (long long)sitesCap * (long long)sizeof(bbmap_msite), "sites");
    if (sitesCap > 0 && !po) return 0;
    int64_t total = 0;
    const int rc = bbmap_map_batch((bbmap_ctx *)(intptr_t)ctx, nReads, pr, pb, basesBytes, ps, pk, keyinfoInts, pn, pf, po, sitesCap, &total);
    if (rc != BBMAP_OK) { throw_runtime(env, "bbmap_map_batch", rc); return 0; }
    return (jlong)total;
}

/* (JILjava/nio/ByteBuffer;Ljava/nio/ByteBuffer;I)J -- the last batch's final records (bbmap_final, 64 bytes each: what BBMap prints per read:
 * mapped, chrom, strand, start, stop, mapScore, paired, ambiguous, perfect, rescued, match length / offset) and their match strings packed
 * into `match`; returns the bytes the strings take (more than matchCap: call again with a larger buffer; `match` may be null: records only) */
JNIEXPORT jlong JNICALL Java_align2_BBMapHIP_getFinal(JNIEnv *env, jclass cls, jlong ctx, jint nReads, jobject records, jobject match, jint matchCap) {
    (void)cls;
    if (nReads <= 0) return 0;
    bbmap_final *pf = (bbmap_final *)direct(env, records, (long long)nReads * (long long)sizeof(bbmap_final), "records");
    if (!pf) return 0;
    uint8_t *pm = NULL;
    if (match) { pm = (uint8_t *)direct(env, match, matchCap, "match"); if (!pm) return 0; }
    int64_t bytes = 0;
    const int rc = bbmap_get_final((bbmap_ctx *)(intptr_t)ctx, nReads, pf, pm, pm ? matchCap : 0, &bytes);
    if (rc != BBMAP_OK) { throw_runtime(env, "bbmap_get_final", rc); return 0; }
    return (jlong)bytes;
}

/* ([B)I -- copies the calling thread's last error text into buf (UTF-8, truncated), returns its length */
JNIEXPORT jint JNICALL Java_align2_BBMapHIP_lastError(JNIEnv *env, jclass cls, jbyteArray buf) {
    (void)cls;
    const char *msg = bbmap_last_error();
    jsize n = (jsize)strlen(msg);
    const jsize cap = (*env)->GetArrayLength(env, buf);
    if (n > cap) n = cap;
    (*env)->SetByteArrayRegion(env, buf, 0, n, (const jbyte *)msg);
    return n;
}
