// Exercises the Java_* entry points of libbbtoolsjni.so against a mock JNIEnv (no JVM in this image): arrays are plain
// buffers behind jobject handles, the function table carries the calls the shim makes, and the mock FAILS the test if a
// critical region is open while any other JNI call is made or while more than one region is open (the JNI
// specification forbids blocking inside a critical region).  Usage: mock_jni_test merge | gpu | threads [N] | glue   (exit code 0 = pass)
// `glue`: the natives of jni/hip_glue.c (the Java classes under jni/java/align2/) with direct-buffer arguments: the batched aligner,
// the index probe and the whole mapper flow through the JNI layer equal the same calls on the C ABI.
// `threads`: N mapping threads (default 32), each with its own arrays like one MSA object per thread, call the two fill symbols
// concurrently; every result and every plane must equal what the same call gave alone, and the calls/s of 1 and N threads are
// printed.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "jni_min.h"
#include "bbmerge_overlap.h"
#include "bbmap_amd.h"

struct _jobject { void *data; int len; int elem; };
static thread_local int g_critical = 0;  // open critical regions of the calling thread
static std::atomic<int> g_violations{0}, g_thrown{0};
static std::atomic<size_t> g_maxRegionBytes{0};      // largest Get<Type>ArrayRegion copy seen
static void touch() { if (g_critical) g_violations++; }

static jclass mFindClass(JNIEnv *, const char *) { touch(); static _jobject cls = {nullptr, 0, 0}; return &cls; }
static jint mThrowNew(JNIEnv *, jclass, const char *msg) { touch(); g_thrown++; fprintf(stderr, "thrown: %s\n", msg); return 0; }
static jsize mGetArrayLength(JNIEnv *, jarray a) { touch(); return a->len; }
template <class T> static void getRegion(jarray a, jsize s, jsize l, T *b) { touch(); if (sizeof(T) * (size_t)l > g_maxRegionBytes.load()) g_maxRegionBytes = sizeof(T) * (size_t)l; memcpy(b, (T *)a->data + s, sizeof(T) * (size_t)l); }
template <class T> static void setRegion(jarray a, jsize s, jsize l, const T *b) { touch(); memcpy((T *)a->data + s, b, sizeof(T) * (size_t)l); }
static void mGetByte(JNIEnv *, jbyteArray a, jsize s, jsize l, jbyte *b) { getRegion(a, s, l, b); }
static void mGetInt(JNIEnv *, jintArray a, jsize s, jsize l, jint *b) { getRegion(a, s, l, b); }
static void mGetLong(JNIEnv *, jlongArray a, jsize s, jsize l, jlong *b) { getRegion(a, s, l, b); }
static void mGetFloat(JNIEnv *, jfloatArray a, jsize s, jsize l, jfloat *b) { getRegion(a, s, l, b); }
static void mSetInt(JNIEnv *, jintArray a, jsize s, jsize l, const jint *b) { setRegion(a, s, l, b); }
static void mSetLong(JNIEnv *, jlongArray a, jsize s, jsize l, const jlong *b) { setRegion(a, s, l, b); }
static void mSetFloat(JNIEnv *, jfloatArray a, jsize s, jsize l, const jfloat *b) { setRegion(a, s, l, b); }
static jobject mGetObjectArrayElement(JNIEnv *, jobjectArray a, jsize i) { touch(); return ((jobject *)a->data)[i]; }
static void mDeleteLocalRef(JNIEnv *, jobject) { touch(); }
static void *mGetDirectBufferAddress(JNIEnv *, jobject b) { touch(); return b->data; }
static jlong mGetDirectBufferCapacity(JNIEnv *, jobject b) { touch(); return b->len; }
static void mSetByte(JNIEnv *, jbyteArray a, jsize s, jsize l, const jbyte *b) { setRegion(a, s, l, b); }
static void *mGetCritical(JNIEnv *, jarray a, jboolean *) { if (g_critical) g_violations++; g_critical++; return a->data; }
static void mReleaseCritical(JNIEnv *, jarray, void *, jint) { g_critical--; }

extern "C" {
void Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI(JNIEnv *, jobject, jbyteArray, jbyteArray, jint, jint, jintArray, jlongArray, jintArray, jintArray, jintArray, jint, jint);
void Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(JNIEnv *, jobject, jbyteArray, jbyteArray, jint, jint, jint, jintArray, jlongArray, jintArray, jintArray, jintArray, jint, jint, jint, jfloat, jintArray, jintArray, jbyteArray, jintArray);
jint Java_align2_BandedAlignerJNI_alignForwardJNI(JNIEnv *, jobject, jbyteArray, jbyteArray, jint, jint, jint, jboolean, jint, jbyteArray, jintArray);
jint Java_jgi_BBMergeOverlapper_mateByOverlapJNI(JNIEnv *, jclass, jbyteArray, jbyteArray, jbyteArray, jbyteArray, jfloatArray, jfloatArray, jintArray, jint, jint, jint, jint, jint, jint, jint);
jint Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI_1WithQualities(JNIEnv *, jclass, jbyteArray, jbyteArray, jbyteArray, jbyteArray, jfloatArray, jfloatArray, jintArray, jint, jint, jint, jint, jfloat, jfloat, jfloat);
jint Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI(JNIEnv *, jclass, jbyteArray, jbyteArray, jintArray, jint, jint, jint, jint, jfloat, jfloat, jfloat, jfloat, jfloat);
// jni/hip_glue.c
jlong Java_align2_MultiStateAligner11tsHIP_create(JNIEnv *, jclass, jint, jint, jint, jint, jfloat, jint);
void Java_align2_MultiStateAligner11tsHIP_destroy(JNIEnv *, jclass, jlong);
void Java_align2_MultiStateAligner11tsHIP_alignBatch(JNIEnv *, jclass, jlong, jint, jobject, jobject, jint, jobject, jint, jobject, jobject, jint);
jlong Java_align2_BBIndexHIP_build(JNIEnv *, jclass, jint, jint, jint, jint, jobjectArray);
void Java_align2_BBIndexHIP_destroy(JNIEnv *, jclass, jlong);
void Java_align2_BBIndexHIP_setMaxReadLen(JNIEnv *, jclass, jlong, jint);
void Java_align2_BBIndexHIP_findBatch(JNIEnv *, jclass, jlong, jint, jobject, jobject, jobject, jint, jobject, jint, jobject, jint, jobject);
jlong Java_align2_BBMapHIP_create(JNIEnv *, jclass, jlong, jint, jboolean, jint, jint, jint);
void Java_align2_BBMapHIP_destroy(JNIEnv *, jclass, jlong);
jlong Java_align2_BBMapHIP_mapBatch(JNIEnv *, jclass, jlong, jint, jobject, jobject, jobject, jint, jobject, jint, jobject, jobject, jobject, jint);
jlong Java_align2_BBMapHIP_getFinal(JNIEnv *, jclass, jlong, jint, jobject, jobject, jint);
jint Java_align2_BBMapHIP_lastError(JNIEnv *, jclass, jbyteArray);
int bbjni_fill(int, const uint8_t *, int, const uint8_t *, int, int, int, int, int32_t *, int64_t *, int32_t *, int, int, int, float);
void bbjni_release_thread(void);
void bbjni_release_all(void);
void bbjni_legacy_stats(int64_t *);
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

static int test_merge(JNIEnv *env) {
    unsigned seed = 7;
    for (int trial = 0; trial < 200; trial++) {
        const int alen = 60 + (int)(rnd(seed) % 90), blen = 60 + (int)(rnd(seed) % 90), ov = 20 + (int)(rnd(seed) % 40);
        std::vector<jbyte> a((size_t)alen), b((size_t)blen), aq((size_t)alen), bq((size_t)blen);
        for (auto &x : a) x = "ACGT"[rnd(seed) & 3];
        for (int i = 0; i < blen; i++) b[(size_t)i] = (i < ov) ? a[(size_t)(alen - ov + i)] : "ACGT"[rnd(seed) & 3];   // b overlaps a's tail
        for (int i = 0; i < (int)(rnd(seed) % 4); i++) b[(size_t)(rnd(seed) % (unsigned)ov)] = 'N';
        for (auto &x : aq) x = (jbyte)(2 + rnd(seed) % 39);
        for (auto &x : bq) x = (jbyte)(2 + rnd(seed) % 39);
        const int cap = (alen > blen ? alen : blen) + 1;
        std::vector<jfloat> ap((size_t)cap, 0.0f), bp((size_t)cap, 0.0f), ap2((size_t)cap, 0.0f), bp2((size_t)cap, 0.0f);
        jint rv[5] = {0, 0, 0, 0, 0}, rv2[5] = {0, 0, 0, 0, 0};
        _jobject ja{a.data(), alen, 1}, jb{b.data(), blen, 1}, jaq{aq.data(), alen, 1}, jbq{bq.data(), blen, 1};
        _jobject jap{ap.data(), cap, 4}, jbp{bp.data(), cap, 4}, jrv{rv, 5, 4};
        const jint r1 = Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI(env, nullptr, &ja, &jb, &jrv, 8, 12, 35, 35, 0.075f, 2.0f, 0.55f, 0.65f, 0.95f);
        const jint e1 = bbmerge_mate_by_overlap_ratio(a.data(), alen, b.data(), blen, rv2, 8, 12, 35, 35, 0.075f, 2.0f, 0.55f, 0.65f, 0.95f);
        CHECK(r1 == e1 && memcmp(rv, rv2, sizeof rv) == 0);
        const jint r2 = Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI_1WithQualities(env, nullptr, &ja, &jb, &jaq, &jbq, &jap, &jbp, &jrv, 8, 12, 35, 35, 0.075f, 2.0f, 0.55f);
        const jint e2 = bbmerge_mate_by_overlap_ratio_with_qualities(a.data(), alen, b.data(), blen, aq.data(), bq.data(), ap2.data(), bp2.data(), rv2, 8, 12, 35, 35, 0.075f, 2.0f, 0.55f);
        CHECK(r2 == e2 && memcmp(rv, rv2, sizeof rv) == 0);
        const jint r3 = Java_jgi_BBMergeOverlapper_mateByOverlapJNI(env, nullptr, &ja, &jb, &jaq, &jbq, &jap, &jbp, &jrv, 8, 14, 35, 2, 3, 3, 10);
        const jint e3 = bbmerge_mate_by_overlap(a.data(), alen, b.data(), blen, aq.data(), bq.data(), ap2.data(), bp2.data(), rv2, 8, 14, 35, 2, 3, 3, 10);
        if (!(r3 == e3 && memcmp(rv, rv2, sizeof rv) == 0)) fprintf(stderr, "trial %d r3=%d e3=%d rv=%d,%d,%d,%d,%d rv2=%d,%d,%d,%d,%d\n", trial, r3, e3, rv[0],rv[1],rv[2],rv[3],rv[4], rv2[0],rv2[1],rv2[2],rv2[3],rv2[4]);
        CHECK(r3 == e3 && memcmp(rv, rv2, sizeof rv) == 0);
    }
    CHECK(g_violations == 0 && g_critical == 0 && g_thrown == 0);
    printf("mock JNI: 600 BBMerge calls equal the plain functions, no JNI call inside a critical region\n");
    return 0;
}

static int test_gpu(JNIEnv *env) {
    const int maxRows = 160, maxColumns = 300;
    unsigned seed = 11;
    std::vector<jbyte> ref(2000);
    for (auto &x : ref) x = "ACGT"[rnd(seed) & 3];
    const size_t plane = (size_t)(maxRows + 1) * (maxColumns + 1);
    std::vector<jint> packed(3 * plane, 0x55555555), packed2(3 * plane, 0x55555555);
    for (int trial = 0; trial < 12; trial++) {
        const int st = 100 + (int)(rnd(seed) % 1500), len = 100 + (int)(rnd(seed) % 50);
        std::vector<jbyte> rd(ref.begin() + st, ref.begin() + st + len);
        rd[(size_t)(rnd(seed) % (unsigned)len)] = 'T';
        if (trial & 1) rd.erase(rd.begin() + 40, rd.begin() + 43);
        const int rlen = (int)rd.size(), a = st - 4, b = st + len + 3;
        const int minScore = (int)(0.56f * (70 + 100 * (rlen - 1)));
        jint res[5] = {0, 0, 0, 0, 0}; jlong it[1] = {1000};
        _jobject jrd{rd.data(), rlen, 1}, jrf{ref.data(), (int)ref.size(), 1}, jres{res, 5, 4}, jit{it, 1, 8}, jpk{packed.data(), (int)packed.size(), 4};
        const bool limited = (trial % 3) != 0;
        const int bw = (trial % 4 == 3) ? 40 : 0; const float bwr = bw ? 0.18f : 0.0f;
        if (limited) Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(env, nullptr, &jrd, &jrf, a, b, minScore, &jres, &jit, &jpk, nullptr, nullptr, maxRows, maxColumns, bw, bwr, nullptr, nullptr, nullptr, nullptr);
        else Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI(env, nullptr, &jrd, &jrf, a, b, &jres, &jit, &jpk, nullptr, nullptr, maxRows, maxColumns);
        CHECK(g_thrown == 0 && g_violations == 0 && g_critical == 0);
        // the same call through the plain layer (what tests/test_msa_gpu.py checks against the oracle's walkers)
        int32_t r2[5] = {0, 0, 0, 0, 0}; int64_t it2 = 1000;
        CHECK(bbjni_fill(limited, (const uint8_t *)rd.data(), rlen, (const uint8_t *)ref.data(), (int)ref.size(), a, b, minScore, r2, &it2, packed2.data(), maxRows, maxColumns, bw, bwr) == 0);
        for (int i = 0; i < (limited ? 5 : 4); i++) CHECK(res[i] == r2[i]);
        CHECK(it[0] == it2 && it2 > 1000);
        for (int s = 0; s < 3; s++) CHECK(memcmp(packed.data() + s * plane, packed2.data() + s * plane, (size_t)(rlen + 1) * (maxColumns + 1) * 4) == 0);
        CHECK(res[0] == rlen);
    }
    {   // `ref` is a whole chromosome in BBMap: a 120 MB array with the window at its far end must cost what a small one costs
        // (only [refStartLoc, refEndLoc] may leave the JVM) and give the planes of the same window in a small array
        const size_t big = (size_t)120 << 20;
        std::vector<jbyte> chrom(big, (jbyte)'N');
        const size_t base = big - ref.size();
        memcpy(chrom.data() + base, ref.data(), ref.size());
        const int st = 700, len = 140;
        std::vector<jbyte> rd(ref.begin() + st, ref.begin() + st + len);
        rd[60] = (rd[60] == 'A') ? 'C' : 'A';
        const int a = st - 4, b = st + len + 3, minScore = (int)(0.56f * (70 + 100 * (len - 1)));
        jint res[5] = {0, 0, 0, 0, 0}; jlong it[1] = {0};
        _jobject jrd{rd.data(), len, 1}, jrf{chrom.data(), (int)big, 1}, jres{res, 5, 4}, jit{it, 1, 8}, jpk{packed.data(), (int)packed.size(), 4};
        Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(env, nullptr, &jrd, &jrf, (jint)(base + a), (jint)(base + b), minScore, &jres, &jit, &jpk,
                                                             nullptr, nullptr, maxRows, maxColumns, 0, 0.0f, nullptr, nullptr, nullptr, nullptr);
        CHECK(g_thrown == 0 && g_violations == 0 && g_critical == 0);
        int32_t r2[5] = {0, 0, 0, 0, 0}; int64_t it2 = 0;
        CHECK(bbjni_fill(1, (const uint8_t *)rd.data(), len, (const uint8_t *)ref.data(), (int)ref.size(), a, b, minScore, r2, &it2, packed2.data(), maxRows, maxColumns, 0, 0.0f) == 0);
        for (int i = 0; i < 5; i++) CHECK(res[i] == r2[i]);
        CHECK(it[0] == it2 && it2 > 0);
        for (int s = 0; s < 3; s++) CHECK(memcmp(packed.data() + s * plane, packed2.data() + s * plane, (size_t)(len + 1) * (maxColumns + 1) * 4) == 0);
        CHECK(g_maxRegionBytes <= 4096);                       // no JNI region copy anywhere near the chromosome's size
    }
    {   // BandedAligner symbol: the survey's known answer (edits 2, {19,18,19,2,1})
        jbyte q[] = "ACGTTGCAAGCTTAGGCTTA", r[] = "ACGTTGCAGCTTAGGCTTAC";
        jint rv[5] = {9, 9, 9, 9, 9};
        _jobject jq{q, 20, 1}, jr{r, 20, 1}, jrv{rv, 5, 4};
        const jint e = Java_align2_BandedAlignerJNI_alignForwardJNI(env, nullptr, &jq, &jr, 0, 0, 5, 1, 11, nullptr, &jrv);
        CHECK(e == 2 && rv[0] == 19 && rv[1] == 18 && rv[2] == 19 && rv[3] == 2 && rv[4] == 1);
    }
    bbjni_release_thread();
    bbjni_release_all();
    CHECK(g_violations == 0 && g_critical == 0 && g_thrown == 0);
    printf("mock JNI: fills and the banded symbol through the JNI layer equal the plain layer, one short critical region per fill\n");
    return 0;
}


// N mapping threads against the process-wide context.  A "call" is what BBMap issues per site: fillLimitedX (or fillUnlimited)
// on a 150-base read against a window with the mapper's padding.
struct Call { std::vector<jbyte> rd; int a, b, minScore; bool limited; jint res[5]; jlong it; unsigned long long sum; std::vector<jint> vl, hl; };

static unsigned long long rect_sum(const std::vector<jint> &packed, size_t plane, int maxColumns, int rows, int cols) {
    unsigned long long h = 1469598103934665603ull;
    for (int s = 0; s < 3; s++)
        for (int r = 1; r <= rows; r++)
            for (int c = 1; c <= cols; c++) { h ^= (unsigned)packed[s * plane + (size_t)r * (maxColumns + 1) + c]; h *= 1099511628211ull; }
    return h;
}

static int run_calls(JNIEnv *env, std::vector<jbyte> &ref, std::vector<Call> &calls, size_t first, size_t step, int maxRows, int maxColumns,
                     bool record, int *failed) {
    const size_t plane = (size_t)(maxRows + 1) * (maxColumns + 1);
    std::vector<jint> packed(3 * plane, 0), vl((size_t)maxRows + 1), hl((size_t)maxColumns + 1);
    for (size_t i = first; i < calls.size(); i += step) {
        Call &c = calls[i];
        jint res[5] = {0, 0, 0, 0, 0}; jlong it[1] = {0};
        const int rlen = (int)c.rd.size(), cols = c.b - c.a + 1;
        _jobject jrd{c.rd.data(), rlen, 1}, jrf{ref.data(), (int)ref.size(), 1}, jres{res, 5, 4}, jit{it, 1, 8}, jpk{packed.data(), (int)packed.size(), 4};
        _jobject jvl{vl.data(), maxRows + 1, 4}, jhl{hl.data(), maxColumns + 1, 4};
        if (c.limited) Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI(env, nullptr, &jrd, &jrf, c.a, c.b, c.minScore, &jres, &jit, &jpk, nullptr, nullptr, maxRows, maxColumns, 0, 0.0f, &jvl, &jhl, nullptr, nullptr);
        else Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI(env, nullptr, &jrd, &jrf, c.a, c.b, &jres, &jit, &jpk, nullptr, nullptr, maxRows, maxColumns);
        const unsigned long long sum = rect_sum(packed, plane, maxColumns, rlen, cols);
        if (record) {
            memcpy(c.res, res, sizeof res); c.it = it[0]; c.sum = sum;
            if (c.limited) { c.vl.assign(vl.begin(), vl.begin() + rlen + 1); c.hl.assign(hl.begin(), hl.begin() + cols + 1); }
        } else {
            bool ok = memcmp(c.res, res, (c.limited ? 5 : 4) * sizeof(jint)) == 0 && c.it == it[0] && c.sum == sum;
            if (ok && c.limited) ok = memcmp(c.vl.data(), vl.data(), (size_t)(rlen + 1) * 4) == 0 && memcmp(c.hl.data(), hl.data(), (size_t)(cols + 1) * 4) == 0;
            if (!ok) { (*failed)++; fprintf(stderr, "call %zu differs from its solo run\n", i); }
        }
    }
    return 0;
}

static int test_threads(JNINativeInterface_ *tbl, int nthreads) {
    const int maxRows = 601, maxColumns = 2000;          // the MSA shape BBMap's mapping threads create (ALIGN_ROWS 601)
    unsigned seed = 23;
    std::vector<jbyte> ref(200000);
    for (auto &x : ref) x = "ACGT"[rnd(seed) & 3];
    const int ncalls = 4096;
    std::vector<Call> calls((size_t)ncalls);
    for (int i = 0; i < ncalls; i++) {
        Call &c = calls[(size_t)i];
        const int st = 1000 + (int)(rnd(seed) % 190000), len = 150;
        c.rd.assign(ref.begin() + st, ref.begin() + st + len + 8);
        for (int k = 0; k < (int)(rnd(seed) % 4); k++) c.rd[(size_t)(rnd(seed) % (unsigned)len)] = "ACGT"[rnd(seed) & 3];
        if (i % 5 == 1) c.rd.erase(c.rd.begin() + 70, c.rd.begin() + 70 + 1 + (int)(rnd(seed) % 6));
        if (i % 7 == 2) c.rd[(size_t)(rnd(seed) % (unsigned)len)] = 'N';
        c.rd.resize((size_t)len);
        c.a = st - 8 - (int)(rnd(seed) % 8); c.b = st + len + 8 + (int)(rnd(seed) % 24);
        if (i % 1024 == 5) c.b = c.a + len - 5;          // a window narrower than the read (chromosome ends only, in BBMap): handed to the one-thread kernel
        c.limited = (i % 6) != 0;
        c.minScore = (int)(0.56f * (70 + 100 * (len - 1)));
    }
    JNIEnv_ env; env.functions = tbl;
    int failed = 0;
    // solo: one call at a time, recorded (this is also the single-thread rate)
    auto t0 = std::chrono::steady_clock::now();
    run_calls(&env, ref, calls, 0, 1, maxRows, maxColumns, true, &failed);
    const double solo = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t st0[6]; bbjni_legacy_stats(st0);
    // together
    std::vector<std::thread> th;
    std::vector<int> fails((size_t)nthreads, 0);
    t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            JNIEnv_ e; e.functions = tbl;
            run_calls(&e, ref, calls, (size_t)t, (size_t)nthreads, maxRows, maxColumns, false, &fails[(size_t)t]);
            if (g_critical != 0) fails[(size_t)t]++;
        });
    for (auto &x : th) x.join();
    const double together = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int f : fails) failed += f;
    int64_t st1[6]; bbjni_legacy_stats(st1);
    CHECK(failed == 0);
    CHECK(g_violations == 0 && g_thrown == 0);
    CHECK(st0[0] == ncalls && st0[1] == ncalls);                       // alone: one launch per call
    CHECK(st1[0] == 2 * ncalls && st1[2] > 0);
    if (nthreads >= 8) CHECK(st1[1] - st0[1] < ncalls / 2);             // together: calls were combined
    printf("mock JNI legacy fills (150-base reads, MSA 601 x 2000): 1 thread %.0f calls/s; %d threads %.0f calls/s in %lld launches "
           "(%.1f calls per launch); %lld of %d calls went to the one-thread kernel; all equal to their solo runs\n",
           ncalls / solo, nthreads, ncalls / together, (long long)(st1[1] - st0[1]), (double)ncalls / (double)(st1[1] - st0[1]),
           (long long)(st1[2] - st0[2]), ncalls);
    printf("  leader time per launch, solo: wavefront pass %.0f us, hand-over pass %.0f us per handed fill; together: wavefront pass %.0f us, "
           "hand-over %.0f us per handed fill, waiting for collectors %.0f us\n", st0[3] / 1e3 / st0[1], st0[2] ? st0[4] / 1e3 / st0[2] : 0.0,
           (st1[3] - st0[3]) / 1e3 / (st1[1] - st0[1]), (st1[2] - st0[2]) ? (st1[4] - st0[4]) / 1e3 / (st1[2] - st0[2]) : 0.0,
           (st1[5] - st0[5]) / 1e3 / (st1[1] - st0[1]));
    bbjni_release_all();
    return 0;
}

// The natives behind jni/java/align2/*.java.  Every bulk argument is a "direct buffer" (a plain allocation behind a handle here).
static int test_glue(JNIEnv *env) {
    unsigned seed = 5;      // (bases from bits 22-23 of the generator: its low bits repeat every 1,024 draws)
    // ---- MultiStateAligner11tsHIP.alignBatch == bbmsa_align_batch
    {
        const int maxRows = 200, maxColumns = 600, n = 96;
        std::vector<uint8_t> ref(20000), reads;
        for (auto &x : ref) x = (uint8_t)"ACGT"[(rnd(seed) >> 14) & 3];
        std::vector<bbmsa_job> jobs((size_t)n);
        for (int i = 0; i < n; i++) {
            const int st = 500 + (int)(rnd(seed) % 18000), len = 100 + (int)(rnd(seed) % 51);
            std::vector<uint8_t> rd(ref.begin() + st, ref.begin() + st + len + 8);
            for (int k = 0; k < (int)(rnd(seed) % 4); k++) rd[rnd(seed) % (unsigned)len] = (uint8_t)"ACGT"[(rnd(seed) >> 14) & 3];
            if (i % 4 == 1) rd.erase(rd.begin() + 50, rd.begin() + 50 + 1 + (int)(rnd(seed) % 5));
            rd.resize((size_t)len);
            bbmsa_job &j = jobs[(size_t)i];
            j.read_off = (int64_t)reads.size(); j.ref_off = 0; j.read_len = len; j.ref_len = (int)ref.size();
            j.refStartLoc = st - 6; j.refEndLoc = st + len + 12; j.minScore = (int)(0.56f * (70 + 100 * (len - 1)));
            j.flags = BBMSA_FILL_AND_SCORE_LIMITED | BBMSA_DO_TRACEBACK;
            reads.insert(reads.end(), rd.begin(), rd.end());
        }
        const int stride = 1024;
        std::vector<bbmsa_result> res((size_t)n), res2((size_t)n);
        std::vector<uint8_t> match((size_t)n * stride, 0), match2((size_t)n * stride, 0);
        memset(res.data(), 0, res.size() * sizeof(bbmsa_result)); memset(res2.data(), 0, res2.size() * sizeof(bbmsa_result));
        const jlong h = Java_align2_MultiStateAligner11tsHIP_create(env, nullptr, 0, maxRows, maxColumns, 0, 0.0f, 0);
        CHECK(h != 0 && g_thrown == 0);
        _jobject bj{jobs.data(), (int)(jobs.size() * sizeof(bbmsa_job)), 1}, br{reads.data(), (int)reads.size(), 1}, bf{ref.data(), (int)ref.size(), 1};
        _jobject bo{res.data(), (int)(res.size() * sizeof(bbmsa_result)), 1}, bm{match.data(), (int)match.size(), 1};
        Java_align2_MultiStateAligner11tsHIP_alignBatch(env, nullptr, h, n, &bj, &br, (jint)reads.size(), &bf, (jint)ref.size(), &bo, &bm, stride);
        CHECK(g_thrown == 0);
        CHECK(bbmsa_align_batch((bbmsa_ctx *)(intptr_t)h, n, jobs.data(), reads.data(), (int64_t)reads.size(), ref.data(), (int64_t)ref.size(),
                                res2.data(), match2.data(), stride) == BBMAP_OK);
        CHECK(memcmp(res.data(), res2.data(), res.size() * sizeof(bbmsa_result)) == 0 && match == match2);
        int scored = 0;
        for (auto &r : res) scored += r.score_len > 0 && r.match_len > 0;
        CHECK(scored > n / 2);
        // a buffer that is too small is refused with an exception, nothing runs
        _jobject small{res.data(), 16, 1};
        Java_align2_MultiStateAligner11tsHIP_alignBatch(env, nullptr, h, n, &bj, &br, (jint)reads.size(), &bf, (jint)ref.size(), &small, &bm, stride);
        CHECK(g_thrown == 1);
        g_thrown = 0;
        Java_align2_MultiStateAligner11tsHIP_destroy(env, nullptr, h);
    }
    // ---- BBIndexHIP.build / findBatch and BBMapHIP.mapBatch == the C ABI; reads come home to their origins
    {
        const int glen = 300000, nreads = 400, L = 150;
        std::vector<jbyte> chrom((size_t)glen);
        for (auto &x : chrom) x = "ACGT"[(rnd(seed) >> 14) & 3];
        _jobject chromArr{chrom.data(), glen, 1};
        jobject arr[2] = {nullptr, &chromArr};
        _jobject outer{arr, 2, 8};
        const jlong ix = Java_align2_BBIndexHIP_build(env, nullptr, 0, BBIDX_PROFILE_BBMAP, 13, -1, &outer);
        CHECK(ix != 0 && g_thrown == 0);
        Java_align2_BBIndexHIP_setMaxReadLen(env, nullptr, ix, L);
        bbkeys_config kc;
        CHECK(bbkeys_default_config(BBIDX_PROFILE_BBMAP, &kc) == BBMAP_OK);
        kc.k = 13;
        std::vector<uint8_t> bases((size_t)nreads * L);
        std::vector<int8_t> bscores((size_t)nreads * L);
        std::vector<bbidx_read> recs((size_t)nreads);
        std::vector<int32_t> keyinfo;
        std::vector<int> origin((size_t)nreads);
        for (int r = 0; r < nreads; r++) {
            const int st = 1000 + (int)(rnd(seed) % (unsigned)(glen - 3000));
            origin[(size_t)r] = st;
            uint8_t *b = bases.data() + (size_t)r * L;
            memcpy(b, chrom.data() + st, (size_t)L);
            for (int k = 0; k < (int)(rnd(seed) % 4); k++) b[rnd(seed) % L] = (uint8_t)"ACGT"[(rnd(seed) >> 14) & 3];
            int32_t offs[64], ks[64];
            const int nk = bbkeys_make(&kc, b, nullptr, L, offs, ks, 64, bscores.data() + (size_t)r * L);
            CHECK(nk > 0);
            recs[(size_t)r] = bbidx_read{(int64_t)r * L, (int64_t)keyinfo.size(), L, nk};
            keyinfo.insert(keyinfo.end(), offs, offs + nk);
            keyinfo.insert(keyinfo.end(), ks, ks + nk);
        }
        const int maxSites = 32;
        std::vector<bbidx_site> sites((size_t)nreads * maxSites), sites2((size_t)nreads * maxSites);
        std::vector<int32_t> ns((size_t)nreads, -9), ns2((size_t)nreads, -9);
        memset(sites.data(), 0, sites.size() * sizeof(bbidx_site)); memset(sites2.data(), 0, sites2.size() * sizeof(bbidx_site));
        _jobject bR{recs.data(), (int)(recs.size() * sizeof(bbidx_read)), 1}, bB{bases.data(), (int)bases.size(), 1}, bS{bscores.data(), (int)bscores.size(), 1};
        _jobject bK{keyinfo.data(), (int)(keyinfo.size() * 4), 1}, bO{sites.data(), (int)(sites.size() * sizeof(bbidx_site)), 1}, bN{ns.data(), (int)(ns.size() * 4), 1};
        Java_align2_BBIndexHIP_findBatch(env, nullptr, ix, nreads, &bR, &bB, &bS, (jint)bases.size(), &bK, (jint)keyinfo.size(), &bO, maxSites, &bN);
        CHECK(g_thrown == 0);
        CHECK(bbidx_find_batch((bbidx_ctx *)(intptr_t)ix, nreads, recs.data(), bases.data(), bscores.data(), (int64_t)bases.size(), keyinfo.data(),
                               (int64_t)keyinfo.size(), sites2.data(), maxSites, ns2.data()) == BBMAP_OK);
        CHECK(ns == ns2);
        for (int r = 0; r < nreads; r++)
            for (int s2 = 0; s2 < ns[(size_t)r]; s2++)
                CHECK(memcmp(&sites[(size_t)r * maxSites + s2], &sites2[(size_t)r * maxSites + s2], sizeof(bbidx_site)) == 0);
        // the whole flow
        const jlong mp = Java_align2_BBMapHIP_create(env, nullptr, ix, BBIDX_PROFILE_BBMAP, 0, nreads, L, maxSites);
        CHECK(mp != 0 && g_thrown == 0);
        const int cap = 8 * nreads;
        std::vector<bbmap_msite> ms((size_t)cap);
        std::vector<int32_t> mn((size_t)nreads, -9);
        std::vector<int64_t> mo((size_t)nreads + 1, -9);
        _jobject bMN{mn.data(), (int)(mn.size() * 4), 1}, bMO{mo.data(), (int)(mo.size() * 8), 1}, bMS{ms.data(), (int)(ms.size() * sizeof(bbmap_msite)), 1};
        const jlong total = Java_align2_BBMapHIP_mapBatch(env, nullptr, mp, nreads, &bR, &bB, &bS, (jint)bases.size(), &bK, (jint)keyinfo.size(), &bMN, &bMO, &bMS, cap);
        if (!(g_thrown == 0 && total > 0 && total <= cap)) fprintf(stderr, "mapBatch: thrown %d total %lld cap %d\n", (int)g_thrown, (long long)total, cap);
        CHECK(g_thrown == 0 && total > 0 && total <= cap);
        int home = 0, perfect = 0;
        for (int r = 0; r < nreads; r++) {
            CHECK(mn[(size_t)r] >= 0 && mo[(size_t)r] >= 0 && mo[(size_t)r] + mn[(size_t)r] <= total);
            if (mn[(size_t)r] == 0) continue;
            const bbmap_msite &t = ms[(size_t)mo[(size_t)r]];                       // lists are sorted: the first site is the best
            home += t.chrom == 1 && t.strand == 0 && t.start == origin[(size_t)r] && t.stop == origin[(size_t)r] + L - 1;
            perfect += t.perfect != 0;
        }
        CHECK(home >= nreads * 97 / 100 && perfect > nreads / 20);
        // BBMapHIP.getFinal: the final records (what BBMap prints) and their match strings; == bbmap_get_final, and consistent with the reads
        {
            std::vector<bbmap_final> fin((size_t)nreads), fin2((size_t)nreads);
            std::vector<uint8_t> mstr((size_t)nreads * 256), mstr2((size_t)nreads * 256);
            _jobject bF{fin.data(), (int)(fin.size() * sizeof(bbmap_final)), 1}, bM{mstr.data(), (int)mstr.size(), 1};
            const jlong mbytes = Java_align2_BBMapHIP_getFinal(env, nullptr, mp, nreads, &bF, &bM, (jint)mstr.size());
            CHECK(g_thrown == 0 && mbytes > 0 && mbytes <= (jlong)mstr.size());
            int64_t mbytes2 = 0;
            CHECK(bbmap_get_final((bbmap_ctx *)(intptr_t)mp, nreads, fin2.data(), mstr2.data(), (int64_t)mstr2.size(), &mbytes2) == BBMAP_OK);
            CHECK(mbytes2 == mbytes && memcmp(fin.data(), fin2.data(), fin.size() * sizeof(bbmap_final)) == 0 && mstr == mstr2);
            int fhome = 0;
            for (int r = 0; r < nreads; r++) {
                const bbmap_final &f = fin[(size_t)r];
                if (!f.mapped) { CHECK(f.match_len == 0); continue; }
                CHECK(f.match_len >= L && f.match_off >= 0 && f.match_off + f.match_len <= mbytes);
                int rd = 0, rf = 0;
                for (int q = 0; q < f.match_len; q++) { const uint8_t c = mstr[(size_t)(f.match_off + q)]; rd += c != 'D'; rf += c != 'I'; }
                CHECK(rd == L && rf == f.stop - f.start + 1);                            // the string consumes the read and spans [start, stop]
                fhome += f.chrom == 1 && f.strand == 0 && f.start == origin[(size_t)r] && f.stop == origin[(size_t)r] + L - 1;
            }
            CHECK(fhome >= nreads * 97 / 100);
            // a records buffer that is too small is refused with an exception
            _jobject tiny{fin.data(), 64, 1};
            Java_align2_BBMapHIP_getFinal(env, nullptr, mp, nreads, &tiny, &bM, (jint)mstr.size());
            CHECK(g_thrown == 1);
            g_thrown = 0;
        }
        // the same batch straight through the C ABI gives the same records
        std::vector<bbmap_msite> ms2((size_t)cap);
        std::vector<int32_t> mn2((size_t)nreads, -9);
        std::vector<int64_t> mo2((size_t)nreads + 1, -9);
        int64_t total2 = 0;
        CHECK(bbmap_map_batch((bbmap_ctx *)(intptr_t)mp, nreads, recs.data(), bases.data(), (int64_t)bases.size(), bscores.data(), keyinfo.data(),
                              (int64_t)keyinfo.size(), mn2.data(), mo2.data(), ms2.data(), cap, &total2) == BBMAP_OK);
        CHECK(total2 == total && mn == mn2 && mo == mo2);
        for (int64_t i = 0; i < total; i++)          // (match_job is an index into the fill log, whose order is the device's: not compared)
            CHECK(memcmp(&ms[(size_t)i], &ms2[(size_t)i], offsetof(bbmap_msite, match_job)) == 0);
        // an error surfaces as an exception and its text is readable through lastError
        const jlong bad = Java_align2_BBMapHIP_mapBatch(env, nullptr, mp, nreads + 1, &bR, &bB, &bS, (jint)bases.size(), &bK, (jint)keyinfo.size(), &bMN, &bMO, &bMS, cap);
        CHECK(bad == 0 && g_thrown == 1);
        g_thrown = 0;
        Java_align2_BBMapHIP_destroy(env, nullptr, mp);
        Java_align2_BBIndexHIP_destroy(env, nullptr, ix);
        printf("mock JNI glue: %d of %d reads mapped to their origin (%d perfect), %lld site records\n", home, nreads, perfect, (long long)total);
    }
    CHECK(g_violations == 0 && g_critical == 0 && g_thrown == 0);
    printf("mock JNI glue: alignBatch, findBatch and mapBatch through the JNI layer equal the C ABI; direct buffers only, no critical region\n");
    return 0;
}

int main(int argc, char **argv) {
    JNINativeInterface_ tbl;
    memset(&tbl, 0, sizeof tbl);
    tbl.FindClass = mFindClass; tbl.ThrowNew = mThrowNew; tbl.GetArrayLength = mGetArrayLength;
    tbl.GetByteArrayRegion = mGetByte; tbl.GetIntArrayRegion = mGetInt; tbl.GetLongArrayRegion = mGetLong; tbl.GetFloatArrayRegion = mGetFloat;
    tbl.SetIntArrayRegion = mSetInt; tbl.SetLongArrayRegion = mSetLong; tbl.SetFloatArrayRegion = mSetFloat;
    tbl.GetPrimitiveArrayCritical = mGetCritical; tbl.ReleasePrimitiveArrayCritical = mReleaseCritical;
    tbl.GetObjectArrayElement = mGetObjectArrayElement; tbl.DeleteLocalRef = mDeleteLocalRef; tbl.SetByteArrayRegion = mSetByte;
    tbl.GetDirectBufferAddress = mGetDirectBufferAddress; tbl.GetDirectBufferCapacity = mGetDirectBufferCapacity;
    JNIEnv_ env; env.functions = &tbl;
    if (argc > 1 && !strcmp(argv[1], "gpu")) return test_gpu(&env);
    if (argc > 1 && !strcmp(argv[1], "glue")) return test_glue(&env);
    if (argc > 1 && !strcmp(argv[1], "threads")) return test_threads(&tbl, argc > 2 ? atoi(argv[2]) : 32);
    return test_merge(&env);
}
