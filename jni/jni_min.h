/*
 * jni_min.h -- the handful of JNI declarations the shim needs, written from the Java Native Interface Specification
 * (chapter 4, "JNI Functions": the interface function table and its fixed slot numbers).  Used ONLY when no JDK header is
 * installed (this image has none), so that the shim is compiled and its logic exercised against a mock JNIEnv
 * (jni/mock_jni_test.cpp); with a JDK present, <jni.h> is used instead (see bbtoolsjni_shim.cpp).  Only the slots the shim
 * calls carry a prototype; the others are padding that keeps every used slot at its specified index (checked below).
 */
#ifndef BBMAP_AMD_JNI_MIN_H
#define BBMAP_AMD_JNI_MIN_H
#include <stddef.h>
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jfloatArray;
typedef jarray jobjectArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

/* C++: JNIEnv is a struct whose first member points to the function table; C: JNIEnv is that pointer (as in <jni.h>) */
#ifdef __cplusplus
struct JNIEnv_;
typedef JNIEnv_ JNIEnv;
#else
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
#endif

struct JNINativeInterface_ {
    void *slots_0_5[6];
    jclass (*FindClass)(JNIEnv *, const char *);                                        /* 6 */
    void *slots_7_13[7];
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);                                   /* 14 */
    void *slots_15_22[8];
    void (*DeleteLocalRef)(JNIEnv *, jobject);                                          /* 23 */
    void *slots_24_170[147];
    jsize (*GetArrayLength)(JNIEnv *, jarray);                                          /* 171 */
    void *slot_172;
    jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);                    /* 173 */
    void *slots_174_199[26];
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);            /* 200 */
    void *slots_201_202[2];
    void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);               /* 203 */
    void (*GetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, jlong *);            /* 204 */
    void (*GetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, jfloat *);         /* 205 */
    void *slots_206_207[2];
    void (*SetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, const jbyte *);      /* 208 */
    void *slots_209_210[2];
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);         /* 211 */
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);      /* 212 */
    void (*SetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, const jfloat *);   /* 213 */
    void *slots_214_221[8];
    void *(*GetPrimitiveArrayCritical)(JNIEnv *, jarray, jboolean *);                   /* 222 */
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *, jarray, void *, jint);              /* 223 */
    void *slots_224_229[6];
    void *(*GetDirectBufferAddress)(JNIEnv *, jobject);                                 /* 230 */
    jlong (*GetDirectBufferCapacity)(JNIEnv *, jobject);                                /* 231 */
    void *slot_232;
};

#ifdef __cplusplus
struct JNIEnv_ {
    const JNINativeInterface_ *functions;
    jclass FindClass(const char *n) { return functions->FindClass(this, n); }
    jint ThrowNew(jclass c, const char *m) { return functions->ThrowNew(this, c, m); }
    jsize GetArrayLength(jarray a) { return functions->GetArrayLength(this, a); }
    void GetByteArrayRegion(jbyteArray a, jsize s, jsize l, jbyte *b) { functions->GetByteArrayRegion(this, a, s, l, b); }
    void GetIntArrayRegion(jintArray a, jsize s, jsize l, jint *b) { functions->GetIntArrayRegion(this, a, s, l, b); }
    void GetLongArrayRegion(jlongArray a, jsize s, jsize l, jlong *b) { functions->GetLongArrayRegion(this, a, s, l, b); }
    void GetFloatArrayRegion(jfloatArray a, jsize s, jsize l, jfloat *b) { functions->GetFloatArrayRegion(this, a, s, l, b); }
    void SetIntArrayRegion(jintArray a, jsize s, jsize l, const jint *b) { functions->SetIntArrayRegion(this, a, s, l, b); }
    void SetLongArrayRegion(jlongArray a, jsize s, jsize l, const jlong *b) { functions->SetLongArrayRegion(this, a, s, l, b); }
    void SetFloatArrayRegion(jfloatArray a, jsize s, jsize l, const jfloat *b) { functions->SetFloatArrayRegion(this, a, s, l, b); }
    void *GetPrimitiveArrayCritical(jarray a, jboolean *c) { return functions->GetPrimitiveArrayCritical(this, a, c); }
    void ReleasePrimitiveArrayCritical(jarray a, void *p, jint m) { functions->ReleasePrimitiveArrayCritical(this, a, p, m); }
    void DeleteLocalRef(jobject o) { functions->DeleteLocalRef(this, o); }
    void SetByteArrayRegion(jbyteArray a, jsize s, jsize l, const jbyte *b) { functions->SetByteArrayRegion(this, a, s, l, b); }
    jobject GetObjectArrayElement(jobjectArray a, jsize i) { return functions->GetObjectArrayElement(this, a, i); }
    void *GetDirectBufferAddress(jobject b) { return functions->GetDirectBufferAddress(this, b); }
    jlong GetDirectBufferCapacity(jobject b) { return functions->GetDirectBufferCapacity(this, b); }
};
#define BBJNI_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define BBJNI_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, FindClass) == 6 * sizeof(void *), "FindClass is slot 6");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, ThrowNew) == 14 * sizeof(void *), "ThrowNew is slot 14");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetArrayLength) == 171 * sizeof(void *), "GetArrayLength is slot 171");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetByteArrayRegion) == 200 * sizeof(void *), "GetByteArrayRegion is slot 200");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetIntArrayRegion) == 203 * sizeof(void *), "GetIntArrayRegion is slot 203");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, SetIntArrayRegion) == 211 * sizeof(void *), "SetIntArrayRegion is slot 211");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, SetLongArrayRegion) == 212 * sizeof(void *), "SetLongArrayRegion is slot 212");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetPrimitiveArrayCritical) == 222 * sizeof(void *), "GetPrimitiveArrayCritical is slot 222");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, DeleteLocalRef) == 23 * sizeof(void *), "DeleteLocalRef is slot 23");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, SetByteArrayRegion) == 208 * sizeof(void *), "SetByteArrayRegion is slot 208");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetObjectArrayElement) == 173 * sizeof(void *), "GetObjectArrayElement is slot 173");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetDirectBufferAddress) == 230 * sizeof(void *), "GetDirectBufferAddress is slot 230");
BBJNI_STATIC_ASSERT(offsetof(struct JNINativeInterface_, GetDirectBufferCapacity) == 231 * sizeof(void *), "GetDirectBufferCapacity is slot 231");
BBJNI_STATIC_ASSERT(sizeof(struct JNINativeInterface_) == 233 * sizeof(void *), "the table has 233 slots (JNI 1.8)");
#endif
