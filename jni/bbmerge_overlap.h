/* BBMerge's overlap natives (host code): plain-C entry points behind the three Java_jgi_BBMergeOverlapper_* symbols.
 * They restate jni/BBMergeOverlapper.c (mateByOverlap :24-125, findBestRatio :127-179, findBestRatio_WithQualities :182-228,
 * mateByOverlapRatio_WithQualities :230-319, mateByOverlapRatio :321-402).  BBMerge is a different tool from the mapper (SURVEY.md
 * 8f N4): nothing here runs on the GPU; the functions exist so that a replacement libbbtoolsjni.so resolves every symbol the
 * reference's library exports.  rvector[2] = best "bad" count, rvector[4] = ambiguous flag, as the reference sets them.
 * aprob / bprob are caller scratch of at least max(alen, blen) floats: mateByOverlap reads bprob[j] with j indexing read a
 * (jni/BBMergeOverlapper.c:70), so entries past blen are whatever the caller left there -- as in the reference. */
#ifndef BBMAP_AMD_BBMERGE_OVERLAP_H
#define BBMAP_AMD_BBMERGE_OVERLAP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int32_t bbmerge_mate_by_overlap(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, const int8_t *aqual, const int8_t *bqual,
                                float *aprob, float *bprob, int32_t *rvector, int32_t minOverlap0, int32_t minOverlap, int32_t minInsert0,
                                int32_t margin, int32_t maxMismatches0, int32_t maxMismatches, int32_t minq);
int32_t bbmerge_mate_by_overlap_ratio(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, int32_t *rvector, int32_t minOverlap0,
                                      int32_t minOverlap, int32_t minInsert0, int32_t minInsert, float maxRatio, float margin, float offset,
                                      float gIncr, float bIncr);
int32_t bbmerge_mate_by_overlap_ratio_with_qualities(const int8_t *a, int32_t alen, const int8_t *b, int32_t blen, const int8_t *aqual,
                                                     const int8_t *bqual, float *aprob, float *bprob, int32_t *rvector, int32_t minOverlap0,
                                                     int32_t minOverlap, int32_t minInsert0, int32_t minInsert, float maxRatio, float margin,
                                                     float offset);
#ifdef __cplusplus
}
#endif
#endif
