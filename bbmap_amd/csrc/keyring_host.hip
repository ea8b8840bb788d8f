// Host side of the probe's inputs: AbstractMapThread.quickMap up to the findAdvanced call (current/align2/AbstractMapThread.java:642-728).
//
// In the reference this is per-read Java on the mapping thread, float arithmetic on the read's qualities: key error probabilities
// (QualityTools.makeKeyProbs), key placement (KeyRing.makeOffsets3), key scores (QualityTools.makeKeyScores) and base scores
// (QualityTools.makeByteScoreArray).  Its integer outputs -- offsets[K], keyScoresP[K], baseScoresP[L] -- are what the device probe
// takes (bbidx_read + keyinfo + baseScores, include/bbmap_amd.h), so it stays on the host here too.
// Java float semantics: every operation below is a single-precision IEEE operation (the library is built with -ffp-contract=off
// -fno-fast-math); Math.round(float) is floor(x + 0.5) evaluated exactly, Math.ceil / Math.pow / Math.log10 work on doubles.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "bbmap_amd.h"

void bbmap_set_error(const char *msg);

namespace {

struct QualTables {
    float probError[128], probCorrect[128], probCorrectInverse[128];
    QualTables() {                                      // QualityTools.java:475-480, :519-539
        for (int i = 0; i < 128; i++) probError[i] = (float)pow(10.0, 0 - .1 * i);
        probError[0] = .8f;
        for (int i = 0; i < 128; i++) { probCorrect[i] = 1 - probError[i]; probCorrectInverse[i] = 1 / probCorrect[i]; }
    }
};
const QualTables &tables() { static const QualTables t; return t; }

inline int java_round(float f) { return (int)floor((double)f + 0.5); }          // Math.round(float)
inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }
inline bool fully_defined(int b) { const int u = b & ~32; return b < 128 && (u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'U'); }

// QualityTools.makeKeyProbs(quality, bases, keylen, out, useModulo=false) :188-247 / :250-279
void make_key_probs(const uint8_t *quality, int len, int keylen, float *out) {
    const int n = len - keylen + 1;
    if (!quality) { for (int i = 0; i < n; i++) out[i] = 0; return; }
    const QualTables &T = tables();
    float key1 = 1;
    int timeSinceZero = 0;
    for (int i = 0; i < keylen; i++) {
        const int q = quality[i] & 127;
        if (q > 0) timeSinceZero++; else timeSinceZero = 0;
        key1 *= T.probCorrect[q];
    }
    out[0] = 1 - key1;
    if (timeSinceZero < keylen) out[0] = 1;
    for (int a = 0, b = keylen; b < len; a++, b++) {
        const int qa = quality[a] & 127, qb = quality[b] & 127;
        if (qb > 0) timeSinceZero++; else timeSinceZero = 0;
        key1 = key1 * T.probCorrectInverse[qa] * T.probCorrect[qb];
        out[a + 1] = 1 - key1;
        if (timeSinceZero < keylen) out[a + 1] = 1;
    }
}

// KeyRing.desiredKeysFromDensity :269-282
int desired_keys_from_density(int readlen, int blocksize, float density, int minKeysDesired) {
    const int slots = readlen - blocksize + 1;
    int desired = (int)ceil((double)((readlen * density) / blocksize));
    desired = imax(minKeysDesired, desired);
    return imin(slots, desired);
}

// KeyRing.makeOffsets3 :396-506 (KEEP_BAD_KEYS = false); returns the number of offsets written, 0 for null
int make_offsets3(const float *keyErrorProb, int readlenOriginal, int blocksize, float density, float maxDensity, int minKeysDesired,
                  bool semiperfectmode, std::vector<int> &offsets) {
    int readlen = readlenOriginal;
    const int maxProbIndex = readlen - blocksize;
    int left = 0, right = maxProbIndex;
    const float errorLimit2 = 0.9999f, errorLimit1 = semiperfectmode ? 0.99f : 0.94f;
    while (left <= right && keyErrorProb[left] >= errorLimit1) left++;
    while (right >= left && keyErrorProb[right] >= errorLimit1) right--;
    int potentialKeys = 0;
    for (int i = left; i <= right; i++) if (keyErrorProb[i] < errorLimit2) potentialKeys++;
    if (potentialKeys == 0) return 0;
    if (right < left) return 0;
    readlen = right - left + blocksize;
    if (readlen < blocksize) return 0;
    int desiredKeys = desired_keys_from_density(readlenOriginal, blocksize, density, minKeysDesired);
    if (readlen < readlenOriginal) desiredKeys = imin(desiredKeys, desired_keys_from_density(readlen, blocksize, maxDensity, minKeysDesired));
    desiredKeys = imin(desiredKeys, potentialKeys);
    offsets.assign((size_t)desiredKeys, 0);
    const float interval = (right - left) / (float)imax(desiredKeys - 1, 1);
    const int intervalInt = ((int)interval) + 1;
    float f = (float)left;
    int prev = -1, misses = 0;
    for (int i = 0, j = left; i < desiredKeys; i++) {
        int x = -1;
        if (prev < j) {
            if (keyErrorProb[j] < errorLimit2 && (prev < 0 || j - prev > 0)) x = j;
            else {
                for (int k = j - 1, lim = prev + 2; k > lim; k--) if (keyErrorProb[k] < errorLimit2) { x = k; break; }
                if (x < 0) for (int k = j + 1, lim = imin(j + intervalInt, right); k < lim; k++) if (keyErrorProb[k] < errorLimit2) { x = k; break; }
            }
        }
        offsets[(size_t)i] = x;
        if (x > -1) prev = x;
        else { misses++; prev = imax(prev, j - 2); }
        f += interval;
        j = imin(maxProbIndex, imax(j + 1, java_round(f)));
    }
    if (misses > 0) {
        size_t m = 0;
        for (size_t i = 0; i < offsets.size(); i++) if (offsets[i] >= 0) offsets[m++] = offsets[i];
        offsets.resize(m);
    }
    return (int)offsets.size();
}

// Read.avgQualityByProbability(false, 0) (current/stream/Read.java:1738-1745, expectedErrors :2115-2132, QualityTools.java:497-517)
int avg_quality_by_probability(const uint8_t *bases, const uint8_t *quality, int len) {
    if (len == 0) return 0;
    const QualTables &T = tables();
    float sum = 0;
    for (int i = 0; i < len; i++) if (fully_defined(bases[i])) sum += T.probError[quality[i] & 127];
    const float p = sum / len;
    const double prob = 1 - (double)(1 - p);
    double phred;
    if (prob >= 1) phred = 0; else if (prob <= 0.000001) phred = 60; else phred = -10 * log10(prob);
    const long q = (long)floor(phred + 0.5);
    return (int)(q < 0 ? 0 : (q > 41 ? 41 : q));          // Read.MAX_CALLED_QUALITY = 41; only `< 2` is ever asked of this value here
}

}  // namespace

extern "C" int bbkeys_default_config(int32_t profile, bbkeys_config *cfg) {
    if (!cfg || (profile != BBIDX_PROFILE_BBMAP && profile != BBIDX_PROFILE_PACBIO)) { bbmap_set_error("bbkeys_default_config: bad argument"); return BBMAP_E_ARG; }
    memset(cfg, 0, sizeof *cfg);
    if (profile == BBIDX_PROFILE_PACBIO) {     // BBMapPacBio.setDefaults, current/align2/BBMapPacBio.java:51-58
        cfg->k = 12; cfg->keyDensity = 3.5f; cfg->maxKeyDensity = 4.5f; cfg->minKeyDensity = 2.8f; cfg->maxDesiredKeys = 63;
    } else {                                   // BBMap.setDefaults, current/align2/BBMap.java:48-55
        cfg->k = 13; cfg->keyDensity = 1.9f; cfg->maxKeyDensity = 3.0f; cfg->minKeyDensity = 1.5f; cfg->maxDesiredKeys = 15;
    }
    cfg->minApproxHitsToKeep = 1;
    return BBMAP_OK;
}

extern "C" int bbkeys_make(const bbkeys_config *cfg, const uint8_t *bases, const uint8_t *quality, int32_t len,
                           int32_t *offsets, int32_t *keyScores, int32_t cap, int8_t *baseScores) {
    if (!cfg || !bases || len < 0 || !offsets || !keyScores || !baseScores || cfg->k < 1) { bbmap_set_error("bbkeys_make: bad argument"); return BBMAP_E_ARG; }
    const int K = cfg->k;
    // makeByteScoreArray(quality, 100, out, negative=true) :145-181 -- written in every case (the probe reads it for any read it is given)
    {
        const QualTables &T = tables();
        for (int i = 0; i < len; i++) baseScores[i] = quality ? (int8_t)(java_round(100 * T.probCorrect[quality[i] & 127]) - 100) : (int8_t)0;
    }
    if (len < K) return 0;                                                                  // :645
    {   // :650-654: `if(PERFECTMODE || SEMIPERFECTMODE){if(r.containsUndefined()){return -1;}}else if(DISCARD_MOSTLY_UNDEFINED_READS){...}`
        int n = 0;
        for (int i = 0; i < len; i++) if (!fully_defined(bases[i])) n++;
        if (cfg->semiperfectMode ? n > 0 : (n > 25 && len - n < n)) return 0;
    }
    const int keyProbLen = len - K + 1;
    std::vector<float> keyProbs((size_t)keyProbLen);
    float keyDen2 = ((cfg->maxDesiredKeys * K) / (float)len);                               // :663-665
    keyDen2 = keyDen2 > cfg->minKeyDensity ? keyDen2 : cfg->minKeyDensity;
    { float m = cfg->keyDensity < keyDen2 ? cfg->keyDensity : keyDen2; keyDen2 = m < (float)K ? m : (float)K; }
    float keyDen3;                                                                          // :667-676
    if (len <= 50) keyDen3 = cfg->maxKeyDensity;
    else if (len >= 200) keyDen3 = cfg->maxKeyDensity - 0.5f;
    else keyDen3 = cfg->maxKeyDensity - 0.003333333333f * (len - 50);
    keyDen3 = keyDen3 > cfg->keyDensity ? keyDen3 : cfg->keyDensity;
    keyDen3 = keyDen3 < (float)K ? keyDen3 : (float)K;
    make_key_probs(quality, len, K, keyProbs.data());
    std::vector<int> offs;
    const int n = make_offsets3(keyProbs.data(), len, K, keyDen2, keyDen3, 2, cfg->semiperfectMode != 0, offs);
    if (n == 0 || n < cfg->minApproxHitsToKeep) return 0;                                  // :701
    if (quality && avg_quality_by_probability(bases, quality, len) < 2) return 0;
    if (n > cap) { bbmap_set_error("bbkeys_make: more keys than the caller's buffers hold"); return BBMAP_E_ARG; }
    // makeKeyScores(keyProbs, keyProbLen, range, baseKeyScore, keyScoresAll) :712-724, QualityTools.java:125-133
    const int a = 100 * K, baseKeyScore = a / 8, range = a - baseKeyScore;                  // BASE_KEY_HIT_SCORE = BASE_HIT_SCORE * KEYLEN
    float probAllErrors = 1.0f;
    for (int i = 0; i < n; i++) {
        const float p = keyProbs[(size_t)offs[(size_t)i]];
        offsets[i] = offs[(size_t)i];
        keyScores[i] = baseKeyScore + java_round(range * (1 - p));
        probAllErrors *= p;
    }
    if (probAllErrors > 0.50f) return 0;
    return n;
}

// The batch form: fills read records, keyinfo and base scores for reads laid out back to back in `bases` (read i occupies
// bases_off[i] .. + len[i]; qualities, when given, at the same offsets).  A read quickMap would refuse gets nkeys = 0 (the probe and
// the mapper report it without sites, as quickMap's -1 leaves r.sites null).
extern "C" int bbkeys_make_batch(const bbkeys_config *cfg, int64_t n_reads, const int64_t *bases_off, const int32_t *lens,
                                 const uint8_t *bases, const uint8_t *quality, bbidx_read *reads, int32_t *keyinfo, int64_t keyinfo_cap,
                                 int8_t *baseScores, int64_t *keyinfo_used) {
    if (!cfg || !bases_off || !lens || !bases || !reads || !keyinfo || !baseScores || n_reads < 0) { bbmap_set_error("bbkeys_make_batch: bad argument"); return BBMAP_E_ARG; }
    int64_t used = 0;
    std::vector<int32_t> o, s;
    for (int64_t i = 0; i < n_reads; i++) {
        const int len = lens[i];
        const int cap = len > 0 ? len : 1;
        o.resize((size_t)cap); s.resize((size_t)cap);
        const int n = bbkeys_make(cfg, bases + bases_off[i], quality ? quality + bases_off[i] : nullptr, len, o.data(), s.data(), cap, baseScores + bases_off[i]);
        if (n < 0) return n;
        if (used + 2 * (int64_t)n > keyinfo_cap) { bbmap_set_error("bbkeys_make_batch: keyinfo buffer too small"); return BBMAP_E_ARG; }
        reads[i].bases_off = bases_off[i]; reads[i].keys_off = used; reads[i].len = len; reads[i].nkeys = n;
        memcpy(keyinfo + used, o.data(), sizeof(int32_t) * (size_t)n);
        memcpy(keyinfo + used + n, s.data(), sizeof(int32_t) * (size_t)n);
        used += 2 * (int64_t)n;
    }
    if (keyinfo_used) *keyinfo_used = used;
    return BBMAP_OK;
}
