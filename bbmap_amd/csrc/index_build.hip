// k-mer index construction on the device: align2.IndexMaker4 (count -> prefix sum -> fill,
// current/align2/IndexMaker4.java:303-421) and BBIndex.analyzeIndex (current/align2/BBIndex.java:101-191) for gfx950.
//
// Every step is a bandwidth-bound pass over the genome or over the 4^k key space, so the whole build runs in HBM and
// nothing but the length histogram (a few thousand counters) comes back to the host:
//   1. emit: one thread per genome position packs its k bases into a key (invalid when a base is undefined or the
//      k-mer is banned: period <= 2, IndexMaker4.java:327-339);
//   2. a stable LSD radix sort (rocPRIM through hipCUB) of (key, site) pairs on the low 2k bits keeps genome order
//      inside every list, which is the order IndexMaker4's one-thread-per-list fill produces;
//   3. Block.starts are the run boundaries of the sorted keys (no counters, no prefix sum);
//   4. COUNTS[key] = len(key) + len(rc(key)) over all blocks (palindromes once), "clumpy" keys zeroed
//      (BBIndex.java:125-153);
//   5. the site-weighted length histogram (Tools.makeLengthHistogram3, current/align2/Tools.java:1797-1850) is
//      finished on the host from the device-side bincount, and the probe tunables follow BBMap.loadIndex's genome-size
//      rules (current/align2/BBMap.java:367-381) and analyzeIndex's thresholds (BBIndex.java:176-190).
// The result is an ordinary bbidx_ctx (same arrays, same fused key table) ready for bbidx_find_batch.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "bbmap_amd.h"
#include "index_common.h"
#include "index_ctx.h"

void bbmap_set_error(const char *msg);

namespace bbidxb {
using namespace bbidx;

// one thread per position of one chromosome: key (or 0xFFFFFFFF) and encoded site.  The block's 256 + k - 1 bases are turned into
// 2-bit codes once, in LDS (one coalesced load per base instead of k loads per position).  No per-key counter is bumped here: list
// lengths come out of the sort (starts_from_sorted_kernel) -- 770 M atomics on a 268 MB table per block were a third of the build.
__global__ __launch_bounds__(256) void emit_kernel(const uint8_t *chrom, int len, int k, int chromNumber, int shift, int lowMask,
                                                   unsigned *keys, int *sites, unsigned long long *defined) {
    __shared__ int8_t code[256 + 16];
    const int npos = len - k;
    unsigned long long def = 0;
    // grid-stride over 256-position tiles: the count of defined bases stays in a register and costs one atomic per wave at the very
    // end (one per wave and tile -- 12 M same-address atomics per block of the hg38 shape -- was 80 % of the whole build)
    for (long long base = (long long)blockIdx.x * 256; base < len; base += (long long)gridDim.x * 256) {
        __syncthreads();
        for (int i = threadIdx.x; i < 256 + k - 1; i += 256) code[i] = base + i < len ? (int8_t)base_num(chrom[base + i]) : (int8_t)-1;
        __syncthreads();
        const long long a = base + threadIdx.x;
        def += (a < len && code[threadIdx.x] >= 0) ? 1ull : 0ull;
        if (a >= npos) continue;
        unsigned key = 0;
        bool valid = true;
        for (int j = 0; j < k; j++) {
            const int x = code[threadIdx.x + j];
            valid = valid && x >= 0;
            key = (key << 2) | (unsigned)(x & 3);
        }
        if (valid) {
            const unsigned banmask = (1u << (2 * k - 4)) - 1u;
            if ((key >> 4) == (key & banmask)) valid = false;              // homopolymers and dinucleotide repeats
        }
        keys[a] = valid ? key : 0xFFFFFFFFu;
        sites[a] = ((chromNumber & lowMask) << shift) | (int)a;
    }
    for (int d = 32; d >= 1; d >>= 1) def += __shfl_xor(def, d, 64);
    if ((threadIdx.x & 63) == 0 && def) atomicAdd(defined, def);
}

// Block.starts from the sorted keys: starts[key] = index of the first pair whose key is >= key, starts[4^k] = number of valid pairs.
// An invalid pair (0xFFFFFFFF) sorts, on the low 2k bits, as the all-T k-mer, which is banned and owns no list: everything from the
// first invalid pair on is padding.  Thread i fills the keys in (key[i-1], key[i]]: an empty range inside a list, the keys without a
// list in front of a list's first pair; thread npos closes the table.
__global__ void starts_from_sorted_kernel(const unsigned *keysSorted, long long npos, int *starts, long long nkeys) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > npos) return;
    long long cur = nkeys, prev = -1;
    if (i < npos) { const unsigned v = keysSorted[i]; if (v != 0xFFFFFFFFu) cur = (long long)v; }
    if (i > 0) { const unsigned v = keysSorted[i - 1]; prev = v != 0xFFFFFFFFu ? (long long)v : nkeys; }
    for (long long kk = prev + 1; kk <= cur; kk++) starts[kk] = (int)i;
}
__global__ void accumulate_counts(const int *starts, unsigned long long *total, long long nkeys) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nkeys) return;
    const int n = starts[i + 1] - starts[i];
    if (n) total[i] += (unsigned long long)n;
}
// BBIndex.java:125-143: adjacent entries of one list 1..5 bases apart
__global__ void clump_kernel(const unsigned *keysSorted, const int *sitesSorted, const int *nvalidDev, int k, unsigned *clump) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= (long long)*nvalidDev) return;
    if (keysSorted[i] != keysSorted[i - 1]) return;
    const long long dif = (long long)sitesSorted[i] - (long long)sitesSorted[i - 1];
    if (dif > 0 && dif <= 5) {
        const int key = (int)keysSorted[i], rc = rc_key(key, k);
        atomicAdd(&clump[min(key, rc)], 1u);
    }
}
// COUNTS[key] = own + other (palindromes once), clamped like the int32 sum it stands for (BBIndex.java:147-153)
__global__ void combine_counts(const unsigned long long *total, int *counts, int k, long long nkeys) {
    const long long key = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    const int rc = rc_key((int)key, k);
    unsigned long long v = total[key];
    if (rc != (int)key) v += total[rc];
    counts[key] = v > 0x7fffffffull ? 0x7fffffff : (int)v;
}
__global__ void zero_clumpy(const unsigned *clump, const int *countsIn, int *countsOut, int k, long long nkeys, int clumpyMinLen, float clumpyFraction) {
    const long long key = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    const unsigned cc = clump[key];
    if (!cc) return;
    const int ln = countsIn[key];
    if (ln > clumpyMinLen && (float)cc > __fmul_rn(clumpyFraction, (float)ln)) { countsOut[key] = 0; countsOut[rc_key((int)key, k)] = 0; }
}
// grid-stride: one atomic per wave at the very end
__global__ void max_kernel(const int *v, long long n, int *out) {
    int m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}
// Histogram of COUNTS.  Almost every key of a small genome has count 0 (never needed: the histogram weights a bin by its
// value) or a tiny count, so the low bins are first accumulated per block in LDS; only long lists hit global atomics.
constexpr int BIN_LDS = 2048;
__global__ void bincount_kernel(const int *v, long long n, unsigned long long *bins, int nbins) {
    __shared__ unsigned low[BIN_LDS];
    for (int i = threadIdx.x; i < BIN_LDS; i += blockDim.x) low[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = v[i];
        if (x <= 0) continue;
        if (x < BIN_LDS) atomicAdd(&low[x], 1u); else atomicAdd(&bins[x], 1ull);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BIN_LDS && i < nbins; i += blockDim.x) if (low[i]) atomicAdd(&bins[i], (unsigned long long)low[i]);
}

}  // namespace bbidxb

static thread_local char g_berr[256];
#define BHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(g_berr, sizeof g_berr, "%s failed: %s", #expr, hipGetErrorString(e_)); bbmap_set_error(g_berr); rc = BBMAP_E_HIP; goto fail; } } while (0)

// Tools.makeLengthHistogram3/4 with its int32 wrap of counts[ptr]*ptr
static void length_histogram(const std::vector<unsigned long long> &cnt, int mx, int *hist, int buckets = 1000) {
    long long total = 0;
    for (int p = 0; p <= mx; p++) total += (long long)cnt[(size_t)p] * p;
    long long csum = 0;         // sum after consuming ptr 0..ptr-1, each term wrapped to int32 first
    int ptr = 0;
    for (int i = 0; i < buckets; i++) {
        const long long limit = ((total * i) + buckets / 2) / buckets;
        while (ptr < mx + 1 && csum < limit) {
            const unsigned long long prod = (cnt[(size_t)ptr] * (unsigned long long)ptr) & 0xFFFFFFFFull;
            csum += (long long)(int32_t)(uint32_t)prod;
            ptr++;
        }
        hist[i] = ptr - 1 > 0 ? ptr - 1 : 0;
    }
    hist[buckets] = mx;
}

// BBMap.loadIndex genome-size adjustments + analyzeIndex thresholds, as bbmap_amd/index.py:_set_params states them
// (the same arithmetic for mapPacBio with BBIndexPacBio's statics, BBMapPacBio.java:351-365, BBIndexPacBio.java:2462-2560)
static void derive_params(bbidx_params &p, long long definedBases, const int *h) {
    const bool pb = p.profile == BBIDX_PROFILE_PACBIO;
    float f = pb ? 0.005f : 0.03f;                                    // FRACTION_GENOME_TO_EXCLUDE
    p.maxIndel = pb ? 100 : 16000; p.maxIndel2 = pb ? 800 : 32000; p.minApproxHitsToKeep = 1; p.kfilter = 0;
    p.maxHitsReduction2 = pb ? 3 : 2; p.maximumMaxHitsReduction = pb ? 6 : 3; p.hitReductionDiv = pb ? 4 : 5;
    p.quitAfterTwoPerfects = 1; p.prescanQscore = 1; p.trimByGreedy = 1; p.slow = 0;
    if (definedBases < 300000000LL) {
        p.maxHitsReduction2 += 1; p.maximumMaxHitsReduction += 1;
        if (definedBases < 30000000LL) {
            f = f * 0.5f; p.maximumMaxHitsReduction += 1;
            p.hitReductionDiv = p.hitReductionDiv - 1 > 3 ? p.hitReductionDiv - 1 : 3;
        } else if (definedBases < 100000000LL) f = f * 0.6f;
        else f = f * 0.75f;
    }
    const double fd = (double)f;
    p.maxAverageListToSearch = (int)(1000 * (1 - 2.3 * fd));
    p.maxAverageListToSearch2 = (int)(1000 * (1 - 1.4 * fd));
    p.maxShortestListToSearch = (int)(1000 * (1 - 2.8 * fd));
    const int i1 = (int)((1.0f - f) * 1000.0f);
    const int i2 = (int)((1.0f - f * 0.25f) * 1000.0f);
    const int SMALL = pb ? 80 : 20;                                   // SMALL_GENOME_LIST
    p.maxUsableLength = h[i1] > 2 * SMALL ? h[i1] : 2 * SMALL;
    p.maxUsableLength2 = h[i2] > 6 * SMALL ? h[i2] : 6 * SMALL;
    const int denom = h[p.maxAverageListToSearch] > 2 * SMALL ? h[p.maxAverageListToSearch] : 2 * SMALL;
    const float q = (-50.0f * 4000.0f) / (float)denom;
    long long pps = (long long)std::floor((double)q);
    p.pointsPerSite = pps != 0 ? pps : -1;
}

extern "C" int bbidx_build(int32_t device, int32_t k, int32_t chromBits, int32_t nchroms,
                           const uint8_t *const *chromArr, const int32_t *chromArrLen, bbidx_ctx **out) {
    return bbidx_build_profile(device, BBIDX_PROFILE_BBMAP, k, chromBits, nchroms, chromArr, chromArrLen, out);
}

extern "C" int bbidx_build_profile(int32_t device, int32_t profile, int32_t k, int32_t chromBits, int32_t nchroms,
                                   const uint8_t *const *chromArr, const int32_t *chromArrLen, bbidx_ctx **out) {
    if (!out || !chromArr || !chromArrLen) { bbmap_set_error("bbidx_build: null argument"); return BBMAP_E_ARG; }
    *out = nullptr;
    if (profile != BBIDX_PROFILE_BBMAP && profile != BBIDX_PROFILE_PACBIO) { bbmap_set_error("bbidx_build: unknown profile"); return BBMAP_E_ARG; }
    if (k <= 0) k = profile == BBIDX_PROFILE_PACBIO ? 12 : 13;            // BBMapPacBio.java:51 / BBMap.java:48
    if (k < 8 || k > 15 || nchroms < 1 || chromBits > 16) { bbmap_set_error("bbidx_build: bad geometry (k must be 8..15)"); return BBMAP_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { bbmap_set_error("bbidx_build: no HIP device (no CPU path)"); return BBMAP_E_NODEVICE; }
    if (device < 0 || device >= ndev) { bbmap_set_error("bbidx_build: bad device ordinal"); return BBMAP_E_ARG; }
    int rc = BBMAP_OK;
    bbidx_ctx *c = nullptr;
    unsigned *d_keys = nullptr, *d_keys2 = nullptr, *d_clump = nullptr;
    int *d_sites2 = nullptr, *d_max = nullptr, *d_countsRaw = nullptr;
    unsigned long long *d_total = nullptr, *d_defined = nullptr, *d_bins = nullptr;
    void *d_tmp = nullptr;
    hipStream_t cs = nullptr, us = nullptr;             // kernels / uploads
    std::vector<hipEvent_t> evUp;
    const bool timers = getenv("BBIDX_BUILD_TIMERS") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double msUpload = 0;
    {
        if (hipSetDevice(device) != hipSuccess) { bbmap_set_error("bbidx_build: hipSetDevice failed"); return BBMAP_E_HIP; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { bbmap_set_error("bbidx_build: hipGetDeviceProperties failed"); return BBMAP_E_HIP; }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { bbmap_set_error("bbidx_build: this build targets gfx950 only"); return BBMAP_E_NODEVICE; }
        int maxlen = 0;
        for (int ch = 1; ch <= nchroms; ch++) { if (chromArrLen[ch] < 0 || !chromArr[ch]) { bbmap_set_error("bbidx_build: bad chromosome"); return BBMAP_E_ARG; } if (chromArrLen[ch] > maxlen) maxlen = chromArrLen[ch]; }
        if (chromBits < 0) {                                     // RefToIndex.AUTO_CHROMBITS, BBMap.java:317-321
            int bl = 0; for (unsigned v = (unsigned)maxlen; v; v >>= 1) bl++;
            chromBits = (32 - bl) - 1; if (chromBits > 16) chromBits = 16; if (chromBits < 0) chromBits = 0;
        }
        if (maxlen > (int)(0xFFFFFFFFu >> (chromBits + 1))) { bbmap_set_error("bbidx_build: a chromosome does not fit the site encoding for this chromBits"); return BBMAP_E_ARG; }
        c = new (std::nothrow) bbidx_ctx();
        if (!c) { bbmap_set_error("bbidx_build: out of memory"); return BBMAP_E_NOMEM; }
        c->device = device; c->kernelKind = BBIDX_KERNEL_AUTO; c->blocks = prop.multiProcessorCount * 8;
        c->totalSites = 0; c->maxReadLen = BBIDX_MAX_READ_LEN;
        memset(&c->dev, 0, sizeof c->dev);
        const int nblocks = (nchroms >> chromBits) + 1;
        const int cpb = 1 << chromBits, shift = 31 - chromBits, lowMask = cpb - 1;
        const long long nkeys = 1LL << (2 * k);
        const unsigned kb = (unsigned)((nkeys + 255) / 256);
        c->dev.p.k = k; c->dev.p.chromBits = chromBits; c->dev.p.minChrom = 1; c->dev.p.maxChrom = nchroms; c->dev.p.profile = profile; c->dev.p.reserved = 0;
        c->dev.nblocks = nblocks; c->dev.nchroms = nchroms;

        // Per block: positions, and the largest block sizes the work buffers (allocated once: a hipFree per block is a device-wide
        // synchronisation, and the next block's chromosomes are meant to upload while this block sorts)
        std::vector<long long> blockPos((size_t)nblocks, 0);
        long long maxPos = 0;
        for (int b = 0; b < nblocks; b++) {
            const int first = b * cpb > 1 ? b * cpb : 1, last = (b * cpb + cpb - 1) < nchroms ? (b * cpb + cpb - 1) : nchroms;
            long long npos = 0;
            for (int ch = first; ch <= last; ch++) npos += chromArrLen[ch] > k ? chromArrLen[ch] - k : 0;
            if (npos > 0x7fffffffLL - 64) { bbmap_set_error("bbidx_build: more than 2^31 - 64 positions in one block"); rc = BBMAP_E_ARG; goto fail; }   // (cursor look-ahead stays in int range)
            blockPos[(size_t)b] = npos; c->totalSites += npos;
            if (npos > maxPos) maxPos = npos;
        }
        BHIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        BHIP(hipStreamCreateWithFlags(&us, hipStreamNonBlocking));
        std::vector<const uint8_t *> hc((size_t)nchroms + 1, nullptr);
        for (int ch = 1; ch <= nchroms; ch++) {
            void *d = nullptr;
            BHIP(hipMalloc(&d, (size_t)(chromArrLen[ch] > 0 ? chromArrLen[ch] : 1)));
            c->allocs.push_back(d);
            hc[(size_t)ch] = (const uint8_t *)d;
        }
        BHIP(hipMalloc(&d_total, (size_t)nkeys * 8));
        BHIP(hipMalloc(&d_clump, (size_t)nkeys * 4));
        BHIP(hipMalloc(&d_defined, 8));
        BHIP(hipMalloc(&d_max, 4));
        BHIP(hipMemsetAsync(d_total, 0, (size_t)nkeys * 8, cs));
        BHIP(hipMemsetAsync(d_clump, 0, (size_t)nkeys * 4, cs));
        BHIP(hipMemsetAsync(d_defined, 0, 8, cs));
        BHIP(hipMemsetAsync(d_max, 0, 4, cs));
        BHIP(hipMalloc(&d_keys, (size_t)(maxPos > 0 ? maxPos : 1) * 4));
        BHIP(hipMalloc(&d_keys2, (size_t)(maxPos > 0 ? maxPos : 1) * 4));
        BHIP(hipMalloc(&d_sites2, (size_t)(maxPos > 0 ? maxPos : 1) * 4));
        size_t tmpNeed = 0;
        if (maxPos > 0) {
            BHIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpNeed, d_keys, d_keys2, d_sites2, (int *)d_sites2, (int)maxPos, 0, 2 * k, cs));
            BHIP(hipMalloc(&d_tmp, tmpNeed));
        }

        std::vector<const int *> hs((size_t)nblocks, nullptr), hsi((size_t)nblocks, nullptr);
        evUp.resize((size_t)nblocks, nullptr);
        for (int b = 0; b < nblocks; b++) {
            const int first = b * cpb > 1 ? b * cpb : 1, last = (b * cpb + cpb - 1) < nchroms ? (b * cpb + cpb - 1) : nchroms;
            const long long npos = blockPos[(size_t)b];
            // this block's chromosomes go up on the upload stream while the previous block is still being sorted on the other
            const double u0 = since();
            for (int ch = first; ch <= last; ch++)
                if (chromArrLen[ch] > 0) BHIP(hipMemcpyAsync((void *)hc[(size_t)ch], chromArr[ch], (size_t)chromArrLen[ch], hipMemcpyHostToDevice, us));
            BHIP(hipEventCreateWithFlags(&evUp[(size_t)b], hipEventDisableTiming));
            BHIP(hipEventRecord(evUp[(size_t)b], us));
            msUpload += since() - u0;
            BHIP(hipStreamWaitEvent(cs, evUp[(size_t)b], 0));
            int *d_sites = nullptr, *d_starts = nullptr;
            BHIP(hipMalloc(&d_starts, (size_t)(nkeys + 1) * 4)); c->allocs.push_back(d_starts);
            BHIP(hipMalloc(&d_sites, (size_t)(npos > 0 ? npos : 1) * 4)); c->allocs.push_back(d_sites);
            long long off = 0;
            for (int ch = first; ch <= last; ch++) {
                const int len = chromArrLen[ch];
                if (len <= 0) continue;
                unsigned eb = (unsigned)((len + 255) / 256);
                if (eb > 8192u) eb = 8192u;
                hipLaunchKernelGGL(bbidxb::emit_kernel, dim3(eb), dim3(256), 0, cs, hc[(size_t)ch], len, k, ch, shift, lowMask,
                                   d_keys + off, d_sites2 + off, d_defined);
                BHIP(hipGetLastError());
                off += len > k ? len - k : 0;
            }
            if (npos > 0) {
                size_t need = tmpNeed;
                BHIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, need, d_keys, d_keys2, d_sites2, d_sites, (int)npos, 0, 2 * k, cs));
            }
            hipLaunchKernelGGL(bbidxb::starts_from_sorted_kernel, dim3((unsigned)((npos + 1 + 255) / 256)), dim3(256), 0, cs, d_keys2, npos, d_starts, nkeys);
            hipLaunchKernelGGL(bbidxb::accumulate_counts, dim3(kb), dim3(256), 0, cs, d_starts, d_total, nkeys);
            if (npos > 1)
                hipLaunchKernelGGL(bbidxb::clump_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, cs,
                                   d_keys2, d_sites, d_starts + nkeys, k, d_clump);
            BHIP(hipGetLastError());
            hs[(size_t)b] = d_starts; hsi[(size_t)b] = d_sites;
        }
        // COUNTS, clumpy keys, length histogram
        int *d_counts = nullptr;
        BHIP(hipMalloc(&d_counts, (size_t)nkeys * 4)); c->allocs.push_back(d_counts);
        BHIP(hipMalloc(&d_countsRaw, (size_t)nkeys * 4));
        hipLaunchKernelGGL(bbidxb::combine_counts, dim3(kb), dim3(256), 0, cs, d_total, d_countsRaw, k, nkeys);
        BHIP(hipMemcpyAsync(d_counts, d_countsRaw, (size_t)nkeys * 4, hipMemcpyDeviceToDevice, cs));
        hipLaunchKernelGGL(bbidxb::zero_clumpy, dim3(kb), dim3(256), 0, cs, d_clump, d_countsRaw, d_counts, k, nkeys,
                           profile == BBIDX_PROFILE_PACBIO ? 2800 : 2000, profile == BBIDX_PROFILE_PACBIO ? 0.8f : 0.75f);   // CLUMPY_MIN_LENGTH_INDEX, CLUMPY_FRACTION
        hipLaunchKernelGGL(bbidxb::max_kernel, dim3(2048), dim3(256), 0, cs, d_counts, nkeys, d_max);
        BHIP(hipGetLastError());
        int mx = 0;
        BHIP(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, cs));
        BHIP(hipStreamSynchronize(cs));
        const double msLists = since();
        BHIP(hipMalloc(&d_bins, (size_t)(mx + 1) * 8));
        BHIP(hipMemsetAsync(d_bins, 0, (size_t)(mx + 1) * 8, cs));
        hipLaunchKernelGGL(bbidxb::bincount_kernel, dim3(2048), dim3(256), 0, cs, d_counts, nkeys, d_bins, mx + 1);
        BHIP(hipGetLastError());
        std::vector<unsigned long long> bins((size_t)mx + 1);
        BHIP(hipMemcpyAsync(bins.data(), d_bins, (size_t)(mx + 1) * 8, hipMemcpyDeviceToHost, cs));
        unsigned long long defined = 0;
        BHIP(hipMemcpyAsync(&defined, d_defined, 8, hipMemcpyDeviceToHost, cs));
        BHIP(hipStreamSynchronize(cs));
        int hist[1001];
        length_histogram(bins, mx, hist);
        derive_params(c->dev.p, (long long)defined, hist);
        c->dev.counts = d_counts;

        auto up = [&](const void *host, size_t bytes, const void **dev) -> int {
            void *d = nullptr;
            if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return BBMAP_E_HIP;
            c->allocs.push_back(d);
            if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return BBMAP_E_HIP;
            *dev = d;
            return BBMAP_OK;
        };
        std::vector<int> clen((size_t)nchroms + 1, 0);
        for (int ch = 1; ch <= nchroms; ch++) clen[(size_t)ch] = chromArrLen[ch];
        if (up(hist, sizeof hist, (const void **)&c->dev.lengthHistogram) != BBMAP_OK ||
            up(hs.data(), hs.size() * sizeof(void *), (const void **)&c->dev.starts) != BBMAP_OK ||
            up(hsi.data(), hsi.size() * sizeof(void *), (const void **)&c->dev.sites) != BBMAP_OK ||
            up(hc.data(), hc.size() * sizeof(void *), (const void **)&c->dev.chromArr) != BBMAP_OK ||
            up(clen.data(), clen.size() * 4, (const void **)&c->dev.chromArrLen) != BBMAP_OK ||
            up(clen.data(), clen.size() * 4, (const void **)&c->dev.chromLengths) != BBMAP_OK) {
            bbmap_set_error("bbidx_build: device allocation failed"); rc = BBMAP_E_HIP; goto fail;
        }
        const double msStats = since();
        rc = bbidx_finish_create(c, hs, hsi);
        if (rc != BBMAP_OK) goto fail;
        if (timers) fprintf(stderr, "bbidx_build: lists of %d block(s) %.1f ms (of which the host spent %.1f ms handing chromosomes to the upload stream), "
                                    "COUNTS / histogram %.1f ms, fused key tables %.1f ms\n", nblocks, msLists, msUpload, msStats - msLists, since() - msStats);
    }
    for (hipEvent_t e : evUp) if (e) (void)hipEventDestroy(e);
    if (cs) (void)hipStreamDestroy(cs);
    if (us) (void)hipStreamDestroy(us);
    if (d_tmp) (void)hipFree(d_tmp);
    (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_sites2);
    (void)hipFree(d_total); (void)hipFree(d_clump); (void)hipFree(d_defined);
    (void)hipFree(d_max); (void)hipFree(d_countsRaw); if (d_bins) (void)hipFree(d_bins);
    *out = c;
    return BBMAP_OK;
fail:
    if (cs) (void)hipStreamSynchronize(cs);
    if (us) (void)hipStreamSynchronize(us);
    for (hipEvent_t e : evUp) if (e) (void)hipEventDestroy(e);
    if (cs) (void)hipStreamDestroy(cs);
    if (us) (void)hipStreamDestroy(us);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_keys) (void)hipFree(d_keys);
    if (d_keys2) (void)hipFree(d_keys2);
    if (d_sites2) (void)hipFree(d_sites2);
    if (d_total) (void)hipFree(d_total);
    if (d_clump) (void)hipFree(d_clump);
    if (d_defined) (void)hipFree(d_defined);
    if (d_max) (void)hipFree(d_max);
    if (d_countsRaw) (void)hipFree(d_countsRaw);
    if (d_bins) (void)hipFree(d_bins);
    if (c) bbidx_destroy(c);
    return rc;
}

// the tunables bbidx_build derived (or bbidx_create was given)
extern "C" int bbidx_get_params(bbidx_ctx *c, bbidx_params *out) {
    if (!c || !out) { bbmap_set_error("bbidx_get_params: null argument"); return BBMAP_E_ARG; }
    *out = c->dev.p;
    return BBMAP_OK;
}

// copies a block's arrays back (tests, saving an index): any pointer may be NULL.  starts: 4^k+1 ints; sites: as many as
// starts[4^k]; counts: 4^k ints; lengthHistogram: 1001 ints.
extern "C" int bbidx_export_block(bbidx_ctx *c, int32_t block, int32_t *starts, int32_t *sites, int64_t sites_cap,
                                  int32_t *counts, int32_t *lengthHistogram) {
    if (!c || block < 0 || block >= c->dev.nblocks) { bbmap_set_error("bbidx_export_block: bad argument"); return BBMAP_E_ARG; }
    int rc = BBMAP_OK;
    const size_t nkeys = (size_t)1 << (2 * c->dev.p.k);
    {
        if (hipSetDevice(c->device) != hipSuccess) { bbmap_set_error("bbidx_export_block: hipSetDevice failed"); return BBMAP_E_HIP; }
        const int *dst = nullptr, *dsi = nullptr;
        BHIP(hipMemcpy(&dst, c->dev.starts + block, sizeof(void *), hipMemcpyDeviceToHost));
        BHIP(hipMemcpy(&dsi, c->dev.sites + block, sizeof(void *), hipMemcpyDeviceToHost));
        int n = 0;
        BHIP(hipMemcpy(&n, dst + nkeys, 4, hipMemcpyDeviceToHost));
        if (starts) BHIP(hipMemcpy(starts, dst, (nkeys + 1) * 4, hipMemcpyDeviceToHost));
        if (sites) {
            if (sites_cap < n) { bbmap_set_error("bbidx_export_block: sites buffer too small"); return BBMAP_E_ARG; }
            if (n > 0) BHIP(hipMemcpy(sites, dsi, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
        if (counts) BHIP(hipMemcpy(counts, c->dev.counts, nkeys * 4, hipMemcpyDeviceToHost));
        if (lengthHistogram) BHIP(hipMemcpy(lengthHistogram, c->dev.lengthHistogram, 1001 * 4, hipMemcpyDeviceToHost));
    }
    return BBMAP_OK;
fail:
    return rc;
}
