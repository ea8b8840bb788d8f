// Host-side context of the MultiStateAligner11ts entry points (msa_host.hip, msa_gapped.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "bbmap_amd.h"

struct bbmsa_ctx {
    bbmsa_config cfg;
    int device;
    int numCUs;
    int scheme;                 // BBMSA_SCHEME_*
    // fast kernel geometry
    int G, R, fastCols, tmpBytes, blocks, ldsBytes, tableLen, wideTableLen;
    long long dirSlotDwords;
    unsigned int *d_dir;
    unsigned int *d_counters;   // [0]=fast queue, [1]=slow count, [2]=generic queue
    int *d_slowList;
    long long slowCap;
    // generic kernel
    int genThreads;
    int *d_matrix;
    int *d_limits;
    // wide pass: the wavefront kernel again, 64 lanes per job and one job per 64-thread block, with an LDS column buffer as
    // wide as maxColumns, for the jobs the first pass found too wide for its own buffer (0 blocks = not needed)
    int wideR, wideCols, wideTmpBytes, wideBlocks, wideLdsBytes;
    long long wideDirSlotDwords;
    unsigned int *d_wideDir;
    int *d_slowList2;
    // narrow-window kernel (msa_fill_narrow.hip): one job per lane
    int narrowBlocks, narrowSlack;     // 0 blocks = disabled
    bool narrowOff;                    // switched off by the caller for launches whose jobs it cannot take (bbmsa_use_narrow)
    bool narrowUsed;                   // whether the last launch ran it
    // widest windows first (bbmsa_sort_by_width): two jobs share a wavefront and step together, so a 600-column job beside a
    // 200-column one idles half the wave for 400 steps; in width order neighbours are alike, and the longest jobs do not end up last
    bool sortByWidth;
    unsigned int *d_widthHist;
    long long latencyJobs;             // launches with at most this many jobs go straight to the 64-lane geometry (bbmsa_set_latency_jobs)
    unsigned long long *d_narrowDir;
    int *d_fastList;
    long long fastCap;
    // gapped-reference scratch (msa_gapped.hip), grown on demand
    uint8_t *d_gref;
    int *d_gaux;
    bbmsa_job *d_gjobs;
    long long gappedCap;
    hipEvent_t ev[4];      // start, after wavefront kernel, after generic kernel, after narrow kernel
    bool timed;
    bool banded;
    bool legacyOnly;            // created with BBMSA_LEGACY_ONLY: bbmsa_fill_submit / _collect / _packed only
    struct bbmsa_legacy *legacy;   // the per-call service of such a context (msa_legacy.hip)
    // strip-tiled wavefront kernel of the 9PacBio scheme (msa_fill_strip.hip)
    int stripBlocks, stripLds;
    long long stripDwords, stripSlotDwords;
    int *d_stripBoundary;
    uint8_t *d_stripTmp;
    // pipelined form of the strip kernel for launches with few jobs: pipeK wavefronts per job, pipeSlots jobs at a time
    int pipeK, pipeSlots, pipeJobsMax;
    int *d_pipeBoundary, *d_pipeSync;
};


// Internal (mapper.hip).  The mapper switches the narrow kernel off for launches it cannot help: small ones (one job per lane is a
// ~1.5 ms dependent chain however few jobs there are, in front of the wavefront kernel on the same stream) and the final alignment
// stage's (realign_new pads its windows by >= 6 columns and passes minScore - 120: no such fill fits the 16-diagonal band).
void bbmsa_use_narrow(bbmsa_ctx *c, bool on);
// Makes `waiter` (a stream) wait until the context's last launch sequence has reached its first wavefront pass.  The mapper's two
// contexts run side by side; the second one's sequence begins with make_gref_kernel, and if the first context's persistent blocks
// are resident on every CU by then, the second context's passes (among them the latency-bound wide pass, which needs something
// to overlap with) only start when those drain: measured, the final stage 86 -> 93 ms.
int bbmsa_wait_first_pass(bbmsa_ctx *c, void *waiter);
// Launches of this context hand their jobs to the wavefront kernel in descending window width (a counting sort by columns / 8 in
// front of the first pass); results are indexed by job as always.  The mapper asks for it on its second context, whose windows
// span 170..640+ columns.
void bbmsa_sort_by_width(bbmsa_ctx *c, bool on);
// Launches of at most n jobs skip the first pass and run in the wide pass's geometry (64 lanes x ceil(rows / 64) rows per lane, one job
// per block): with few jobs a launch costs one wavefront's dependent chain, and three rows per lane make a step ~450 instructions
// instead of ~740.  Measured on the mapper's late rounds (a few hundred fills each): 236.0 -> 230.9 ms per step for the second
// context alone.  0 = off (the default of a context).
int bbmsa_set_latency_jobs(bbmsa_ctx *c, int64_t n);

// msa_legacy.hip: persistent buffers, stream and the call combiner of a BBMSA_LEGACY_ONLY context (c->d_matrix / d_limits exist)
int bbmsa_legacy_create(bbmsa_ctx *c);
void bbmsa_legacy_destroy(bbmsa_ctx *c);
