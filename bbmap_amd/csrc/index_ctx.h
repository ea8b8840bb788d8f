// Host-side context of the index-probe entry points (index_probe.hip, index_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "bbmap_amd.h"
#include "index_common.h"

struct bbidx_ctx {
    int device;
    bbidx::DevIndex dev;
    std::vector<void *> allocs;
    unsigned int *d_queue;
    unsigned long long *d_stats;
    int blocks;
    hipEvent_t ev[2];
    bool timed;
    int kernelKind;       // BBIDX_KERNEL_*
    long long totalSites; // list entries over all blocks (picks the wave kernel's long-list variant)
    int maxReadLen;       // bbidx_set_max_read_len: picks the wave kernel's LDS sizing (default BBIDX_MAX_READ_LEN)
    int *d_longWs;        // per-wave workspace of the long-read kernel (index_probe_long.hip), allocated at its first launch
    int longBlocks;
};


// Shared tail of bbidx_create / bbidx_build: the fused key table (from the per-block device arrays), queue, work counters,
// events.  `starts` / `sites` are the per-block DEVICE pointers, already recorded in c->allocs.
int bbidx_finish_create(bbidx_ctx *c, const std::vector<const int *> &starts, const std::vector<const int *> &sites);

// index_probe_long.hip: one read per wavefront for reads of up to 6016 bases with up to 2047 keys (either profile)
long long bbidx_long_workspace_ints_per_block();
int bbidx_long_blocks(int profile);
int bbidx_launch_long(const bbidx::Params &P, hipStream_t stream, int profile, int *ws, int blocks);
