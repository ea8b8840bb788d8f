// Host-side context of the index-probe entry points (index_probe.hip, index_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "bbmap_amd.h"
#include "index_common.h"

// What ONE probe launch writes besides its outputs: the persistent waves' work queue, the work counters, the two timing events and
// the long-read kernel's per-wave workspace.  The index itself (bbidx_ctx::dev) is read-only once built, so any number of launches
// may be in flight on it -- each with a launch state of its own.  The public entry points use the context's own (`own`: one launch
// at a time per bbidx_ctx); a bbmap_ctx makes one for itself, so that several mapping threads, each with its own bbmap_ctx, share one
// index as BBMap's threads share its BBIndex.
struct bbidx_launch {
    unsigned int *d_queue = nullptr;
    unsigned long long *d_stats = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool timed = false;
    int *d_longWs = nullptr;       // allocated at the first launch of the long-read kernel (index_probe_long.hip)
    int longBlocks = 0;
};

struct bbidx_ctx {
    int device;
    bbidx::DevIndex dev;
    std::vector<void *> allocs;
    bbidx_launch own;
    int blocks;
    int kernelKind;       // BBIDX_KERNEL_*
    long long totalSites; // list entries over all blocks (picks the wave kernel's long-list variant)
    int maxReadLen;       // bbidx_set_max_read_len: picks the wave kernel's LDS sizing (default BBIDX_MAX_READ_LEN)
};

int bbidx_launch_init(bbidx_ctx *c, bbidx_launch *ls);
void bbidx_launch_free(bbidx_launch *ls);
// bbidx_find_batch_device_rc / bbidx_last_stats with the caller's launch state
int bbidx_find_batch_device_with(bbidx_ctx *c, bbidx_launch *ls, void *stream, int64_t n, const bbidx_read *reads, const uint8_t *bases,
                                 const int8_t *baseScores, const int32_t *keyinfo, bbidx_site *sites, int32_t max_sites, int32_t *nsites,
                                 uint8_t *bases_rc_out);
int bbidx_last_stats_with(bbidx_ctx *c, bbidx_launch *ls, int64_t *stats5, float *kernel_ms);


// Shared tail of bbidx_create / bbidx_build: the fused key table (from the per-block device arrays), queue, work counters,
// events.  `starts` / `sites` are the per-block DEVICE pointers, already recorded in c->allocs.
int bbidx_finish_create(bbidx_ctx *c, const std::vector<const int *> &starts, const std::vector<const int *> &sites);

// index_probe_long.hip: one read per wavefront for reads of up to 6016 bases with up to 2047 keys (either profile)
long long bbidx_long_workspace_ints_per_block();
int bbidx_long_blocks(int profile);
int bbidx_launch_long(const bbidx::Params &P, hipStream_t stream, int profile, int *ws, int blocks);
